"""Full-size check of the captured training step against the eager step (dropout off): loss / norm trajectories over
N steps cycling mlm / mrc / sap, two identically initialised models.  Runs on the default runtime settings (kernel-only graphs)."""
import os, sys, copy
sys.path.insert(0, ".")
import numpy as np, torch
from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
from gridmm_amd.pretrain_loop import PreTrainer, default_opts
from gridmm_amd.synthetic import batch_to, make_pretrain_batch
from gridmm_amd.train_graph import GraphedTrainStep
from gridmm_amd.vilmodel import default_config
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=["mlm", "mrc", "sap"], image_prob_size=1000, obj_prob_size=0,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(0)
m0 = GlocalTextPathCMTPreTraining(cfg).to(dev)
if not os.environ.get('FP16_GRID_PROJ'):
    m0.bert.grid_proj.float()     # see tests/train_graph_cases.py: the fp16 parameter's rounding noise splits trajectories
tasks = tuple(os.environ.get("TASKS", "mlm,mrc,sap").split(","))
batches = {t: batch_to(make_pretrain_batch(np.random.RandomState(i), 32, t, max_steps=5, L=80, vocab=30000, image_prob_size=1000,
                                           n_pts=(588 * 3, 588 * 5)), dev) for i, t in enumerate(tasks)}
ma, mb = copy.deepcopy(m0), copy.deepcopy(m0)
ta, tb = PreTrainer(ma, default_opts(warmup_steps=20)), PreTrainer(mb, default_opts(warmup_steps=20))
graphs = {}
for t in tasks:                               # building a graph runs two eager steps of its task (record + warm-up)
    for _ in range(2):
        ta.train_step(batches[t], t)
    if os.environ.get('EAGER_BOTH'):
        for _ in range(2):
            tb.train_step(batches[t], t)
        graphs[t] = (lambda t=t: tb.train_step(batches[t], t))
    else:
        graphs[t] = GraphedTrainStep(tb, batches[t], t)
worst = 0.0
for i in range(n):
    t = tasks[i % len(tasks)]
    la, na = ta.train_step(batches[t], t)
    lb, nb = graphs[t]()
    d = float((la - lb).abs().max()) / max(1.0, float(la.abs().max()))
    worst = max(worst, d)
    print("%2d %s loss %.5f / %.5f  norm %.4f / %.4f  rel diff %.2e" % (i, t, float(la.mean()), float(lb.mean()), float(na), float(nb), d), flush=True)
pd = max(float((pa.float() - pb.float()).abs().max()) for pa, pb in zip(ma.parameters(), mb.parameters()))
print("worst relative loss difference %.2e, max parameter difference %.2e after %d + 6 steps" % (worst, pd, n))
