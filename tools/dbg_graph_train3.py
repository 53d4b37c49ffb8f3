"""Diagnostic driver of the captured pre-training step (reduced or `full` size; env DROP / TIMING): builds the trainer, captures
one task graph and replays it -- the workload that tools/train_graph_gaps.sh traces to tell a GPU-bound replay from a
dispatch-bound one.  usage (repo root): python tools/dbg_graph_train3.py [full]"""
import sys, copy, time
sys.path.insert(0, ".")
import numpy as np, torch, os
from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
from gridmm_amd.pretrain_loop import PreTrainer, default_opts
from gridmm_amd.synthetic import batch_to, make_pretrain_batch
from gridmm_amd.train_graph import GraphedTrainStep
from gridmm_amd.vilmodel import default_config
dev = torch.device("cuda:0")
full = len(sys.argv) > 1 and sys.argv[1] == "full"
kw = {} if full else dict(num_l_layers=2, num_pano_layers=1, num_x_layers=2)
B = 32 if full else 4
cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=["mlm", "mrc", "sap"], image_prob_size=1000, obj_prob_size=0, **kw)
if os.environ.get('DROP'):
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = float(os.environ['DROP'])
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg).to(dev)
tr = PreTrainer(model, default_opts(warmup_steps=100, learning_rate=float(os.environ.get('LR', '5e-5'))))
tasks = ("mlm", "mrc", "sap") if len(sys.argv) < 3 else tuple(sys.argv[2].split(","))
batches = {t: batch_to(make_pretrain_batch(np.random.RandomState(i), B, t, max_steps=5 if full else 3, L=80 if full else 40, vocab=30000, image_prob_size=1000,
                                           n_pts=(588 * 3, 588 * 5) if full else (588, 1176)), dev) for i, t in enumerate(tasks)}
graphs = {}
if os.environ.get('EAGER_REPLAY'):
    from gridmm_amd import hostsync as hs, autograd as ag
    ag.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    t = tasks[0]
    with hs.record() as tape:
        tr.train_step(batches[t], t)
    for i in range(8):
        with hs.replay(tape):
            l, n = tr.train_step(batches[t], t)
        torch.cuda.synchronize(); print('eager replay', i, float(l.mean()), float(n), flush=True)
    sys.exit(0)
for t in tasks:
    graphs[t] = GraphedTrainStep(tr, batches[t], t, capture_optimizer=not os.environ.get('EAGER_OPT'))
    torch.cuda.synchronize(); print("captured", t, flush=True)
    graphs[t](); torch.cuda.synchronize(); print("replayed", t, flush=True)
if os.environ.get('NOREFRESH'):
    tr.optimizer.refresh_graph_tables = lambda *a, **k: None
for i in range(6):
    t = tasks[i % len(tasks)]
    l, n = graphs[t]()
    torch.cuda.synchronize()
    print(i, t, float(l.mean()), float(n), flush=True)
    if os.environ.get('LIGHT'):
        w = model.bert.grid_proj
        print('   grid_proj |w|max %.4g |b|max %.4g' % (float(w.weight.float().abs().max()), float(w.bias.float().abs().max())), 'state', {k: (float(v.float().abs().max()) if torch.is_tensor(v) else v) for k, v in tr.optimizer.state[w.weight].items()}, flush=True)
    if os.environ.get('SLEEP'):
        time.sleep(float(os.environ['SLEEP']))
    if os.environ.get('ALLOC'):
        junk = [torch.full((1 << 20,), 1e30, device=dev) for _ in range(int(os.environ['ALLOC']))]
        torch.cuda.synchronize(); del junk
if os.environ.get('TIMING'):
    torch.cuda.synchronize()
    t = tasks[0]
    hs_, tot = 0.0, time.perf_counter()
    for i in range(10):
        a = time.perf_counter(); graphs[t](); hs_ += time.perf_counter() - a
    torch.cuda.synchronize(); tot = time.perf_counter() - tot
    print("per step: host-side call %.2f ms, wall %.2f ms" % (hs_ * 100, tot * 100))
