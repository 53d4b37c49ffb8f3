"""Stability soak of the captured training step: N replays cycling the three task graphs at full size (the configuration of
bench.py's train leg), losses finite and decreasing, no device fault.  usage: soak_train_graph.py [replays=300]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
from gridmm_amd.pretrain_loop import PreTrainer, default_opts
from gridmm_amd.synthetic import batch_to, make_pretrain_batch
from gridmm_amd.train_graph import GraphedTrainStep
from gridmm_amd.vilmodel import default_config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=["mlm", "mrc", "sap"], image_prob_size=1000, obj_prob_size=0)
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg).to(dev)
tr = PreTrainer(model, default_opts(warmup_steps=100))
tasks = ("mlm", "mrc", "sap")
batches = {t: batch_to(make_pretrain_batch(np.random.RandomState(i), 32, t, max_steps=5, L=80, vocab=30000, image_prob_size=1000,
                                           n_pts=(588 * 3, 588 * 5)), dev) for i, t in enumerate(tasks)}
graphs = {t: GraphedTrainStep(tr, batches[t], t) for t in tasks}
first, last = {}, {}
t0 = time.perf_counter()
for i in range(n):
    t = tasks[i % 3]
    l, g = graphs[t]()
    if i < 3 or i >= n - 3 or i % 60 == 0:
        v = float(l.mean())
        assert np.isfinite(v) and np.isfinite(float(g)), (i, t, v, float(g))
        first.setdefault(t, v); last[t] = v
        print(i, t, "loss %.4f norm %.3f" % (v, float(g)), flush=True)
torch.cuda.synchronize()
print("replays %d in %.1f s (%.2f ms/step); loss first -> last:" % (n, time.perf_counter() - t0, 1e3 * (time.perf_counter() - t0) / n),
      {t: (round(first[t], 3), round(last[t], 3)) for t in tasks})
assert all(last[t] < first[t] for t in tasks)
print("ok")
