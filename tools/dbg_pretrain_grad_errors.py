"""Diagnostic: per-parameter gradient error of the full-size pre-training twin against the reference's fixture
(tests/golden/pretrain_full_b2.npz), for A/B runs of a kernel change (e.g. GRIDMM_TRAIN_ATTENTION_BF16=0/1: the numbers in
profiles/r5_attention_train_bench.txt).  usage (repo root): python tools/dbg_pretrain_grad_errors.py mlm|mrc|sap"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from conftest import load_golden
import test_hip_pretrain as T
from oracle import gen_golden
from gridmm_amd.synthetic import batch_to
task = sys.argv[1]
fx = load_golden("pretrain_full_b2.npz")
model = T._model(fx)
batch = batch_to(gen_golden.pretrain_full_batch(task), "cuda")
loss = model(batch, task=task, compute_loss=True)
loss.mean().backward()
names = json.loads(str(fx["grad_names_" + task])); params = dict(model.named_parameters())
norms, samples = fx["grad_norms_" + task], fx["grad_samples_" + task]
scale = float(norms.max()); o = 0; errs = []
for k, n_ref in zip(names, norms):
    g = params[k].grad.detach().float().reshape(-1).cpu()
    idx = gen_golden.grad_sample_index(k, g.numel()); ref = samples[o:o + len(idx)]; o += len(idx)
    denom = max(float(np.abs(ref).max()), 1e-3 * scale / np.sqrt(max(g.numel(), 1)), 1e-12)
    e = float(np.abs(g[torch.from_numpy(idx)].numpy() - ref).max()) / denom
    errs.append((e, k, float(np.abs(ref).max()), float(n_ref)))
errs.sort(reverse=True)
for e in errs[:12]: print("%.2e %s refmax %.2e norm %.2e" % (e[0], e[1], e[2], e[3]))
print("median", np.median([e[0] for e in errs]))
