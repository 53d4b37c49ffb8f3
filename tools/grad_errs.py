"""Diagnostic: loss and worst gradient errors of the pre-training twin against both reference fixtures (reduced and full
size), all three tasks -- the table behind the bounds asserted in tests/test_hip_pretrain.py.
usage (repo root): python tools/grad_errs.py"""
import os
import json, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_golden
from oracle import gen_golden
import test_hip_pretrain as TP
from gridmm_amd.synthetic import batch_to
for task, kind in (("mlm", False), ("mrc", False), ("sap", False), ("mlm", "full"), ("mrc", "full"), ("sap", "full")):
    full = kind == "full"
    fx = load_golden("pretrain_full_b2.npz" if full else "pretrain_reduced.npz")
    model = TP._model(fx)
    batch = batch_to(gen_golden.pretrain_full_batch(task) if full else gen_golden.pretrain_batch(task, False), "cuda")
    loss = model(batch, task=task, compute_loss=True)
    want = fx["loss_" + task]
    print(task, kind, "loss rel err", np.abs(loss.detach().cpu().numpy() - want).max() / max(1, np.abs(want).max()))
    loss.mean().backward()
    names = json.loads(str(fx["grad_names_" + task])); params = dict(model.named_parameters())
    norms, samples = fx["grad_norms_" + task], fx["grad_samples_" + task]
    scale = float(norms.max()); o = 0; errs = []
    for k, n_ref in zip(names, norms):
        g = params[k].grad.detach().float().reshape(-1).cpu()
        idx = gen_golden.grad_sample_index(k, g.numel()); ref = samples[o:o + len(idx)]; o += len(idx)
        denom = max(float(np.abs(ref).max()), 1e-3 * scale / np.sqrt(max(g.numel(), 1)), 1e-12)
        e = float(np.abs(g[torch.from_numpy(idx)].numpy() - ref).max()) / denom
        en = abs(float(g.norm()) - float(n_ref)) / max(float(n_ref), 1e-3 * scale)
        errs += [(e, k), (en, k + " [norm]")]
    errs.sort(reverse=True)
    print("   top:", [(round(e, 4), k[-50:]) for e, k in errs[:5]])
    print("   median err: %.2e   frac > 5e-3: %.3f" % (np.median([e for e, _ in errs]), np.mean([e > 5e-3 for e, _ in errs])))
