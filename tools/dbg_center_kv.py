"""Experiment: does centring K / V over the keys (softmax is invariant to a shift of K; P V' + mean V = P V) shrink the
error of the bf16x3 attention backward on the near-zero q / k weight gradients?  Patches the op-by-op path (no fused layer)."""
import os, sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import load_golden
import test_hip_pretrain as T
from oracle import gen_golden
from gridmm_amd.synthetic import batch_to
from gridmm_amd import autograd as ag, ops, vilmodel_train as VT
task, center = sys.argv[1], int(sys.argv[2])
VT.FUSED_XLAYER = False

def retag(t):
    a = ops.split_rows(t.detach().contiguous())
    return ag._tag_planes(t, a.hi.view(t.shape), a.lo.view(t.shape))

if center:
    orig_self, orig_cross = ag.self_attention, ag.cross_attention
    def self_attention(qkv, kmask, heads, dropout_p=0.0):
        H = heads * 64
        w = (kmask.float() / kmask.float().sum(1, keepdim=True).clamp_min(1))[..., None]
        mean = (qkv[..., H:] * w).sum(1, keepdim=True)
        q2 = torch.cat([qkv[..., :H], qkv[..., H:] - mean], -1)
        y = orig_self(retag(q2), kmask, heads, dropout_p)
        return y + mean[..., H:]
    def cross_attention(q, kv, kmask, heads, kv_col=0, dropout_p=0.0):
        H = heads * 64
        w = (kmask.float() / kmask.float().sum(1, keepdim=True).clamp_min(1))[..., None]
        mean = (kv * w).sum(1, keepdim=True)
        y = orig_cross(retag(q.contiguous() + 0), retag(kv - mean), kmask, heads, kv_col, dropout_p)
        return y + mean[..., kv_col + H:kv_col + 2 * H]
    ag.self_attention, ag.cross_attention = self_attention, cross_attention
    VT.ag = ag
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
    errs.append((e, k))
errs.sort(reverse=True)
print("center", center, "bf16", ag.BF16_ATTENTION, " ".join("%.1e:%s" % (e, k.split("bert.")[-1][-40:]) for e, k in errs[:6]))
