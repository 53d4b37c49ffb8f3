import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gridmm_amd import autograd as ag
from test_hip_train_rowops import _fusion_inputs
g_raw, l_raw, gr_raw, f_raw, gm, gv, vn, con, cv = [t.cuda() for t in _fusion_inputs(4, 7, 6, 1)]
leaves = [t.clone().requires_grad_(True) for t in (g_raw, l_raw, gr_raw, f_raw)]
outs = ag.fuse_logits(*leaves, gm, gv, vn, con, cv)
f = outs[3]
torch.where(torch.isfinite(f), f, torch.zeros_like(f)).sum().backward()
print([None if l.grad is None else float(l.grad.abs().sum()) for l in leaves])
