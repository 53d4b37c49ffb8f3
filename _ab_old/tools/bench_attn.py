"""Micro-benchmark of gridmm_attention_planes: fused-QKV strided views vs contiguous planes (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import ops

def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

dev = torch.device("cuda")
B = 32
for (Sq, Sk, W) in [(216, 216, 2304), (57, 296, 6144), (57, 57, 2304), (216, 80, 1536)]:
    big = ops.split_rows(torch.randn(B, max(Sq, Sk), W, device=dev))
    mask = torch.ones(B, Sk, dtype=torch.uint8, device=dev)
    qs = (big.hi[:, :Sq, :768], big.lo[:, :Sq, :768])
    ks = (big.hi[:, :Sk, W - 1536:W - 768], big.lo[:, :Sk, W - 1536:W - 768])
    vs = (big.hi[:, :Sk, W - 768:], big.lo[:, :Sk, W - 768:])
    qc = tuple(x.contiguous() for x in qs); kc = tuple(x.contiguous() for x in ks); vc = tuple(x.contiguous() for x in vs)
    a = t(lambda: ops.attention_planes(qs, ks, vs, mask))
    c = t(lambda: ops.attention_planes(qc, kc, vc, mask))
    # f32 kernel for comparison
    f = (big.hi.float() + big.lo.float())
    fq, fk, fv = f[:, :Sq, :768], f[:, :Sk, W - 1536:W - 768], f[:, :Sk, W - 768:]
    d = t(lambda: ops.attention(fq, fk, fv, mask))
    print("Sq=%3d Sk=%3d rowstride=%4d | bf16x3 strided %6.1f us | bf16x3 contiguous %6.1f us | f32 strided %6.1f us"
          % (Sq, Sk, W, a, c, d), flush=True)
