"""Device time (hipGraph-timed) of gridmm_transpose_v and gridmm_attention_planes per attention shape of the step."""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")   # development build: tile overrides + the whole experiment table (make -C gridmm_amd/csrc debug)
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops

def gtime(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

lib = _lib.load(); dev = torch.device("cuda"); B, heads = 32, 12
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for (name, Sq, Sk, Wq, Wk) in [("grid self", 216, 216, 2304, 2304), ("grid x text", 216, 80, 768, 1536), ("local x kv", 57, 296, 768, 6144), ("local self", 57, 57, 2304, 2304)]:
    qb = ops.split_rows(torch.randn(B, Sq, Wq, device=dev)); kb = qb if (Wq == Wk and Sq == Sk) else ops.split_rows(torch.randn(B, Sk, Wk, device=dev))
    qh, ql = qb.hi[..., :768], qb.lo[..., :768]
    kh, kl = kb.hi[..., Wk - 1536:Wk - 768], kb.lo[..., Wk - 1536:Wk - 768]
    vh, vl = kb.hi[..., Wk - 768:], kb.lo[..., Wk - 768:]
    mask = torch.ones(B, Sk, dtype=torch.uint8, device=dev)
    Skp = (Sk + 31) // 32 * 32
    th = torch.empty(B, heads, Skp // 32, 64, 32, dtype=torch.bfloat16, device=dev); tl = torch.empty_like(th)
    hi, lo = ops._planes_like((B, Sq, 768), dev)
    def tr():
        assert lib.gridmm_transpose_v(p(vh), p(vl), vh.stride(0), vh.stride(1), p(th), p(tl), B, heads, Sk, Skp, st()) == 0
    def at():
        assert lib.gridmm_attention_planes(p(qh), p(ql), qh.stride(0), qh.stride(1), p(kh), p(kl), kh.stride(0), kh.stride(1), p(th), p(tl), Skp,
            p(mask), mask.stride(0), None, 0, 0, p(hi), p(lo), Sq * 768, 768, B, heads, Sq, Sk, 0.125, st()) == 0
    def rows(cfg):
        def f():
            assert lib.gridmm_attention_rows_cfg(p(qh), p(ql), qh.stride(0), qh.stride(1), p(kh), p(kl), kh.stride(0), kh.stride(1),
                p(vh), p(vl), vh.stride(0), vh.stride(1), p(mask), mask.stride(0), None, 0, 0, p(hi), p(lo), Sq * 768, 768,
                B, heads, Sq, Sk, 0.125, cfg, st()) == 0
        return f
    only = [int(v) for v in sys.argv[1:]]
    if only:      # profiling runs: eager launches of the chosen configurations only
        for c in only:
            for _ in range(3): rows(c)()
        torch.cuda.synchronize()
        continue
    print("%-12s attention_rows cfg 1,2,3,5,6,7,8,9 | 14..18:" % name, " ".join("%5.1f" % gtime(rows(c)) for c in (1, 2, 3, 5, 6, 7, 8, 9)), "|", " ".join("%5.1f" % gtime(rows(c)) for c in (14, 15, 16, 17, 18)), "us | cfg 2 without staging %.1f, without math %.1f, neither %.1f" % (gtime(rows(11)), gtime(rows(12)), gtime(rows(13))), flush=True)
    mf = 4.0 * B * Sq * Sk * 768 * 3
    t1, t2 = gtime(tr), gtime(at)
    print("%-12s Sq=%3d Sk=%3d | transpose_v %5.1f us | attention_planes %5.1f us (%.0f TF on the pipe; MFMA floor %.1f us)" % (name, Sq, Sk, t1, t2, mf / t2 / 1e6, mf / 2.5e9), flush=True)

# ---- layout probe: the same work with per-head CONTIGUOUS planes ([B*heads][S][64], heads = 1 per "batch" entry)
print("per-head contiguous layout (B*12 single-head problems):")
for (name, Sq, Sk) in [("grid self", 216, 216), ("grid x text", 216, 80), ("local x kv", 57, 296), ("local self", 57, 57)]:
    Bh = B * heads
    qa = ops.split_rows(torch.randn(Bh, Sq, 64, device=dev)); ka = ops.split_rows(torch.randn(Bh, Sk, 64, device=dev)); va = ops.split_rows(torch.randn(Bh, Sk, 64, device=dev))
    mask = torch.ones(Bh, Sk, dtype=torch.uint8, device=dev)
    hi, lo = ops._planes_like((Bh, Sq, 64), dev)
    def rows(cfg):
        def f():
            assert lib.gridmm_attention_rows_cfg(p(qa.hi), p(qa.lo), Sq * 64, 64, p(ka.hi), p(ka.lo), Sk * 64, 64, p(va.hi), p(va.lo), Sk * 64, 64,
                p(mask), Sk, None, 0, 0, p(hi), p(lo), Sq * 64, 64, Bh, 1, Sq, Sk, 0.125, cfg, st()) == 0
        return f
    print("%-12s attention_rows cfg 1,2,3,5,6:" % name, " ".join("%5.1f" % gtime(rows(c)) for c in (1, 2, 3, 5, 6)), "us | cfg 2 without staging %.1f, without math %.1f, neither %.1f" % (gtime(rows(11)), gtime(rows(12)), gtime(rows(13))), flush=True)
