"""Is the planes GEMM bound by where its operands come from?  Same launch with the A rows collapsed onto one row
(lda = 0: the whole A stream is L1/L2-resident) and / or a tiny W (all tiles read the same weight rows) -- GPU only."""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")   # development build: tile overrides + the whole experiment table (make -C gridmm_amd/csrc debug)
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda")
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def gtime(call, n=40):
    assert call() == 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): call()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K, cfgs) in [(6912, 768, 768, (15, 14, 36)), (6912, 2304, 768, (36, 15)), (6912, 768, 3072, (15,)), (1824, 768, 768, (8,)), (9472, 6144, 768, (36,))]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.zeros(N, device=dev)
    pw = ops.PackedLinear(w, b); a = ops.split_rows(x); c = torch.empty(M, N, device=dev)
    for cfg in cfgs:
        def mk(lda, out):
            return lambda: lib.gridmm_linear_planes_cfg(a.hi.data_ptr(), a.lo.data_ptr(), lda, pw.hi.data_ptr(), pw.lo.data_ptr(), pw.Kp,
                b.data_ptr(), None, 0, c.data_ptr() if out else None, N, None if out else a.hi.data_ptr(), None if out else a.lo.data_ptr(), 0 if out else 0, M, N, K, 0, cfg, st())
        t_full = gtime(mk(K, True)); t_res = gtime(mk(0, True))
        print("%5d x %4d x %4d cfg %3d | normal %6.1f us | A rows collapsed (cache-resident A) %6.1f us" % (M, N, K, cfg, t_full, t_res), flush=True)
