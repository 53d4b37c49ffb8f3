"""Is the planes GEMM power/clock-limited?  Same launches on random vs zero-filled operands (GPU only)."""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")   # development build: tile overrides + the whole experiment table (make -C gridmm_amd/csrc debug)
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops

lib = _lib.load(); dev = torch.device("cuda")
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, cfgs) in [(9472, 6144, 768, (36, 15)), (6912, 768, 768, (15, 50)), (6912, 3072, 768, (50, 15)), (1824, 768, 768, (8,))]:
    for cfg in cfgs:
        line = "%5d x %4d x %4d cfg %3d |" % (M, N, K, cfg)
        for kind in ("randn", "zeros", "small-int"):
            if kind == "randn":
                x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
            elif kind == "zeros":
                x = torch.zeros(M, K, device=dev); w = torch.zeros(N, K, device=dev)
            else:   # exactly representable in one bf16: the lo planes are all zero
                x = torch.randint(-8, 8, (M, K), device=dev).float(); w = torch.randint(-8, 8, (N, K), device=dev).float()
            b = torch.zeros(N, device=dev)
            pw = ops.PackedLinear(w, b); a = ops.split_rows(x); c = torch.empty(M, N, device=dev)
            def call():
                return lib.gridmm_linear_planes_cfg(a.hi.data_ptr(), a.lo.data_ptr(), K, pw.hi.data_ptr(), pw.lo.data_ptr(), pw.Kp,
                    b.data_ptr(), None, 0, c.data_ptr(), N, None, None, 0, M, N, K, 0, cfg, st())
            assert call() == 0
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(40): call()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 40
            line += " %s %6.1fus %4.0fTF |" % (kind, us, 2.0 * M * N * K / us / 1e6)
        print(line, flush=True)
