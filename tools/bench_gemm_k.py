"""Fixed cost vs per-k-step cost of gridmm_linear_planes: time over K for a few (M, N, cfg) -- GPU only."""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")   # development build: tile overrides + the whole experiment table (make -C gridmm_amd/csrc debug)
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops

def run():
    lib = _lib.load(); dev = torch.device("cuda")
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (M, N, cfgs) in [(1824, 768, (8, 43, 53, 15)), (1824, 2304, (15, 8)), (6912, 768, (15, 50, 36)), (6912, 2304, (36, 15))]:
        for cfg in cfgs:
            line = "%5d x %4d cfg %3d |" % (M, N, cfg)
            for K in (64, 256, 768, 1536, 3072):
                x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
                pws = [ops.PackedLinear(w, b) for _ in range(8)]
                a = ops.split_rows(x); c = torch.empty(M, N, device=dev)
                ctr = [0]
                def call():
                    q = pws[ctr[0] % 8]; ctr[0] += 1
                    return lib.gridmm_linear_planes_cfg(a.hi.data_ptr(), a.lo.data_ptr(), K, q.hi.data_ptr(), q.lo.data_ptr(), q.Kp,
                        b.data_ptr(), None, 0, c.data_ptr(), N, None, None, 0, M, N, K, 0, cfg, st())
                assert call() == 0
                for _ in range(3): call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50): call()
                e1.record(); torch.cuda.synchronize()
                line += " K=%4d %6.1fus |" % (K, e0.elapsed_time(e1) * 1e3 / 50)
            print(line, flush=True)
    # launch floor: a trivial kernel back to back
    x = torch.randn(32, 57, 768, device=dev); y = torch.empty_like(x)
    for _ in range(3): ops.copy_rows(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ops.copy_rows(x, y)
    e1.record(); torch.cuda.synchronize()
    print("copy_rows 1824x768 fp32: %.1f us per launch (eager back-to-back)" % (e0.elapsed_time(e1) * 10))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(100): ops.copy_rows(x, y)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("copy_rows in a graph: %.2f us per launch" % (e0.elapsed_time(e1) * 10))

if __name__ == "__main__":
    run()
