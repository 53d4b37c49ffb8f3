"""How a tile configuration's time grows with the number of row tiles (GPU only): T(M) at fixed (N, K, cfg), M in whole
BM-row tiles.  A chip of 256 CUs with `occ` resident workgroups per CU runs tiles <= 256 at one per CU, <= 256 * occ all
resident; if T(tiles = 324) is well above T(tiles = 252) the launch is throughput-bound and the doubly loaded CUs set its
time; if the two agree the launch is latency-bound per workgroup and a hybrid tail has nothing to win.
usage: PYTHONPATH=. python tools/bench_gemm_msweep.py"""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")
import ctypes
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops


def time_call(call, n=40, rounds=3):
    for _ in range(3):
        call()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            call()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def run():
    lib = _lib.load()
    dev = torch.device("cuda")
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # (N, K, cfg, BM, BN, row-tile counts)
    plans = [
        (768, 768, 65, 192, 128, (14, 28, 36, 42, 43, 56)),
        (768, 3072, 65, 192, 128, (14, 28, 36, 42, 43, 56)),
        (768, 3072, 66, 192, 128, (28, 36, 42)),
        (768, 3072, 67, 192, 128, (28, 36, 42)),
        (3072, 768, 65, 192, 128, (10, 21, 32, 36)),
    ] if len(sys.argv) > 1 and sys.argv[1] == "r6" else [
        (768, 768, 15, 128, 128, (21, 32, 42, 43, 48, 54, 64, 85, 86)),
        (768, 3072, 15, 128, 128, (21, 32, 42, 43, 48, 54, 64, 85, 86)),
        (3072, 768, 15, 128, 128, (10, 11, 16, 21, 22, 32, 43, 54)),
        (2304, 768, 36, 256, 256, (14, 21, 27, 28, 29, 42, 56)),
        (768, 768, 13, 128, 64, (8, 15, 21, 22, 32, 43)),
        (768, 3072, 13, 128, 64, (8, 15, 21, 22, 32, 43)),
    ]
    for (N, K, cfg, BM, BN, tms) in plans:
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        pws = [ops.PackedLinear(w, b) for _ in range(8)]
        print("N=%d K=%d cfg %d (%dx%d tiles)" % (N, K, cfg, BM, BN), flush=True)
        for tm in tms:
            M = tm * BM
            x = torch.randn(M, K, device=dev)
            a = ops.split_rows(x)
            c = torch.empty(M, N, device=dev)
            ctr = [0]

            def call():
                q = pws[ctr[0] % len(pws)]
                ctr[0] += 1
                rc = lib.gridmm_linear_planes_cfg(a.hi.data_ptr(), a.lo.data_ptr(), K, q.hi.data_ptr(), q.lo.data_ptr(), q.Kp,
                                                  b.data_ptr(), None, 0, c.data_ptr(), N, None, None, 0, M, N, K, 0, cfg, st())
                assert rc == 0
            us = time_call(call)
            tiles = tm * ((N + BN - 1) // BN)
            print("   M=%5d tiles=%4d (%.2f per CU)  %7.1f us  %6.0f TF alg  %6.3f us/tile" %
                  (M, tiles, tiles / 256.0, us, 2.0 * M * N * K / us / 1e6, us / tiles), flush=True)


if __name__ == "__main__":
    run()
