"""Micro-benchmark of gridmm_linear_planes tile configurations on the step's GEMM shapes (GPU only)."""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")   # development build: tile overrides + the whole experiment table (make -C gridmm_amd/csrc debug)
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops

SHAPES = [(int(v) for v in os.environ['GEMM_SHAPE'].split('x'))] if os.environ.get('GEMM_SHAPE') else [(6912, 2304, 768), (6912, 768, 768), (6912, 3072, 768), (6912, 768, 3072), (9472, 6144, 768),
          (2560, 1536, 768), (1824, 768, 768), (1824, 2304, 768), (1824, 3072, 768), (1824, 768, 3072),
          (2560, 512, 768), (6272, 768, 512)]
CFGS = {8: "64x64 BK64", 43: "TR 64x64 BK64", 14: "128x128 8w", 15: "128x128 16w", 36: "256x256 16w", 7: "256x256 8w", 16: "256x128 16w",
        65: "192x128 16w NS3", 66: "192x128 16w BK64", 67: "192x128 16w NS4", 68: "256x128 16w NS3", 56: "96x64 NS3 BK64", 69: "96x64 TR",
        76: "192x128 BK32 two per CU", 70: "192x128 BK64 RP", 71: "128x64 NS3 RP", 72: "128x128 BK64 RP", 75: "128x64 NS3 RP TR", 13: "128x64 NS3",
        466: "192x128 BK64 DMAonly same tile", 566: "192x128 BK64 DMAonly 8 row tiles", 467: "192x128 NS4 DMAonly same tile", 365: "192x128 NS3 DMAonly", 367: "192x128 NS4 DMAonly", 165: "192x128 NS3 noMFMA", 167: "192x128 NS4 noMFMA", 265: "192x128 NS3 noDMA", 166: "192x128 BK64 noMFMA", 266: "192x128 BK64 noDMA", 366: "192x128 BK64 DMAonly", 115: "128x128 noMFMA", 215: "128x128 noDMA", 315: "128x128 DMAonly",
        136: "256x256 noMFMA", 236: "256x256 noDMA", 336: "256x256 DMAonly", 113: "128x64 noMFMA", 213: "128x64 noDMA", 313: "128x64 DMAonly"}


def run(cfgs=None):
    lib = _lib.load()
    dev = torch.device("cuda")
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (M, N, K) in SHAPES:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        # COLD weights, as in the real step (161 M parameters never stay in the 256 MB Infinity Cache):
        # rotate over enough copies of W to exceed the cache
        ncopy = max(2, int(700e6 // (N * K * 4)) + 1)
        pws = [ops.PackedLinear(w, b) for _ in range(min(ncopy, 64))]
        pw = pws[0]
        a = ops.split_rows(x)
        ref = (x.double() @ w.double().t() + b.double()).float()
        c = torch.empty(M, N, device=dev)
        line = "%5d x %4d x %4d |" % (M, N, K)
        for cfg, name in CFGS.items():
            if cfgs and cfg not in cfgs:
                continue
            ctr = [0]

            def call():
                q = pws[ctr[0] % len(pws)]
                ctr[0] += 1
                return lib.gridmm_linear_planes_cfg(
                    a.hi.data_ptr(), a.lo.data_ptr(), K, q.hi.data_ptr(), q.lo.data_ptr(), q.Kp, b.data_ptr(), None, 0,
                    c.data_ptr(), N, None, None, 0, M, N, K, 0, cfg, st())
            c.zero_()
            assert call() == 0
            torch.cuda.synchronize()
            err = (c - ref).abs().max().item() / max(1.0, ref.abs().max().item())
            for _ in range(3):
                call()
            # device time: 40 calls captured in one hipGraph (eager launches from python are host-bound at ~8 us)
            n = 40
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n):
                    call()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            line += "\n      %-20s %6.1fus %5.0fTF%s" % (name, us, 2.0 * M * N * K / us / 1e6, "" if (err < 1e-4 or 100 <= cfg < 600) else " ERR %.1e" % err)
        print(line, flush=True)


if __name__ == "__main__":
    run([int(v) for v in sys.argv[1:]] or None)
