"""Micro-benchmark: row-major vs tiled W vs tiled W + tiled A planes (16 x 32 blocks) for the BK = 32 tiles, hipGraph of 40 calls
with rotating cold weights (GPU only).  The A-side experiment: is tiling the ACTIVATION planes worth a change through every
producer kernel?"""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")   # development build: tile overrides + the whole experiment table (make -C gridmm_amd/csrc debug)
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops

SHAPES = [(6912, 3072, 768), (6912, 768, 3072), (6912, 768, 768), (6912, 2304, 768), (9472, 6144, 768), (1824, 2304, 768),
          (1824, 3072, 768)]


def tile(p):
    M, K = p.shape
    Mp = (M + 15) // 16 * 16
    q = torch.zeros(Mp, K, dtype=p.dtype, device=p.device)
    q[:M] = p
    return q.view(Mp // 16, 16, K // 32, 32).permute(0, 2, 1, 3).contiguous()


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (M, N, K) in SHAPES:
        x = torch.randn(M, K, device=dev)
        a = ops.split_rows(x)
        ncopy = min(64, max(2, int(700e6 // (N * K * 4)) + 1))
        pws = [ops.PackedLinear(torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)) for _ in range(ncopy)]
        for q in pws:
            q.tiled()
        c = torch.empty(M, N, device=dev)
        res = {}
        outs = {}
        for mode in ("row-major", "tiled W"):      # (the tiled-A probe of round 4 -- -5..-7 % per launch, profiles/r4_tiled_weights.txt --
                                                   # was a kernel flag that round 5 removed with the other experiment residue)
            def call(i):
                q = pws[i % ncopy]
                wh, wl, lay = (q.hi, q.lo, _lib.W_ROWMAJOR) if mode == "row-major" else (q._tiled[0], q._tiled[1], _lib.W_TILED)
                rc = lib.gridmm_linear_planes_map(a.hi.data_ptr(), a.lo.data_ptr(), K, 0, 0, wh.data_ptr(), wl.data_ptr(), q.Kp, lay,
                                                  q.bias.data_ptr(), None, 0, c.data_ptr(), N, None, None, 0, M, N, K, 0, st())
                assert rc == 0, rc
            call(0)
            torch.cuda.synchronize()
            outs[mode] = c.clone()
            for i in range(3):
                call(i)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(40):
                    call(i)
            g.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 40)
            res[mode] = best
        same = torch.equal(outs["row-major"], outs["tiled W"])
        print("%5d x %4d x %4d | " % (M, N, K) + " | ".join("%s %6.1f us" % (k, v) for k, v in res.items()) + " | same bits: %s" % same, flush=True)


if __name__ == "__main__":
    main()
