"""Does the row pitch of the operand planes matter (GPU only)?  The planes of a K = 3072 operand are 6144 B apart row to row: a
power-of-two-ish pitch that may map all rows of a k-slice onto a few L2 channels.  Same GEMM with the A planes at pitch
K + pad and the W planes at Kp = K + pad (the C-ABI takes lda and Kp separately from K).
usage: PYTHONPATH=. python tools/bench_gemm_pitch.py"""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")
import ctypes
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops


def planes(x, pitch):
    M, K = x.shape
    hi = torch.zeros(M, pitch, dtype=torch.bfloat16, device=x.device)
    lo = torch.zeros(M, pitch, dtype=torch.bfloat16, device=x.device)
    h = x.to(torch.bfloat16)
    hi[:, :K] = h
    lo[:, :K] = (x - h.float()).to(torch.bfloat16)
    return hi, lo


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (M, N, K, cfgs) in [(6912, 768, 3072, (15, 66, 70)), (6912, 768, 768, (15, 66, 70)), (6912, 3072, 768, (15,)),
                            (9472, 6144, 768, (36,)), (1824, 768, 3072, (13, 75)), (1824, 768, 768, (13, 71))]:
        x = torch.randn(M, K, device=dev)
        b = torch.randn(N, device=dev)
        ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(6)]
        ref = (x.double() @ ws[0].double().t() + b.double()).float()
        for cfg in cfgs:
            line = "%5d x %4d x %4d cfg %2d |" % (M, N, K, cfg)
            for pad_a, pad_w in ((0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (128, 128), (96, 96)):
                ahi, alo = planes(x, K + pad_a)
                pw = [planes(w, K + pad_w) for w in ws]
                c = torch.empty(M, N, device=dev)
                ctr = [0]

                def call():
                    whi, wlo = pw[ctr[0] % len(pw)]
                    ctr[0] += 1
                    rc = lib.gridmm_linear_planes_cfg(ahi.data_ptr(), alo.data_ptr(), K + pad_a, whi.data_ptr(), wlo.data_ptr(), K + pad_w,
                                                      b.data_ptr(), None, 0, c.data_ptr(), N, None, None, 0, M, N, K, 0, cfg, st())
                    assert rc == 0, rc
                call()
                torch.cuda.synchronize()
                err = (c - ref).abs().max().item() / ref.abs().max().item()
                for _ in range(3):
                    call()
                n = 30
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(n):
                        call()
                g.replay()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    g.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / n)
                line += " a+%d w+%d: %6.1f%s |" % (pad_a, pad_w, best, "" if err < 1e-4 else " ERR")
            print(line, flush=True)


if __name__ == "__main__":
    main()
