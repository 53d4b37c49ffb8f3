"""How fast is the vendor library on the SAME matrix work?  The 3-term product as one bf16 GEMM with the contraction
concatenated: [a_hi | a_lo | a_hi] (M x 3K) . [w_hi | w_hi | w_lo]^T (N x 3K)  (fp32 accumulate inside the library;
torch returns bf16 unless out_dtype is available -- timing only).  Device time over hipGraph replays."""
import sys, torch
sys.path.insert(0, ".")
dev = torch.device("cuda:0")
shapes = [(9472, 6144, 768), (6912, 2304, 768), (6912, 3072, 768), (6912, 768, 3072), (6912, 768, 768), (1824, 768, 768), (1824, 2304, 768),
          (1824, 3072, 768), (1824, 768, 3072), (2560, 1536, 768)]
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3
for M, N, K in shapes:
    a = torch.randn(M, 3 * K, device=dev).bfloat16(); w = torch.randn(N, 3 * K, device=dev).bfloat16()
    a1 = torch.randn(M, K, device=dev).bfloat16(); w1 = torch.randn(N, K, device=dev).bfloat16()
    out = {}
    for name, (x, y) in (("3K", (a, w)), ("K", (a1, w1))):
        try:
            us = timed(lambda: torch.mm(x, y.t(), out_dtype=torch.float32))
            kind = "f32 out"
        except Exception:
            us = timed(lambda: torch.mm(x, y.t())); kind = "bf16 out"
        out[name] = (us, kind)
    alg = 2.0 * M * N * K / 1e6
    print("%5d x %5d x %5d | library 3K-concat %7.1f us = %5.0f TF algorithmic (%s) | plain bf16 K %6.1f us = %5.0f TF" % (M, N, K, out["3K"][0], alg / out["3K"][0], out["3K"][1], out["K"][0], alg / out["K"][0]))
