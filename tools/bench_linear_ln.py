"""Micro-benchmark: GEMM + residual + LayerNorm as ONE launch (gridmm_linear_planes_ln, rendezvous of a row block's column
tiles) against gridmm_linear_planes followed by gridmm_layernorm, on the step's dense + LayerNorm shapes (GPU only;
device time over hipGraph replays of 40 calls with rotating cold weights)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import ops

SHAPES = [(1824, 768, 768), (1824, 768, 3072), (6912, 768, 768), (6912, 768, 3072)]


def timed(fn, n=40):
    for _ in range(3):
        fn(0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    dev = torch.device("cuda")
    ops.LN_FUSE = True
    for (M, N, K) in SHAPES:
        x = ops.split_rows(torch.randn(M, K, device=dev))
        r = torch.randn(M, N, device=dev)
        gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        ncopy = min(64, max(2, int(700e6 // (N * K * 4)) + 1))
        pws = [ops.PackedLinear(torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)) for _ in range(ncopy)]
        t_gemm = timed(lambda i: ops.linear(x, pws[i % ncopy], residual=r))
        t_two = timed(lambda i: ops.layernorm(ops.linear(x, pws[i % ncopy], residual=r).f32, gamma, beta, 1e-12, want_planes=True))
        t_one = timed(lambda i: ops.linear_ln(x, pws[i % ncopy], gamma, beta, 1e-12, residual=r))
        print("%5d x %4d x %4d | GEMM %6.1f us | GEMM + LayerNorm (2 launches) %6.1f us | fused (1 launch) %6.1f us"
              % (M, N, K, t_gemm, t_two, t_one), flush=True)


if __name__ == "__main__":
    main()
