"""Device time of the fused embedding / head kernels of the nav step, replayed from a hipGraph (us per launch).
usage: PYTHONPATH=. python tools/bench_embed.py"""
import os
import sys
import numpy as np
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gridmm_amd import ops


def graph_time(fn, n=20, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (n * reps)


def main():
    dev = torch.device("cuda")
    B, H, G, V, L = 32, 768, 20, 37, 80
    S = 196 + G
    proj = torch.randn(B, 196, H, device=dev)
    pos = torch.randn(B, 196, 5, device=dev)
    occ = torch.ones(B, 196, dtype=torch.uint8, device=dev)
    lin, ln = nn.Linear(5, H).to(dev), nn.LayerNorm(H, eps=1e-12).to(dev)
    out = torch.empty(B, S, H, device=dev)
    masks = torch.zeros(B, S + L, dtype=torch.uint8, device=dev)
    gm = torch.ones(B, G, dtype=torch.uint8, device=dev)
    wT = ops.linear_wt(lin)
    with torch.no_grad():
        for flags in [0] + [int(x) for x in os.environ.get("EMBED_FLAGS", "").split(",") if x]:
            os.environ["GRIDMM_EMBED_DEBUG"] = str(flags)
            print("cells_embed flags=%d: %.1f us" % (flags, graph_time(lambda: ops.cells_embed(proj, pos, lin, ln, occ, out, masks, tail_mask=gm, wT=wT))))
        pe = ln(pos @ lin.weight.t() + lin.bias).contiguous()
        m2 = torch.zeros(B, S, dtype=torch.uint8, device=dev)
        print("cells_compact (old): %.1f us" % graph_time(lambda: ops.cells_compact(proj, pe, occ, out, m2)))
        x = torch.randn(B * 196, H, device=dev)
        print("layernorm 6272 rows: %.1f us" % graph_time(lambda: ops.layernorm(x, ln.weight, ln.bias, 1e-12)))
        print("copy_rows 6272 rows: %.1f us" % graph_time(lambda: ops.copy_rows(proj, out, 0)))


if __name__ == "__main__":
    main()
