"""Aggregation launch time (us, hipGraph replays) across instruction lengths: the pipelined paths vs the generic kernel.
usage: PYTHONPATH=. python tools/bench_agg_long.py"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gridmm_amd import ops
from gridmm_amd.grid_memory import pack_reference_lists
from bench_embed import graph_time


def main():
    dev = torch.device("cuda")
    B = 32
    rng = np.random.default_rng(0)
    for D, n in ((512, 7056), (512, 35280), (768, 588 * 5), (768, 588 * 15)):
        fts = [(torch.randn(n, D, device=dev) * 0.5).half() for _ in range(B)]
        maps = [torch.from_numpy(rng.integers(0, 196, size=n)).double().to(dev) for _ in range(B)]
        slab, perm, cs = pack_reference_lists(fts, maps)
        for L in (80, 120, 200):
            frag = ops.text_fragments(torch.randn(B, L, D, device=dev) * 0.3)
            byts = B * (n * D * 2 + n * 4 + 2 * L * D * 2 + 196 * D * 4 + 196)
            flops = 2.0 * B * n * D * L * 2
            res = []
            for force in (0, 1):
                os.environ["GRIDMM_AGG_FORCE_GENERIC"] = str(force)
                us = graph_time(lambda: ops.grid_aggregate(slab, perm, cs, frag, L), n=5, reps=3)
                res.append((us, ops.LAST_AGGREGATE_RC))
            os.environ["GRIDMM_AGG_FORCE_GENERIC"] = "0"
            print("D=%d N=%6d L=%3d | pipelined %7.1f us (rc %d) %5.2f TB/s alg, %5.0f TFLOP/s f16 issued | generic %7.1f us (rc %d)"
                  % (D, n, L, res[0][0], res[0][1], byts / res[0][0] / 1e6, flops / res[0][0] / 1e6, res[1][0], res[1][1]))


if __name__ == "__main__":
    main()
