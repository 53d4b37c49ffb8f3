"""bench.py's `rollout` leg alone (GMapNavAgent.rollout end to end on the synthetic environment), with the batched
collation on and off.   usage: python tools/bench_rollout.py [--batch 32]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--slow-too", action="store_true")
    ap.add_argument("--profile", action="store_true", help="cProfile of the host side of the timed rollouts")
    ap.add_argument("--nosync-sections", action="store_true", help="section timers without device synchronisation: where the HOST spends a step")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.nosync_sections:
        from gridmm_amd.agent import GMapNavAgent
        GMapNavAgent.timers_sync = False
    if a.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        r = bench.rollout_leg(a, dev, profiler=pr)
        print(json.dumps(r, indent=1))
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
        pstats.Stats(pr).sort_stats("tottime").print_stats(30)
        return
    r = bench.rollout_leg(a, dev)
    print(json.dumps(r, indent=1))
    if a.slow_too:
        from gridmm_amd.agent import GMapNavAgent
        init = GMapNavAgent.__init__

        def slow_init(self, *x, **k):
            init(self, *x, **k)
            self.fast_collate = False
        GMapNavAgent.__init__ = slow_init
        print(json.dumps(bench.rollout_leg(a, dev), indent=1))


if __name__ == "__main__":
    main()
