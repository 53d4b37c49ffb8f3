"""Experiment: the B = 32 step as S independent episode groups, one hipGraph each, replayed on S streams at once
(episodes are independent: a scheduling choice, same work).  usage: PYTHONPATH=. python tools/bench_two_streams.py [S ...]"""
import argparse
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    groups = [int(x) for x in sys.argv[1:]] or [1, 2, 4]
    dev = torch.device("cuda")
    for S in groups:
        args = argparse.Namespace(batch=32 // S, shape="baseline", mem_steps=1, eager=False)
        streams = [torch.cuda.Stream() for _ in range(S)]
        steps = []
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                model, batch, mem, eps, step, eager_step, geom = bench.build_workload(args, dev, device_feats=True)
                steps.append((step, model, batch, mem))
        torch.cuda.synchronize()

        def run():
            for s in range(S):
                with torch.cuda.stream(streams[s]):
                    steps[s][0]()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("groups=%d (B=%d each): %.3f ms per 32-episode step  -> %.0f steps/s" % (S, 32 // S, 1e3 * dt, 32 / dt))
        del steps
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
