"""cProfile of the HOST side of the headline step (index maps, pose upload, graph replay) over 200 steps: the device needs ~2 ms per
step, whatever the host needs beyond that is lost.   usage (GPU box): python tools/prof_headline_host.py"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

a = argparse.Namespace(batch=32, shape="baseline", mem_steps=1, eager=False)
model, batch, mem, eps, step, eager_step, geom = bench.build_workload(a, torch.device("cuda:0"))
for _ in range(20):
    step()
torch.cuda.synchronize()
# host-only cost: time the enqueue loop without waiting for the device (the stream queue absorbs 50 steps)
t0 = time.perf_counter()
for _ in range(50):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms per step; with the device %.3f ms per step" % (1e3 * (t1 - t0) / 50, 1e3 * (t2 - t0) / 50))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
