"""What separates two replays of the captured headline step: the graph launch itself, or the per-step host inputs (pose floats +
fused-logit index maps: one pinned H2D in front of every replay)?   usage (GPU box): python tools/exp_replay_gap.py"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

a = argparse.Namespace(batch=32, shape="baseline", mem_steps=1, eager=False)
dev = torch.device("cuda:0")
model, batch, mem, eps, step, eager_step, geom = bench.build_workload(a, dev)
g = step.graph


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print("full step (index maps + pose H2D + replay)     %.4f ms" % timed(step))
print("replay only                                     %.4f ms" % timed(g.graph.replay))
poses = [(e[0]["x"], e[0]["y"]) for e in eps]
heads = [e[0]["heading"] for e in eps]


def pose_replay():
    mem.set_pose(poses, heads)
    g.graph.replay()


print("pose H2D + replay (no index-map rebuild)        %.4f ms" % timed(pose_replay))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
g.graph.replay()
e1.record()
torch.cuda.synchronize()
print("one replay between two events                   %.4f ms" % e0.elapsed_time(e1))
