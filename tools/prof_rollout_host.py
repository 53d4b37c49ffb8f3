"""Host-side cost of GMapNavAgent.rollout without a GPU: the synthetic environment + a stub model that returns random
embeddings / logits of the right shapes, so that only the Python / numpy / tensor-assembly sections cost time.
usage: python tools/prof_rollout_host.py [--batch 32] [--profile]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


class StubMem:
    slab = True

    def reset(self):
        pass

    def step(self, *a):
        pass


class StubStore:
    def __init__(self, env):
        self.env = env

    def append(self, mem, keys):
        return None, [(0.0, 0.0)] * len(keys)


class StubModel:
    def __init__(self, H=768):
        self.H = H
        self.g = torch.Generator().manual_seed(0)

    def __call__(self, mode, b):
        if mode == "language":
            return torch.randn(*b["txt_ids"].shape, self.H, generator=self.g)
        if mode == "panorama":
            B, V = b["view_img_fts"].shape[:2]
            m = torch.arange(V)[None] < b["view_lens"][:, None]
            return torch.randn(B, V, self.H, generator=self.g), m
        gm = b["gmap_masks"] & ~b["gmap_visited_masks"]
        gl = torch.randn(gm.shape, generator=self.g).masked_fill(~gm, -float("inf"))
        ll = torch.randn(b["vp_nav_masks"].shape, generator=self.g).masked_fill(~b["vp_nav_masks"], -float("inf"))
        return {"global_logits": gl, "local_logits": ll, "fused_logits": gl, "grid_logits": gl}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rollouts", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    from gridmm_amd import synthetic as S
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.sim_env import SyntheticNavEnv
    env = SyntheticNavEnv(a.batch, StubMem(), n_scans=4, n_episodes=4 * a.batch, seed=3, geom=S.BASELINE, vocab=30000)
    env.device_store = StubStore(env)
    agent = GMapNavAgent(default_args(max_action_len=15), env, StubModel(), device="cpu")
    agent.feedback = "argmax"
    torch.cuda.synchronize = lambda: None
    with torch.no_grad():
        agent.rollout()
        n0, t0 = agent.nav_steps, time.perf_counter()
        pr = cProfile.Profile() if a.profile else None
        if pr:
            pr.enable()
        for _ in range(a.rollouts):
            agent.rollout()
        if pr:
            pr.disable()
        dt, steps = time.perf_counter() - t0, agent.nav_steps - n0
        agent.timers = {}
        n1 = agent.nav_steps
        agent.rollout()
        ps = agent.nav_steps - n1
    print("host ms per step: %.2f  (%d steps)" % (1e3 * dt / steps, steps))
    for k, v in sorted(agent.timers.items(), key=lambda kv: -kv[1]):
        print("  %-55s %.2f ms" % (k, 1e3 * v / ps))
    if pr:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
