"""Fine-tune iteration (BASELINE.json config 2 read as training): GMapNavAgent.train on the synthetic environment --
teacher-forced rollout of B episodes (language + panorama + fill_gridmap + navigation per step, with autograd), one
backward through every step, clip 40, AdamW.  Secondary measurement; the headline metric is bench.py's inference step.
usage: PYTHONPATH=. python tools/bench_finetune.py [--batch 32] [--shape baseline|native] [--iters 4]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--shape", default="baseline", choices=["baseline", "native"])
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--max-action-len", type=int, default=7)
    ap.add_argument("--host-store", action="store_true", help="observations assembled on the host per step (round 1-2 form)")
    a = ap.parse_args()
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.sim_env import SyntheticNavEnv
    from gridmm_amd import synthetic
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    geom = synthetic.BASELINE if a.shape == "baseline" else synthetic.NATIVE
    torch.manual_seed(0)
    np.random.seed(0)
    model = GlocalTextPathNavCMT(default_config(grid_feat_size=geom.feat_dim)).cuda()
    mem = GridMemoryBatch(a.batch, geom, max_steps=a.max_action_len + 2, device="cuda")
    env = SyntheticNavEnv(a.batch, mem, n_scans=4, n_episodes=4 * a.batch, seed=3, geom=geom, vocab=30000)
    if not a.host_store:
        env.build_device_store("cuda")        # observations resident in HBM: an env step moves no feature bytes over PCIe
    agent = GMapNavAgent(default_args(max_action_len=a.max_action_len, train_alg="imitation", lr=1e-5), env, model,
                         device="cuda")
    agent.train(max(a.warmup, 4))             # also fills the environment's feature memo (4 passes over the episode list)
    torch.cuda.synchronize()
    steps0 = getattr(agent, "nav_steps", None)
    t0 = time.perf_counter()
    losses = agent.train(a.iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"iters": a.iters, "batch": a.batch, "shape": a.shape, "s_per_iter": round(dt / a.iters, 3),
           "episodes_per_s": round(a.batch * a.iters / dt, 1), "loss": [round(float(x), 4) for x in losses]}
    if steps0 is not None:
        out["nav_steps_per_s"] = round((agent.nav_steps - steps0) * a.batch / dt, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
