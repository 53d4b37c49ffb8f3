"""Fine-tune iteration: where the host time that is not kernel launches goes (pinned allocations of the growing upload rings,
garbage collections inside the autograd-heavy loop).  usage (GPU box): python tools/exp_finetune_host.py"""
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

PINS = [0]
_pin = torch.Tensor.pin_memory


def counted_pin(self, *a, **k):
    PINS[0] += 1
    return _pin(self, *a, **k)


torch.Tensor.pin_memory = counted_pin


def build():
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.sim_env import SyntheticNavEnv
    from gridmm_amd import synthetic
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    geom = synthetic.BASELINE
    torch.manual_seed(0)
    np.random.seed(0)
    model = GlocalTextPathNavCMT(default_config(grid_feat_size=geom.feat_dim)).cuda()
    mem = GridMemoryBatch(32, geom, max_steps=9, device="cuda")
    env = SyntheticNavEnv(32, mem, n_scans=4, n_episodes=128, seed=3, geom=geom, vocab=30000)
    env.build_device_store("cuda")
    return GMapNavAgent(default_args(max_action_len=7, train_alg="imitation", lr=1e-5), env, model, device="cuda")


def timed(agent, iters, label):
    torch.cuda.synchronize()
    p0, g0 = PINS[0], [s["collections"] for s in gc.get_stats()]
    t0 = time.perf_counter()
    agent.train(iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    g1 = [s["collections"] for s in gc.get_stats()]
    print(json.dumps({"leg": label, "s_per_iter": round(dt, 4), "pin_memory_calls": PINS[0] - p0,
                      "gc_collections": [b - a for a, b in zip(g0, g1)]}), flush=True)


agent = build()
agent.train(4)
timed(agent, 4, "iterations 5-8 (what bench.py's leg times after its 4 warm-up iterations)")
timed(agent, 4, "iterations 9-12")
timed(agent, 4, "iterations 13-16")
gc.collect()
gc.freeze()
gc.disable()
timed(agent, 4, "gc disabled (frozen heap)")
timed(agent, 4, "gc disabled, again")
gc.enable()
timed(agent, 4, "gc enabled again")
