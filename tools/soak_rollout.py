"""Stability soak of the graph-replay rollout: N rollouts over the synthetic environment (all mini-batches several times),
device memory and graph-cache size reported; every 10th rollout is repeated eagerly on the same mini-batch and compared.
usage: python tools/soak_rollout.py [rollouts=60]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gridmm_amd import synthetic as S
from gridmm_amd.agent import GMapNavAgent, default_args
from gridmm_amd.grid_memory import GridMemoryBatch
from gridmm_amd.sim_env import SyntheticNavEnv
from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev, B, T = torch.device("cuda"), 32, 15
torch.manual_seed(0)
model = GlocalTextPathNavCMT(default_config(grid_feat_size=S.BASELINE.feat_dim)).eval().to(dev)
model.varlen_buckets = GlocalTextPathNavCMT.DEFAULT_BUCKETS
mem = GridMemoryBatch(B, S.BASELINE, max_steps=T + 2, device=dev)
env = SyntheticNavEnv(B, mem, n_scans=4, n_episodes=8 * B, seed=3, geom=S.BASELINE, vocab=30000)
env.build_device_store(dev)
agent = GMapNavAgent(default_args(max_action_len=T), env, model, device=dev)
agent.feedback = "argmax"
agent._set_mode(False)
agent.enable_graph_replay()
t0, steps0 = time.perf_counter(), 0
with torch.no_grad():
    for i in range(n):
        ix0 = env.ix
        traj = agent.rollout()
        if i % 10 == 9:
            g, agent._graphs, env.ix = agent._graphs, None, ix0
            again = agent.rollout()
            agent._graphs = g
            assert [t["path"] for t in traj] == [t["path"] for t in again], "graph replay and eager rollouts diverged at rollout %d" % i
            torch.cuda.synchronize()
            print("rollout %3d: graphs %d (captures %d, replays %d), device memory %.2f GB allocated / %.2f GB reserved" % (
                i + 1, len(agent._graphs[1].graphs), agent._graphs[1].captures, agent._graphs[1].replays,
                torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
torch.cuda.synchronize()
print("ok: %d rollouts, %d nav steps in %.1f s" % (n, agent.nav_steps, time.perf_counter() - t0))
