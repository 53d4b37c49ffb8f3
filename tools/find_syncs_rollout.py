"""Which Python lines of an inference rollout (GMapNavAgent.rollout, argmax feedback, graph replay) make the host wait for the
device?  One read-back per step is inherent (the action); everything else is a candidate.  Same method as
tools/find_syncs_finetune.py (torch's sync-debug mode).  usage (GPU box, repo root): python tools/find_syncs_rollout.py"""
import collections, os, sys, traceback, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gridmm_amd import synthetic as S
from gridmm_amd.agent import GMapNavAgent, default_args
from gridmm_amd.grid_memory import GridMemoryBatch
from gridmm_amd.sim_env import SyntheticNavEnv
from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config

geom, B, T = S.BASELINE, 32, 15
torch.manual_seed(0)
dev = torch.device("cuda")
model = GlocalTextPathNavCMT(default_config(grid_feat_size=geom.feat_dim)).eval().to(dev)
model.varlen_buckets = GlocalTextPathNavCMT.DEFAULT_BUCKETS
mem = GridMemoryBatch(B, geom, max_steps=T + 2, device=dev)
env = SyntheticNavEnv(B, mem, n_scans=4, n_episodes=4 * B, seed=3, geom=geom, vocab=30000)
env.build_device_store(dev)
agent = GMapNavAgent(default_args(max_action_len=T), env, model, device=dev)
agent.feedback = "argmax"
agent._set_mode(False)
agent.enable_graph_replay()
seen = collections.Counter()

def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/gridmm_amd/" in f.filename][-4:]
    seen[" <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st))] += 1

with torch.no_grad():
    for _ in range(8):
        agent.rollout()
    torch.cuda.synchronize()
    warnings.showwarning = show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode(1)
    n0 = agent.nav_steps
    agent.rollout()
    torch.cuda.set_sync_debug_mode(0)
print("synchronising calls in one rollout of %d steps: %d at %d sites" % (agent.nav_steps - n0, sum(seen.values()), len(seen)))
for k, n in seen.most_common():
    print("%4d  %s" % (n, k))
