"""Which Python lines of a fine-tune iteration (GMapNavAgent.train) make the host wait for the device?  torch's sync-debug
mode warns at every synchronising call (.cpu(), .item(), pageable H2D copies, nonzero ...); each warning is printed once with
the innermost frames of this repo.  usage (GPU box, repo root): python tools/find_syncs_finetune.py"""
import collections, os, sys, traceback, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gridmm_amd.agent import GMapNavAgent, default_args
from gridmm_amd.grid_memory import GridMemoryBatch
from gridmm_amd.sim_env import SyntheticNavEnv
from gridmm_amd import synthetic
from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config

geom = synthetic.BASELINE
torch.manual_seed(0); np.random.seed(0)
model = GlocalTextPathNavCMT(default_config(grid_feat_size=geom.feat_dim)).cuda()
mem = GridMemoryBatch(32, geom, max_steps=9, device="cuda")
env = SyntheticNavEnv(32, mem, n_scans=4, n_episodes=128, seed=3, geom=geom, vocab=30000)
env.build_device_store("cuda")
alg = sys.argv[1] if len(sys.argv) > 1 else "imitation"
agent = GMapNavAgent(default_args(max_action_len=7, train_alg=alg, lr=1e-5), env, model, device="cuda")
agent.train(3); torch.cuda.synchronize()

seen = collections.Counter()
def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/gridmm_amd/" in f.filename or "/tools/" in f.filename][-4:]
    key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st))
    seen[key] += 1
warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
agent.train(1)
torch.cuda.set_sync_debug_mode(0)
torch.cuda.synchronize()
print("synchronising calls in one iteration: %d at %d sites" % (sum(seen.values()), len(seen)))
for k, n in seen.most_common():
    print("%4d  %s" % (n, k))
