"""Host-side cProfile of three fine-tune iterations (GMapNavAgent.train) after warm-up: which Python functions the\nteacher-forced rollout (forward under autograd) spends its time in.  usage (repo root, GPU box): python tools/prof_finetune_host.py"""
import cProfile, pstats, sys, os
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
agent = GMapNavAgent(default_args(max_action_len=7, train_alg="imitation", lr=1e-5), env, model, device="cuda")
agent.train(4); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
agent.train(3); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
