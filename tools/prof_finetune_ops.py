"""Which python lines of a fine-tune iteration (GMapNavAgent.train, teacher-forced rollout + one backward) issue torch ops on
large CUDA tensors (fills, adds, copies, cats) -- candidates for fusion into the library's kernels."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
from gridmm_amd import synthetic
from gridmm_amd.agent import GMapNavAgent, default_args
from gridmm_amd.grid_memory import GridMemoryBatch
from gridmm_amd.sim_env import SyntheticNavEnv
from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
geom = synthetic.BASELINE
torch.manual_seed(0); np.random.seed(0)
model = GlocalTextPathNavCMT(default_config(grid_feat_size=geom.feat_dim)).cuda()
mem = GridMemoryBatch(32, geom, max_steps=9, device="cuda")
env = SyntheticNavEnv(32, mem, n_scans=4, n_episodes=128, seed=3, geom=geom, vocab=30000)
env.build_device_store("cuda")
agent = GMapNavAgent(default_args(max_action_len=7, train_alg="imitation", lr=1e-5), env, model, device="cuda")
agent.train(2)
torch.cuda.synchronize()
cnt, heavy = collections.Counter(), collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        st = [f for f in traceback.extract_stack() if "/gridmm_amd/" in f.filename]
        where = "%s:%d" % (st[-1].filename.split("/")[-1], st[-1].lineno) if st else "(engine)"
        cnt[name] += 1
        big = [a for a in args if torch.is_tensor(a) and a.is_cuda and a.numel() >= 500000]
        if big and name.split(".")[0] in ("copy_", "mul", "where", "add", "clone", "contiguous", "_to_copy", "masked_fill", "fill_", "zero_", "zeros_like", "zeros", "cat", "gather", "index_select", "slice_backward", "constant_pad_nd", "sum", "index_put_", "index_put", "stack", "div", "index", "select_backward"):
            heavy[(name, where, tuple(big[0].shape))] += 1
        return func(*args, **(kwargs or {}))
torch.autograd.set_multithreading_enabled(False)
with Spy():
    agent.train(1)
torch.cuda.synchronize()
print("ops per iteration:", sum(cnt.values()))
print(cnt.most_common(25))
print("---- ops on large CUDA tensors (name, site, shape): count, by bytes")
for (k, c) in sorted(heavy.items(), key=lambda kv: -kv[1] * int(np.prod(kv[0][2])))[:40]:
    print(c, k, "%.0f MB total" % (c * np.prod(k[2]) * 4 / 1e6))
