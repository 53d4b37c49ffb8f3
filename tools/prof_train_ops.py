"""Which python lines of the pre-training step issue the fill / copy / small elementwise launches (torch profiler)."""
import sys, collections
sys.path.insert(0, ".")
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
from gridmm_amd.pretrain_loop import PreTrainer, default_opts
from gridmm_amd.synthetic import batch_to, make_pretrain_batch
from gridmm_amd.vilmodel import default_config
dev = torch.device("cuda:0")
cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=["mlm", "mrc", "sap"], image_prob_size=1000, obj_prob_size=0)
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg).to(dev)
tr = PreTrainer(model, default_opts(warmup_steps=100))
task = sys.argv[1] if len(sys.argv) > 1 else "sap"
batch = batch_to(make_pretrain_batch(np.random.RandomState(0), 32, task, max_steps=5, L=80, vocab=30000, image_prob_size=1000, n_pts=(588 * 3, 588 * 5)), dev)
for _ in range(2):
    tr.train_step(batch, task)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter()
heavy = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        st = [f for f in traceback.extract_stack() if "/gridmm_amd/" in f.filename]
        where = "%s:%d" % (st[-1].filename.split("/")[-1], st[-1].lineno) if st else "(engine)"
        cnt[(name, where)] += 1
        big = [a for a in args if torch.is_tensor(a) and a.is_cuda and a.numel() >= 200000]
        if big and name.split(".")[0] in ("copy_", "mul", "where", "add", "clone", "contiguous", "_to_copy", "masked_fill", "fill_", "zero_", "zeros_like", "native_dropout", "native_dropout_backward", "cat", "gather", "index_select", "slice_backward", "constant_pad_nd", "sum"):
            heavy[(name, where, tuple(big[0].shape), big[0].is_contiguous())] += 1
        return func(*args, **(kwargs or {}))
torch.autograd.set_multithreading_enabled(False)
with Spy():
    tr.train_step(batch, task)
torch.cuda.synchronize()
tot = collections.Counter()
for (n, w), c in cnt.items():
    tot[n] += c
print("ops per step:", sum(tot.values()))
print(tot.most_common(25))
for (n, w), c in cnt.most_common(60):
    print(c, n, w)

print("---- ops on large CUDA tensors (name, site, shape, contiguous): count")
for k, c in heavy.most_common(40):
    print(c, k)
