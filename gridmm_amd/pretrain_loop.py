"""The pre-training step loop around GlocalTextPathCMTPreTraining.

Reference (relative to /root/reference/pretrain_src):
  train_r2r.py:231-327      task-mixed loop: forward(batch, task) -> loss.mean() -> backward -> [DDP all-reduce] ->
                            lr schedule -> clip_grad_norm_(grad_norm) -> optimizer.step() -> zero_grad
  data/loader.py:24-75      MetaLoader: the task of every step is drawn with torch.multinomial over the mix ratios and
                            broadcast from rank 0 so that all ranks train the same task (one int per step)
  utils/misc.py:52-65       wrap_model: DistributedDataParallel(find_unused_parameters=True)
Here: one process per GPU, the DDP wrapper is replaced by gridmm_amd.dist.GradientReducer (bucketed RCCL all-reduce of
the gradients after backward, unused-parameter semantics), clip + AdamW are the fused HIP step of gridmm_amd.optim.
"""
from types import SimpleNamespace

import torch
import torch.distributed as dist

from . import dist as D
from .optim import build_optimizer, get_lr_sched


def default_opts(**over):
    """config/r2r_pretrain.json."""
    o = dict(learning_rate=5e-5, betas=(0.9, 0.98), weight_decay=0.01, grad_norm=5.0, warmup_steps=10000,
             num_train_steps=100000, optim="adamw", gradient_accumulation_steps=1, train_batch_size=32,
             tasks=("mlm", "mrc", "sap"), mix_ratio=(1, 1, 1), seed=0)
    o.update(over)
    return SimpleNamespace(**o)


class TaskSampler:
    """MetaLoader's task draw (data/loader.py:50-58): multinomial over the mix ratios, rank 0's draw wins."""

    def __init__(self, tasks, mix_ratio, device="cpu", seed=0):
        self.tasks = list(tasks)
        self.ratios = torch.tensor(list(mix_ratio), dtype=torch.float32)
        self.gen = torch.Generator().manual_seed(seed)
        self.device = device

    def next_task(self):
        tid = torch.multinomial(self.ratios, 1, generator=self.gen)
        if D.is_dist():
            t = tid.to(self.device)
            dist.broadcast(t, 0)
            tid = t.cpu()
        return self.tasks[int(tid.item())]


class PreTrainer:
    def __init__(self, model, opts, reducer_kw=None):
        """reducer_kw: GradientReducer options (bucket_mb, overlap, algo = ring | rsag | direct | auto, payload = fp32 | bf16)."""
        self.model, self.opts = model, opts
        D.broadcast_parameters(model.parameters())            # DDP construction broadcasts rank 0's weights
        self.optimizer = build_optimizer(model, opts)
        self.reducer = D.GradientReducer(model.parameters(), **(reducer_kw or {}))
        self.global_step = 0
        self._micro = 0
        self.exchange = True      # False: skip the gradient exchange (bench.py times the step with and without it)
        self.optimizer.zero_grad()

    def train_step(self, batch, task):
        """One micro-step of train_r2r.py:233-303.  Returns (per-sample loss vector, grad norm or None)."""
        o = self.opts
        self.model.train()
        losses = self.model(batch, task=task, compute_loss=True)
        loss = losses.mean()
        if o.gradient_accumulation_steps > 1:
            loss = loss / o.gradient_accumulation_steps
        last = (self._micro + 1) % o.gradient_accumulation_steps == 0
        self.reducer.enabled = self.exchange
        if self.exchange:
            self.reducer.expect(task if o.gradient_accumulation_steps == 1 else None, final=last)
        loss.backward()                                        # the reducer's hooks launch the exchange from in here
        self._micro += 1
        norm = None
        if self._micro % o.gradient_accumulation_steps == 0:
            self.global_step += 1
            lr = get_lr_sched(self.global_step, o)
            for g in self.optimizer.param_groups:
                g["lr"] = lr
            if self.exchange:
                self.reducer.reduce()                          # the DDP exchange (no-op on one rank)
            norm = self.optimizer.step(max_grad_norm=o.grad_norm if o.grad_norm != -1 else None)
            self.optimizer.zero_grad()
        return losses.detach(), norm
