"""Optimizer, parameter groups and schedule of the pre-training loop, on the fused HIP step.

Reference (relative to /root/reference/pretrain_src):
  optim/adamw.py:13-112     AdamW with the HuggingFace "weight decay fix" (eps outside the bias correction, decay after
                            the update, scaled by the raw lr)
  optim/misc.py:12-37       build_optimizer: no weight decay on 'bias', 'LayerNorm.bias', 'LayerNorm.weight'
  optim/sched.py:17-30      warmup_linear / get_lr_sched (lr floor 1e-8)
  train_r2r.py:288-303      clip_grad_norm_(model.parameters(), opts.grad_norm) then optimizer.step()
The clip and the update are two streaming HIP kernels per tensor (gridmm_grad_sumsq, gridmm_adamw_step); the global
norm stays on the device, so a step issues no host synchronisation.
"""
import ctypes
import math

import torch

from . import _lib
from .ops import _p, _stream


def _bump_version(p):
    """The kernel writes the parameter through its raw pointer: advance the tensor's version counter by hand so that
    the packed-weight caches (vilmodel._pack, autograd.WEIGHTS) and autograd's saved-tensor checks see the change."""
    setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if setter is not None:
        setter([p], [p._version + 1])
    else:
        p.add_(0)


class AdamW(torch.optim.Optimizer):
    """adamw.py:13-112 (decay_first=False, the pre-training optimizer) or torch.optim.AdamW ordering
    (decay_first=True, the fine-tune optimizer of agent_base.py:131)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True,
                 decay_first=False):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      correct_bias=correct_bias, decay_first=decay_first))
        self._sumsq = None

    def _live(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    yield group, p

    @torch.no_grad()
    def grad_norm_sq(self):
        """Device scalar: sum of squares of all gradients (fp32 accumulate)."""
        lib = _lib.load()
        acc = None
        for _, p in self._live():
            if acc is None:
                acc = torch.zeros(1, dtype=torch.float32, device=p.device)
            g = p.grad.contiguous()
            _lib.check(lib.gridmm_grad_sumsq(_p(g), g.numel(), 0 if g.dtype == torch.float32 else 1, _p(acc), _stream()),
                       "gridmm_grad_sumsq")
        return acc

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        """max_grad_norm: fuse clip_grad_norm_(all parameters of this optimizer, max_grad_norm) into the update.
        Returns the (pre-clip) gradient norm as a device scalar when clipping, else None."""
        lib = _lib.load()
        sumsq = self.grad_norm_sq() if max_grad_norm is not None else None
        for group, p in self._live():
            if p.dtype not in (torch.float32, torch.float16) or not p.is_contiguous():
                raise ValueError("AdamW: contiguous fp32 / fp16 parameters expected")
            state = self.state[p]
            if len(state) == 0:
                state["step"] = 0
                state["exp_avg"] = torch.zeros_like(p)
                state["exp_avg_sq"] = torch.zeros_like(p)
            state["step"] += 1
            b1, b2 = group["betas"]
            step_size, eps = group["lr"], group["eps"]
            if group["correct_bias"]:
                bc2 = math.sqrt(1.0 - b2 ** state["step"])
                step_size = step_size * bc2 / (1.0 - b1 ** state["step"])
                if group["decay_first"]:
                    eps = eps * bc2       # torch.optim.AdamW: m/bc1 / (sqrt(v/bc2) + eps) == step_size * m / (sqrt(v) + eps*sqrt(bc2))
            g = p.grad.contiguous()
            if g.dtype != p.dtype:
                g = g.to(p.dtype)
            _lib.check(lib.gridmm_adamw_step(
                _p(p), _p(g), _p(state["exp_avg"]), _p(state["exp_avg_sq"]), p.numel(), 0 if p.dtype == torch.float32 else 1,
                float(group["lr"]), float(b1), float(b2), float(eps), float(group["weight_decay"]),
                float(step_size), int(bool(group["decay_first"])), _p(sumsq) if sumsq is not None else ctypes.c_void_p(0),
                float(max_grad_norm or 0.0), _stream()), "gridmm_adamw_step")
            _bump_version(p)
        return None if sumsq is None else sumsq.sqrt()


def build_optimizer(model, opts):
    """optim/misc.py:12-37."""
    named = list(model.named_parameters())
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    groups = [
        {"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": opts.weight_decay},
        {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]
    if opts.optim != "adamw":
        raise ValueError("invalid optimizer %r (the released configs use adamw, config/r2r_pretrain.json:15)" % opts.optim)
    return AdamW(groups, lr=opts.learning_rate, betas=tuple(opts.betas))


def warmup_linear(step, warmup_step, tot_step):
    """optim/sched.py:17-21."""
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def get_lr_sched(global_step, opts):
    """optim/sched.py:24-30."""
    lr = opts.learning_rate * warmup_linear(global_step, opts.warmup_steps, opts.num_train_steps)
    return lr if lr > 0 else 1e-8
