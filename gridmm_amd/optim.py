"""Optimizer, parameter groups and schedule of the pre-training loop, on the fused HIP step.

Reference (relative to /root/reference/pretrain_src):
  optim/adamw.py:13-112     AdamW with the HuggingFace "weight decay fix" (eps outside the bias correction, decay after
                            the update, scaled by the raw lr)
  optim/misc.py:12-37       build_optimizer: no weight decay on 'bias', 'LayerNorm.bias', 'LayerNorm.weight'
  optim/sched.py:17-30      warmup_linear / get_lr_sched (lr floor 1e-8)
  train_r2r.py:288-303      clip_grad_norm_(model.parameters(), opts.grad_norm) then optimizer.step()
The clip and the update are two multi-tensor HIP launches over all live parameters (gridmm_multi_grad_sumsq,
gridmm_multi_adamw_step; fp32 tensors and the fp16 grid_proj in the same table); the global norm stays on the device, so a
step issues no host synchronisation.
"""
import ctypes
import math

import numpy as np
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

    # ---- multi-tensor launch tables (fp32 tensors): one record per parameter, rebuilt every step on the host
    # (gradient tensors are re-allocated by zero_grad) and shipped with ONE small H2D copy
    _REC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"),
                     ("lr", "<f4"), ("step_size", "<f4"), ("eps", "<f4"), ("wd", "<f4"), ("dtype", "<i4"), ("pad", "<i4")])
    _PREC = np.dtype([("hi", "<u8"), ("lo", "<u8"), ("thi", "<u8"), ("tlo", "<u8"), ("N", "<i4"), ("K", "<i4"), ("ldw", "<i4"),
                      ("ldt", "<i4")])       # csrc/optim.hip PlaneDesc: optimizer-owned weight planes
    _CHUNK = 16384

    def _tables(self, items, dev, graph_tabs=None, key=None, ring_min=1):
        """items: list of (p, g, exp_avg, exp_avg_sq, lr, step_size, eps, wd) for contiguous fp32 / fp16 tensors.
        graph_tabs (captured steps): the table lives in a pinned host pool mirrored by a device pool, both allocated
        BEFORE the capture; nothing is copied inside the graph -- refresh_graph_tables() rewrites lr / step_size / eps in
        the pinned copy and uploads the pool on the replay's stream before every replay.  (An H2D copy node inside the
        graph was tried first: replayed, it raced with the kernels reading the table -- wild pointers at full size.)"""
        from . import autograd as ag
        rec = np.zeros(len(items), self._REC)
        first = np.zeros(len(items) + 1, np.int32)
        prec = np.zeros(len(items), self._PREC)
        owned = []
        for i, (p, g, m, v, lr, ss, eps, wd) in enumerate(items):
            rec[i] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, ss, eps, wd,
                      int(p.dtype == torch.float16), 0)
            first[i + 1] = first[i] + -(-p.numel() // self._CHUNK)
            pl = ag.WEIGHTS.optimizer_planes(p) if g.dtype == torch.float32 else None
            if pl is not None:      # the update writes this weight's bf16 planes in both orientations itself
                prec[i] = (pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), pl[3].data_ptr(), pl[4], pl[5], pl[6], pl[7])
                owned.append(p)
        pad = (-(rec.nbytes + first.nbytes)) % 16
        blob = np.concatenate([rec.view(np.uint8), first.view(np.uint8), np.zeros(pad, np.uint8), prec.view(np.uint8)])
        self._last_planes = (rec.nbytes + first.nbytes + pad) if owned else None       # byte offset of the plane records
        self._last_owned = owned
        if graph_tabs is None:
            # through pinned memory (a ring: a slot is rewritten only after the copy issued from it has completed), so that
            # the upload is a plain asynchronous DMA and never a staged pageable copy (measured host-synchronous on this
            # ROCm stack, tools/repro_pageable_h2d.py -- correct either way, but it stalls the launching thread)
            # The DEVICE table is allocated per call (stream-ordered by the caching allocator and kept alive by
            # self._keepalive): step() builds the tables of ALL (betas, decay order) classes before it launches any kernel
            # -- the clip norm spans every class -- so a device ring shorter than the class count would be overwritten
            # under kernels that have not run yet; the pinned ring holds at least one slot per class too.
            pinned = self._pinned_slot(len(blob), ring_min)
            pinned.numpy()[:] = blob
            d = torch.empty(len(blob), dtype=torch.uint8, device=dev)
            d.copy_(pinned, non_blocking=True)
            self._ring[self._ring_pos][1].record()
        else:
            pinned, d = self._carve(graph_tabs, len(blob))
            pinned.numpy()[:] = blob
            graph_tabs["multi"][key] = (pinned, d, rec.nbytes, [it[0] for it in items])
            graph_tabs.setdefault("owned", []).extend(owned)
        return d, rec.nbytes, int(first[-1]), self._last_planes, owned

    _RING = 4

    def _pinned_slot(self, nbytes, ring_min=1):
        """Pinned host bytes from a ring of max(_RING, ring_min + 1) slots: a slot is reused only after the copy issued from it
        has completed (the event recorded behind it), so the host never rewrites a table the device has not fetched yet."""
        ring = self.__dict__.setdefault("_ring", [])
        size = max(self._RING, ring_min + 1)
        if len(ring) < size:
            ring.append([torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory(), torch.cuda.Event()])
            self._ring_pos = len(ring) - 1
        else:
            self._ring_pos = (self._ring_pos + 1) % len(ring)
            ring[self._ring_pos][1].synchronize()
        slot = ring[self._ring_pos]
        if slot[0].numel() < nbytes:
            slot[0] = torch.empty(2 * nbytes, dtype=torch.uint8).pin_memory()
        return slot[0][:nbytes]

    @staticmethod
    def _carve(graph_tabs, nbytes):
        """Matching slices of the pinned pool and of its device mirror (both allocated by the caller BEFORE the capture)."""
        o = (graph_tabs["used"] + 63) // 64 * 64
        if o + nbytes > graph_tabs["pool"].numel():
            raise RuntimeError("graph_tabs: pinned pool too small")
        graph_tabs["used"] = o + nbytes
        return graph_tabs["pool"][o:o + nbytes], graph_tabs["dev_pool"][o:o + nbytes]

    def _step_scalars(self, group, state):
        b1, b2 = group["betas"]
        step_size, eps = group["lr"], group["eps"]
        if group["correct_bias"]:
            bc2 = math.sqrt(1.0 - b2 ** state["step"])
            step_size = step_size * bc2 / (1.0 - b1 ** state["step"])
            if group["decay_first"]:
                eps = eps * bc2       # torch.optim.AdamW: m/bc1 / (sqrt(v/bc2) + eps) == step_size * m / (sqrt(v) + eps*sqrt(bc2))
        return float(group["lr"]), float(step_size), float(eps)

    def refresh_graph_tables(self, graph_tabs, advance=True):
        """Before a replay of a captured step: the next step's lr / bias-corrected step size / eps of every parameter
        of that graph go into its pinned tables (the graph's own copy nodes upload them)."""
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        for pinned, _, rec_bytes, params in graph_tabs["multi"].values():
            rec = pinned.numpy()[:rec_bytes].view(self._REC)
            for i, p in enumerate(params):
                st = self.state[p]
                if advance:
                    st["step"] += 1
                rec["lr"][i], rec["step_size"][i], rec["eps"][i] = self._step_scalars(group_of[id(p)], st)
        n = graph_tabs["used"]
        graph_tabs["dev_pool"][:n].copy_(graph_tabs["pool"][:n], non_blocking=True)      # ordered before the replay

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None, graph_tabs=None):
        """max_grad_norm: fuse clip_grad_norm_(all parameters of this optimizer, max_grad_norm) into the update.
        Returns the (pre-clip) gradient norm as a device scalar when clipping, else None.
        graph_tabs: dict(multi={}, pool=<pinned uint8>, dev_pool=<device uint8>, used=0) filled during the
        capture of a training step (train_graph.py)."""
        lib = _lib.load()
        multi = []
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
            _, step_size, eps = self._step_scalars(group, state)
            g = p.grad.contiguous()
            if g.dtype != p.dtype:
                g = g.to(p.dtype)
            multi.append((p, g, state["exp_avg"], state["exp_avg_sq"], float(group["lr"]), float(step_size), float(eps),
                          float(group["weight_decay"]), float(b1), float(b2), int(bool(group["decay_first"]))))
        if not multi:
            return None
        dev = multi[0][0].device
        # all tensors of one (betas, decay order) class go into one launch -- fp32 and the fp16 grid_proj alike (a record
        # carries its dtype); the released configs have one class
        classes = {}
        for it in multi:
            classes.setdefault(it[8:], []).append(it[:8])
        tables = {k: self._tables(v, dev, graph_tabs, k, ring_min=len(classes)) for k, v in classes.items()}
        sumsq = None
        if max_grad_norm is not None:
            sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
            part = torch.empty(max(t[2] for t in tables.values()), dtype=torch.float32, device=dev)
            tmp = torch.empty(1, dtype=torch.float32, device=dev)
            for (blob, rec_bytes, n_chunks, _po, _ow), items in zip(tables.values(), classes.values()):
                _lib.check(lib.gridmm_multi_grad_sumsq(_p(blob), ctypes.c_void_p(blob.data_ptr() + rec_bytes), len(items),
                                                       n_chunks, _p(part), _p(tmp), _stream()), "gridmm_multi_grad_sumsq")
                sumsq += tmp
        for (b1, b2, df), items in classes.items():
            blob, rec_bytes, n_chunks, plane_off, owned = tables[(b1, b2, df)]
            _lib.check(lib.gridmm_multi_adamw_step(_p(blob), ctypes.c_void_p(blob.data_ptr() + rec_bytes), len(items),
                                                   n_chunks, b1, b2, df, _p(sumsq) if sumsq is not None else ctypes.c_void_p(0),
                                                   float(max_grad_norm or 0.0),
                                                   ctypes.c_void_p(blob.data_ptr() + plane_off) if plane_off is not None else ctypes.c_void_p(0),
                                                   _stream()), "gridmm_multi_adamw_step")
        from . import autograd as ag
        for it in multi:
            _bump_version(it[0])
        for _b, _r, _n, _po, owned in tables.values():
            for p in owned:
                ag.WEIGHTS.mark_written(p)            # its planes are current for the new version: no pack launch follows
        self._keepalive = (tables, multi)              # device tables / cast gradients must outlive the async launches
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
