"""A pre-training step (forward + losses + backward + gradient clip + AdamW) as ONE hipGraph.

The eager step (pretrain_loop.PreTrainer.train_step) is ~2000 launches issued from Python: two thirds of its wall
time is host work.  Captured, the same launches replay from the device's command processor.  What a capture cannot
contain, and where it goes instead:

  host decisions of the forward        recorded once by an ordinary eager step (hostsync.record), replayed as constants
  (shapes from lengths, index tensors  during the capture: a graph therefore belongs to ONE batch metadata signature
  from vpid lists, x[mask] gathers)    (lengths, vpid lists, masked positions); the batch's DATA tensors are static inputs
  attention-dropout seeds              frozen kernel arguments + a device seed word the kernels read (autograd.SEED_DEV),
                                       uploaded before every replay; torch.dropout uses torch's graph-safe generator
  lr schedule, AdamW bias correction   per-parameter lr / step_size / eps live in pinned host tables mirrored on the
                                       device; AdamW.refresh_graph_tables rewrites and uploads them before a replay
                                       (no copy / memset NODES in the graph: both misbehaved when replayed)
  packed-weight caches                 the re-split of every weight is part of the captured step; versions are bumped
                                       after a replay so that eager users of the caches re-pack

ROCm 7.2 note (measured, tools/dbg_graph_train3.py): with the runtime's pre-recorded graph packets (the default) the
replay of this graph faults on the device (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION at the third replay of the
full-size step; it needs dropout on, a non-zero lr and the fp16 grid_proj tensors in the captured update -- through their
own launches or through the multi-tensor table alike -- and goes away with unrelated eager work between replays), and
the memset nodes of this graph did not clear their destinations in time (csrc/common.h);
with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment BEFORE the HIP runtime starts, the same graph replays
correctly.  GraphedTrainStep refuses to run without it (GRIDMM_TRAIN_GRAPH_ANY_RUNTIME=1 overrides the check, for
re-testing a newer runtime); bench.py and the tests run this leg in a subprocess that sets it
(the navigation-step graph of the headline is unaffected and keeps the default).

Several ranks (torch.distributed initialised): the step is SEVERAL graphs -- forward + loss, then one graph per segment
of the backward (hostsync.boundary marks in the model) -- so that the gradient exchange overlaps the backward as DDP's
does (pretrain_src/utils/misc.py:52-65): the reducer's hooks, in capture mode, copy every finished gradient into its
bucket slot inside the graph; after launching segment k the host tells the reducer which parameters it finished and
every complete bucket's RCCL exchange is enqueued on the side stream, behind segment k and beside segments k+1...; after
the last segment: reduce() (control flag, join), then clip + AdamW launched eagerly on the reduced gradients (two
multi-tensor launches).  The batch tensors recorded on the host tape (the packed grid features, the label counts of the sap stop
re-weighting) are FROZEN at record time: a graph belongs to one batch, in-place edits of those inputs are not seen.

Reference loop: pretrain_src/train_r2r.py:231-303 (gradient_accumulation_steps == 1).  A training loop over
real data would keep one graph per (task, padded-shape bucket) and feed index tensors as inputs; this class covers the
fixed-metadata case (bench.py's train leg, tests/test_hip_train_graph.py).
"""
import os

import torch

from . import autograd as ag, dist as D, hostsync as hs
from .optim import _bump_version, get_lr_sched


RUNTIME_ENV = ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


class GraphedTrainStep:
    def __init__(self, trainer, batch, task, warmup=1, capture_optimizer=True, segments=None):
        if os.environ.get(RUNTIME_ENV[0]) != RUNTIME_ENV[1] and not os.environ.get("GRIDMM_TRAIN_GRAPH_ANY_RUNTIME"):
            raise RuntimeError("GraphedTrainStep needs %s=%s in the environment before torch / HIP start (see the module "
                               "docstring): replaying this graph with pre-recorded packets faults on ROCm 7.2" % RUNTIME_ENV)
        o = trainer.opts
        if o.gradient_accumulation_steps != 1:
            raise ValueError("GraphedTrainStep: gradient_accumulation_steps == 1")
        self.dist = D.is_dist()
        self.segmented = self.dist if segments is None else bool(segments)   # backward as one graph per segment
        if self.dist or self.segmented:
            capture_optimizer = False                       # exchange between the captured backward and the update
        self.tr, self.batch, self.task = trainer, batch, task
        model, opt = trainer.model, trainer.optimizer
        dev = next(model.parameters()).device
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.seed_host = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.tabs = dict(multi={}, pool=torch.empty(1 << 20, dtype=torch.uint8).pin_memory(),
                         dev_pool=torch.zeros(1 << 20, dtype=torch.uint8, device=dev), used=0)
        self._done = None
        prev = ag.SEED_DEV
        ag.SEED_DEV = self.seed_dev
        try:
            with hs.record() as tape:                       # an ordinary (training) step that tapes its host decisions
                trainer.train_step(batch, task)
            self.tape = tape
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                   # the same step from the tape: allocator pools, weight caches
                for _ in range(warmup):
                    with hs.replay(tape):
                        trainer.train_step(batch, task)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            opt.zero_grad(set_to_none=True)
            model.train()
            trainer.reducer.enabled = False                 # no host bookkeeping / collectives inside the capture
            if self.segmented:
                self._capture_segments(model, batch, task, tape)
                return
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                with hs.replay(tape):
                    losses = model(batch, task=task, compute_loss=True)
                    losses.mean().backward()
                    self.norm = None
                    if capture_optimizer:
                        self.norm = opt.step(max_grad_norm=o.grad_norm if o.grad_norm != -1 else None, graph_tabs=self.tabs)
                self.losses = losses.detach()
            # the gradient buffers of the capture belong to the graph from here on: an eager step that found them in
            # p.grad would ACCUMULATE into them (train_step clears the gradients at the end of a step, a capture computes
            # nothing)
            self.grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}   # static buffers (inspection)
            self._grad_of = [(p, p.grad) for p in model.parameters() if p.grad is not None]
            opt.zero_grad(set_to_none=True)
        finally:
            ag.SEED_DEV = prev
            trainer.reducer.enabled = True
        self.capture_optimizer = capture_optimizer
        self.params = [p for tab in self.tabs["multi"].values() for p in tab[3]]
        for p in self.params:                               # the capture pass counted a step that never ran
            opt.state[p]["step"] -= 1
        if not capture_optimizer:
            self.params = [p for p, _ in self._grad_of]

    def _capture_segments(self, model, batch, task, tape):
        """Several ranks: forward + loss as one graph, the backward as one graph PER SEGMENT of the autograd graph (the
        model's hostsync.boundary marks: local encoder + heads | grid encoders | panorama encoder + upper text layers |
        middle text layers | lower text layers + embeddings).  The reducer's hooks run in capture mode: a finished
        gradient is copied into its bucket slot by a captured kernel.  At replay time the host launches segment k and
        then tells the reducer which parameters that segment finished (mark_ready): every complete bucket leaves on the
        side stream while the next segments are still in backward (pretrain_src/utils/misc.py:52-65: DDP's overlap of
        the bucket all-reduces with backward)."""
        red = self.tr.reducer
        self.graphs, logs = [], []
        hs.CUTS = cuts = []
        g0 = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g0):
                with hs.replay(tape):
                    losses = model(batch, task=task, compute_loss=True)
                    loss = losses.mean()
                self.losses = losses.detach()
        finally:
            hs.CUTS = None
        self.graphs.append(g0)
        pool = g0.pool()
        import contextlib

        @contextlib.contextmanager
        def segment(k):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                yield
            self.graphs.append(g)
            logs.append(red.take_capture_log())
        red.begin_capture()
        try:
            hs.segmented_backward(loss, cuts, segment)
        finally:
            red.end_capture()
        del cuts, loss
        last = {i: k for k, log in enumerate(logs) for i in log}        # a tied weight is final in its LAST segment
        self.seg_final = [[i for i, kk in last.items() if kk == k] for k in range(len(logs))]
        self.norm = None
        self.capture_optimizer = False
        self._grad_of = [(p, p.grad) for p in model.parameters() if p.grad is not None]
        self._grad_by_idx = {i: red.params[i].grad for i in last}
        self.grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        self.params = [p for p, _ in self._grad_of]
        self.tr.optimizer.zero_grad(set_to_none=True)

    def _replay_segments(self):
        tr, red = self.tr, self.tr.reducer
        exchange = tr.exchange and self.dist
        if exchange:
            red.expect(self.task)
        self.graphs[0].replay()
        n0 = red.stats["launched_early"]
        self.launched_after_segment = []                    # buckets handed to the exchange so far, per segment
        for k, g in enumerate(self.graphs[1:]):
            g.replay()
            if exchange:
                red.mark_ready(self.seg_final[k], self._grad_by_idx)   # complete buckets leave behind this segment
            self.launched_after_segment.append(red.stats["launched_early"] - n0)
        self._done = torch.cuda.Event()
        self._done.record()
        if exchange:
            red.reduce()                                    # control flag, join the side stream, p.grad = mean
        else:
            for p, g in self._grad_of:
                p.grad = g
        o = tr.opts
        self.norm = tr.optimizer.step(max_grad_norm=o.grad_norm if o.grad_norm != -1 else None)
        tr.optimizer.zero_grad(set_to_none=True)
        return self.losses, self.norm

    def __call__(self):
        """One training step on the static batch: returns (per-sample losses, pre-clip gradient norm) -- static device
        tensors, overwritten by the next call."""
        tr, opt = self.tr, self.tr.optimizer
        if self._done is not None:
            self._done.synchronize()                        # the previous replay has read the pinned tables / seed
        tr.global_step += 1
        lr = get_lr_sched(tr.global_step, tr.opts)
        for g in opt.param_groups:
            g["lr"] = lr
        if self.capture_optimizer:
            opt.refresh_graph_tables(self.tabs)
        self.seed_host[0] = int(torch.randint(0, 2 ** 62, (1,)).item())      # torch's CPU generator, as the eager path
        self.seed_dev.copy_(self.seed_host, non_blocking=True)
        if self.segmented:
            return self._replay_segments()
        self.graph.replay()
        self._done = torch.cuda.Event()
        self._done.record()
        if not self.capture_optimizer:                      # several ranks (or debugging): exchange + update launched eagerly
            o = tr.opts
            for p, g in self._grad_of:
                p.grad = g                                  # the graph's static gradient buffers
            if self.dist:
                tr.reducer.expect(self.task)
                tr.reducer.reduce()                         # copies into the buckets, RCCL exchange, p.grad = mean
            self.norm = opt.step(max_grad_norm=o.grad_norm if o.grad_norm != -1 else None)
            opt.zero_grad(set_to_none=True)
            return self.losses, self.norm                   # (opt.step bumped the parameter versions itself)
        for p in self.params:
            _bump_version(p)
        return self.losses, self.norm
