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
                                       (no copy / memset NODES in the graph: see KERNEL NODES ONLY below)
  packed-weight caches                 the re-split of every weight is part of the captured step; versions are bumped
                                       after a replay so that eager users of the caches re-pack

KERNEL NODES ONLY.  Rounds 2-3 ran this class only with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: with the runtime's default
pre-recorded graph packets the full-size step died with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION as soon as eager work
alternated with replays.  Root cause (round 4; tools/dbg_train_graph_fault.py, tools/find_copy_nodes.py,
profiles/r4_train_graph_fault.txt): the 1100-1300-node graph held SEVEN non-kernel nodes issued by torch -- 4 memcpy
(two .clone() in the aggregation's forward, two select_backward copies of the token-type rows) and 3 memset (the
semaphores of two large `sum`s = gradients of the broadcast token-type adds, and the sort-based embedding backward of
nav_type_embedding).  With such nodes in a large graph the replay leaves ONE queue slot holding the packet of an EARLIER
dispatch (the fault dump showed a rocprim partition kernel from the record step, always 322-324 packets before the
write pointer); its kernel-argument block has been recycled by later eager launches, hence wild pointers -- and "memset
nodes that did not clear in time" (round 2) is the same thing seen from the other side.  With no eager work between
replays the stale slot re-executes a harmless packet of the previous replay, which is why soaks passed.  The fix is ours:
every one of those operations now has a kernel form (autograd.kernel_copy / add_row / small_embedding, the reducer's slot
copies), the graphs are captured with keep_graph and CHECKED (hipGraphNodeGetType): a graph with a non-kernel node is
refused unless DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is set.  All scenarios that faulted pass on the default runtime.

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


from .graph import CAPTURE_MODE   # "thread_local": RCCL's watchdog thread polls events while this thread captures


def _new_graph():
    """A CUDAGraph that keeps its hipGraph_t so that the node types can be inspected (older torch: a plain one)."""
    try:
        return torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:
        return torch.cuda.CUDAGraph()


def check_kernel_only(graphs, what):
    """Refuse a captured step that holds memcpy / memset / other non-kernel nodes when the runtime replays pre-recorded
    packets (see the module docstring); returns the summed node-type histogram (None if the runtime hides the graph)."""
    from .graph import graph_node_types
    total = {}
    for g in graphs:
        h = graph_node_types(g)
        if h is None:
            return None
        for k, v in h.items():
            total[k] = total.get(k, 0) + v
    other = {k: v for k, v in total.items() if k != "kernel"}
    if other and os.environ.get(RUNTIME_ENV[0]) != RUNTIME_ENV[1] and not os.environ.get("GRIDMM_TRAIN_GRAPH_ANY_RUNTIME"):
        raise RuntimeError("%s: the captured graph holds non-kernel nodes %s.  The runtime's pre-recorded graph packets replay "
                           "such graphs incorrectly (stale queue slot -> wild kernel arguments): give the operation a kernel "
                           "form (autograd.kernel_copy and friends; tools/find_copy_nodes.py names the torch op), or set "
                           "%s=%s before torch / HIP start" % (what, other, RUNTIME_ENV[0], RUNTIME_ENV[1]))
    return total


class GraphedTrainStep:
    def __init__(self, trainer, batch, task, warmup=1, capture_optimizer=True, segments=None):
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
            self.graph = _new_graph()
            with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
                with hs.replay(tape):
                    losses = model(batch, task=task, compute_loss=True)
                    losses.mean().backward()
                    self.norm = None
                    if capture_optimizer:
                        self.norm = opt.step(max_grad_norm=o.grad_norm if o.grad_norm != -1 else None, graph_tabs=self.tabs)
                self.losses = losses.detach()
            self.node_types = check_kernel_only([self.graph], "GraphedTrainStep(%s)" % task)
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
        g0 = _new_graph()
        try:
            with torch.cuda.graph(g0, capture_error_mode=CAPTURE_MODE):
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
            g = _new_graph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=CAPTURE_MODE):
                yield
            self.graphs.append(g)
            logs.append(red.take_capture_log())
        red.begin_capture()
        try:
            hs.segmented_backward(loss, cuts, segment)
        finally:
            red.end_capture()
        del cuts, loss
        self.node_types = check_kernel_only(self.graphs, "GraphedTrainStep(%s, segmented)" % task)
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
        # the captured step holds no weight-pack launches (the optimizer writes the planes): anything that changed a weight
        # behind the cache since the last step (load_state_dict, an update without plane records) is re-packed now
        ag.WEIGHTS.ensure_current(self.params)
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
        for p in self.tabs.get("owned", ()):                # the replayed update wrote these weights' planes
            ag.WEIGHTS.mark_written(p)
        return self.losses, self.norm
