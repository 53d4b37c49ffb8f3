"""Episode sharding over the GPUs of one node.

Episodes are independent units: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI
on ROCm; "gloo" in CPU tests), every rank advances its own shard of the episode batch, and the step
path has NO collective.  The only exchanges are the end-of-split result gather (reference:
map_nav_src/utils/distributed.py:90-130, main_nav.py:188) and the max-over-ranks timing of bench.py.
"""
import pickle

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world():
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def shard_indices(n_items, rank=None, world=None):
    """Contiguous split used by the reference's eval sharding (map_nav_src/r2r/env.py:427-435):
    rank r takes items [r*ceil(n/w), (r+1)*ceil(n/w))."""
    if rank is None:
        rank, world = rank_world()
    per = -(-n_items // world)
    return list(range(min(rank * per, n_items), min((rank + 1) * per, n_items)))


def all_gather_objects(obj, device=None):
    """Pickled all_gather of arbitrary python objects (reference utils/distributed.py:90-130): sizes first,
    then padded uint8 payloads.  Returns the list of every rank's object (identity when not distributed)."""
    if not is_dist():
        return [obj]
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    buf = torch.frombuffer(bytearray(pickle.dumps(obj)), dtype=torch.uint8).to(device)
    size = torch.tensor([buf.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    pad[:buf.numel()] = buf
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [pickle.loads(o[:int(s.item())].cpu().numpy().tobytes()) for o, s in zip(outs, sizes)]


def max_over_ranks(seconds, device=None):
    """bench.py timing contract: the slowest rank defines the step time."""
    if not is_dist():
        return float(seconds)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class GradientReducer:
    """The one exchange of a training step: all-reduce(mean) of the parameter gradients across ranks.

    Replaces DistributedDataParallel(find_unused_parameters=True) of the reference (fine-tune:
    map_nav_src/r2r/agent_base.py:115-117; pre-training: pretrain_src/utils/misc.py:52-65,
    train_r2r.py:256-258) with explicit, few and large collectives:
      * PERSISTENT flat fp32 buckets of `bucket_mb` (default 128 MiB: xGMI is point-to-point, a ring all-reduce is
        per-link bound, so few large transfers beat many 25 MiB DDP buckets; ~645 MB of fp32 gradients for the
        161 M-parameter model = 5 buckets), filled in reverse parameter order -- the order backward produces them;
      * a post-accumulate hook per parameter copies the finished gradient into its bucket slot (the only copy: fp32
        parameters then keep the slot as their .grad, so the reduced values need no copy back) and, once every
        gradient the bucket expects has arrived, launches the bucket's asynchronous all-reduce WHILE backward is
        still running;
      * `find_unused_parameters` semantics without a per-step host sync: which parameters receive a gradient depends
        only on the code path (the pre-training task), so the used-set of a step is compared across ranks the first
        time it is seen (one small all-reduce + .cpu()) and trusted afterwards; `expect(key)` tells the reducer which
        used-set the coming backward will produce (e.g. the task name), which is what lets buckets launch early.
        A parameter unused on every rank keeps grad None, so the optimizer skips it exactly as it does
        single-process; one unused here but used elsewhere contributes zeros.
    World size 1 (or no process group): no-op.
    """

    def __init__(self, params, bucket_mb=128, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        self.overlap = overlap
        limit = max(1, int(bucket_mb * (1 << 20) // 4))
        self.buckets, self.slot = [], {}                 # bucket: dict(idx, numel, flat, work, pending); slot[i] = (b, off)
        cur, n = [], 0
        for i in reversed(range(len(self.params))):
            k = self.params[i].numel()
            if cur and n + k > limit:
                self.buckets.append(dict(idx=cur, numel=n))
                cur, n = [], 0
            self.slot[i] = (len(self.buckets), n)
            cur.append(i)
            n += k
        if cur:
            self.buckets.append(dict(idx=cur, numel=n))
        for b in self.buckets:
            b.update(flat=None, work=None, pending=None)
        self._ready = [False] * len(self.params)
        self._late = []
        self._verified = set()                           # used-sets already compared across ranks
        self._sig_by_key, self._key, self._expected = {}, None, None
        self._hook_fns = [self._make_hook(i) for i in range(len(self.params))]
        self._hooks = [p.register_post_accumulate_grad_hook(f) for p, f in zip(self.params, self._hook_fns)]

    # ---- bucket plumbing
    def _flat(self, b):
        if b["flat"] is None:
            b["flat"] = torch.zeros(b["numel"], dtype=torch.float32, device=self.params[b["idx"][0]].device)
        return b["flat"]

    def _view(self, i):
        bi, off = self.slot[i]
        p = self.params[i]
        return self._flat(self.buckets[bi])[off:off + p.numel()].view(p.shape)

    def _launch(self, b):
        for i in b["idx"]:                               # slots nobody filled this step must not carry last step's values
            if not self._ready[i]:
                self._view(i).zero_()
        b["work"] = dist.all_reduce(self._flat(b), op=dist.ReduceOp.SUM, async_op=True)

    def _make_hook(self, i):
        def hook(p):
            if not is_dist():
                return
            bi, _ = self.slot[i]
            b = self.buckets[bi]
            if b["work"] is not None:                    # the prediction said "unused": reduced separately in reduce()
                self._late.append(i)
                return
            v = self._view(i)
            v.copy_(p.grad)
            if p.dtype == torch.float32:
                p.grad = v                               # gradient IS the bucket slot from here on
            self._ready[i] = True
            if b["pending"] is not None:
                b["pending"].discard(i)
                if not b["pending"]:
                    self._launch(b)
        return hook

    def expect(self, key, final=True):
        """Announce the code path of the coming backward (any hashable, e.g. the pre-training task).  Steps with a key
        whose used-set is already known launch their buckets during backward.  final=False: a gradient-accumulation
        micro-step that is NOT followed by reduce() -- gradients keep accumulating in the slots, nothing is launched."""
        self._key = key
        sig = self._sig_by_key.get(key)
        self._expected = sig
        early = sig is not None and self.overlap and final
        for b in self.buckets:
            b["pending"] = ({i for i in b["idx"] if sig[i] and not self._ready[i]} or None) if early else None

    def reduce(self):
        """Call after backward: finishes the exchange; parameters' .grad then hold the mean over ranks."""
        if not is_dist() or not self.params:
            return
        world = dist.get_world_size()
        for i, p in enumerate(self.params):              # gradients produced before the hooks existed / outside autograd
            if p.grad is not None and not self._ready[i] and i not in self._late:
                self._hook_fns[i](p)
        sig = tuple(self._ready[i] or (i in self._late) for i in range(len(self.params)))
        union = sig
        if sig not in self._verified:
            dev = self.params[0].device
            used = torch.tensor([int(u) for u in sig], dtype=torch.int32, device=dev)
            dist.all_reduce(used, op=dist.ReduceOp.SUM)
            counts = used.cpu().tolist()
            union = tuple(c > 0 for c in counts)
            if all(c in (0, world) for c in counts):
                self._verified.add(sig)                  # every rank took the same path: no exchange next time
        for b in self.buckets:
            if b["work"] is None and any(union[i] for i in b["idx"]):
                self._launch(b)
        late = None
        if self._late:                                   # mispredicted parameters: one extra small all-reduce
            late = torch.cat([self.params[i].grad.reshape(-1).float() for i in self._late])
            dist.all_reduce(late, op=dist.ReduceOp.SUM)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["flat"].div_(world)
        o = 0
        for i in self._late:
            p = self.params[i]
            p.grad = (late[o:o + p.numel()] / world).view_as(p).to(p.dtype)
            o += p.numel()
        for i, p in enumerate(self.params):
            if i in self._late:
                continue
            if not union[i]:
                p.grad = None                            # unused on every rank: the optimizer skips it
            elif p.dtype != torch.float32:
                p.grad = self._view(i).to(p.dtype)
            elif not self._ready[i]:
                p.grad = self._view(i)                   # unused here, used elsewhere: the others' mean contribution
        self._sig_by_key[self._key] = union
        self._ready = [False] * len(self.params)
        self._late = []
        for b in self.buckets:
            b["work"], b["pending"] = None, None
        self.expect(self._key)                           # same path next step unless the caller says otherwise


def broadcast_parameters(params, src=0, chunk_mb=256):
    """Same initial weights on every rank (DDP does this at construction): parameters travel flattened, one
    broadcast per dtype and <= chunk_mb, instead of one collective per tensor."""
    if not is_dist():
        return
    params = list(params)
    groups = {}
    for p in params:
        groups.setdefault((p.dtype, p.device), []).append(p)
    for (dtype, dev), ps in groups.items():
        limit = max(1, int(chunk_mb * (1 << 20)) // ps[0].element_size())
        i = 0
        while i < len(ps):
            j, n = i, 0
            while j < len(ps) and (j == i or n + ps[j].numel() <= limit):
                n += ps[j].numel()
                j += 1
            flat = torch.cat([p.data.reshape(-1) for p in ps[i:j]])
            dist.broadcast(flat, src)
            o = 0
            for p in ps[i:j]:
                p.data.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            i = j
