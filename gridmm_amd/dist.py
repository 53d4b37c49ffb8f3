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
      * gradients are packed into flat fp32 buckets of `bucket_mb` (default 128 MiB: xGMI is point-to-point, a
        ring all-reduce is per-link bound, so few large transfers beat many 25 MiB DDP buckets; ~645 MB of fp32
        gradients for the 161 M-parameter model = 5 buckets), all buckets are launched asynchronously
        back-to-back and waited once;
      * `find_unused_parameters` semantics: a parameter that received no gradient on this rank contributes zeros;
        a parameter unused on EVERY rank keeps grad None (a 1-int-per-parameter usage vector is summed first), so
        the optimizer skips it exactly as it does single-process.
    World size 1 (or no process group): no-op.
    """

    def __init__(self, params, bucket_mb=128):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)

    def reduce(self):
        if not is_dist() or not self.params:
            return
        world = dist.get_world_size()
        dev = self.params[0].device
        used = torch.tensor([0 if p.grad is None else 1 for p in self.params], dtype=torch.int32, device=dev)
        dist.all_reduce(used, op=dist.ReduceOp.SUM)
        used = used.cpu().tolist()
        live = [p for p, u in zip(self.params, used) if u > 0]
        buckets, cur, n = [], [], 0
        for p in live:
            if cur and n + p.numel() > self.bucket_elems:
                buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            buckets.append(cur)
        flats, works = [], []
        for b in buckets:
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in b])
            flats.append(flat)
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        for b, flat in zip(buckets, flats):
            flat.div_(world)
            o = 0
            for p in b:
                g = flat[o:o + p.numel()].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                o += p.numel()


def broadcast_parameters(params, src=0):
    """Same initial weights on every rank (DDP does this at construction)."""
    if not is_dist():
        return
    for p in params:
        dist.broadcast(p.data, src)
