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
