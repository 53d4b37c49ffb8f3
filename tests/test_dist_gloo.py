"""CPU, world_size 2 over gloo: episode sharding, pickled result gather, max-over-ranks timing (the N>1 path
of bench.py / the eval loop has no other collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gridmm_amd import dist as D
    idx = D.shard_indices(n_items)
    results = [{"instr_id": i, "path": [i, i * i]} for i in idx]          # per-episode trajectories
    merged = [r for part in D.all_gather_objects(results) for r in part]
    t = D.max_over_ranks(1.0 + rank)                                      # rank 1 is the slow one
    dist.barrier()
    q.put((rank, idx, merged, t))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 32])
def test_sharding_gather_and_timing_world2(n_items):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx0, idx1 = outs[0][1], outs[1][1]
    assert sorted(idx0 + idx1) == list(range(n_items)) and not set(idx0) & set(idx1)   # disjoint cover
    for rank, _, merged, t in outs:
        assert sorted(m["instr_id"] for m in merged) == list(range(n_items))             # every rank sees all
        assert t == 2.0                                                                    # max over ranks


def test_single_process_degrades_gracefully():
    from gridmm_amd import dist as D
    assert D.rank_world() == (0, 1)
    assert D.shard_indices(5) == [0, 1, 2, 3, 4]
    assert D.all_gather_objects({"a": 1}) == [{"a": 1}]
    assert D.max_over_ranks(0.25) == 0.25
