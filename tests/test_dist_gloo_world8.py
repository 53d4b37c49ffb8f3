"""CPU, gloo, world 4 and 8: the host protocol of the gradient exchange at the reference's rank counts
(pretrain_src/run_r2r.sh:2-8 launches 3-8 ranks; pretrain_src/utils/misc.py:52-65 wraps the model in DDP with
find_unused_parameters=True; map_nav_src/scripts/run_r2r.sh:65 launches the fine-tune the same way).  World 2 is in
tests/test_dist_gloo.py; these cases are what VERDICT r4 item 3 lists: buckets whose length is not a multiple of the world
(padded shards of the direct reduce-scatter / all-gather), eight ranks with eight different used-sets and a deviation step,
all_to_all_single shard order, segmented mark_ready with four ranks, and the eval-side sharding / gather at eight ranks."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_dist_gloo import _Chain, _free_port


class _Wide(torch.nn.Module):
    """A body and five heads with ODD sizes (bucket lengths that no world size divides); rank r uses head r % 4 -- and on
    odd ranks head 4 as well -- so that with eight ranks no two neighbours have the same used-set; `never` stays unused."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.body = torch.nn.Linear(7, 11)
        self.heads = torch.nn.ModuleList([torch.nn.Linear(11, 3) for _ in range(5)])
        self.never = torch.nn.Linear(11, 1)
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.3

    def forward(self, x, rank, swap=False):
        h = torch.tanh(self.body(x))
        k = (rank + (2 if swap else 0)) % 4
        out = self.heads[k](h)
        if rank % 2 == 1:
            out = out + self.heads[4](h)
        return out


def _loss(model, x, y, rank, swap):
    return torch.nn.functional.cross_entropy(model(x, rank, swap), y, reduction="mean")


def _batch(gen, world):
    return torch.randn(2 * world, 7, generator=gen), torch.randint(0, 3, (2 * world,), generator=gen)


def _wide_worker(rank, world, port, algo, payload, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gridmm_amd import dist as D
    torch.manual_seed(50 + rank)
    model = _Wide()
    if rank:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.1 * rank)                    # broadcast_parameters must undo this on every rank
    D.broadcast_parameters(model.parameters())
    red = D.GradientReducer(model.parameters(), bucket_mb=3e-4, overlap=True, algo=algo, payload=payload)
    pads = [int(b["padded"]) for b in red.buckets]
    g = torch.Generator().manual_seed(41)
    outs = []
    for step in range(4):
        x, y = _batch(g, world)
        xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
        for p in model.parameters():
            p.grad = None
        red.expect("k")
        _loss(model, xs, ys, rank, swap=(step == 2)).backward()          # step 2: every rank leaves its remembered used-set
        red.reduce()
        outs.append({k: (None if p.grad is None else p.grad.detach().numpy().copy()) for k, p in model.named_parameters()})
    q.put((rank, outs, dict(red.stats), pads))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, world, args, timeout=240):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,algo,payload", [(4, "ring", "fp32"), (4, "direct", "fp32"), (8, "direct", "fp32"),
                                                (8, "direct", "bf16"), (8, "ring", "fp32")])
def test_gradient_reducer_eight_used_sets_padded_buckets(world, algo, payload):
    res = {r: rest for r, *rest in _spawn(_wide_worker, world, (algo, payload))}
    model = _Wide()
    g = torch.Generator().manual_seed(41)
    tol = 1e-6 if payload == "fp32" else 2.0 ** -7
    for step in range(4):
        x, y = _batch(g, world)
        for p in model.parameters():
            p.grad = None
        sum(_loss(model, x[2 * r:2 * r + 2], y[2 * r:2 * r + 2], r, swap=(step == 2)) for r in range(world)).div(world).backward()
        for k, p in model.named_parameters():
            for r in range(world):
                got = res[r][0][step][k]
                if p.grad is None:
                    assert got is None, (step, k, r)          # unused on EVERY rank: stays None everywhere
                else:
                    assert got is not None, (step, k, r)      # used on SOME rank: every rank holds the mean
                    err = (torch.from_numpy(got) - p.grad).abs().max().item()
                    assert err <= tol * max(1.0, p.grad.abs().max().item()), (step, k, r, err)
    # ranks agree bit for bit after the exchange (the direct sum runs in rank order on every shard owner)
    for step in range(4):
        for k in res[0][0][step]:
            a = res[0][0][step][k]
            for r in range(1, world):
                b = res[r][0][step][k]
                assert (a is None) == (b is None) and (a is None or (a == b).all()), (step, k, r)
    pads = res[0][2]
    assert len(pads) >= 2, pads                               # several buckets
    if algo == "direct":
        assert all(n % world == 0 for n in pads), pads        # shards of equal length: the flat buffers are padded
    n_params = sum(p.numel() for p in model.parameters())
    assert sum(pads) >= n_params
    assert all(res[r][1]["launched_early"] > 0 for r in range(world)), [res[r][1] for r in range(world)]
    assert all(res[r][1]["repairs"] >= 1 for r in range(world))          # the deviation step ran a repair round everywhere


def _segment_worker_n(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib
    from gridmm_amd import dist as D, hostsync as hs
    model = _Chain()
    red = D.GradientReducer(model.parameters(), bucket_mb=1e-4, algo="direct")
    g = torch.Generator().manual_seed(31)
    outs, early = [], []
    for step in range(3):
        x, y = torch.randn(2 * world, 6, generator=g), torch.randint(0, 3, (2 * world,), generator=g)
        xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
        for p in model.parameters():
            p.grad = None
        red.expect("k")
        if step == 0:
            torch.nn.functional.cross_entropy(model(xs), ys).backward()
        else:
            hs.CUTS = cuts = []
            loss = torch.nn.functional.cross_entropy(model(xs), ys)
            hs.CUTS = None
            logs = []

            @contextlib.contextmanager
            def seg(k):
                yield
                logs.append(red.take_capture_log())
            red.enabled = False
            red.begin_capture()
            n = hs.segmented_backward(loss, cuts, seg)
            red.end_capture()
            red.enabled = True
            last = {i: k for k, log in enumerate(logs) for i in log}
            seg_final = [[i for i, kk in last.items() if kk == k] for k in range(n)]
            for p in model.parameters():
                p.grad = None
            for k in range(n):
                red.mark_ready(seg_final[k])
                if k == 0:
                    early.append(sum(b["work"] is not None for b in red.buckets))
        red.reduce()
        outs.append({k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()})
    q.put((rank, outs, early, dict(red.stats)))
    dist.barrier()
    dist.destroy_process_group()


def test_segmented_mark_ready_world4_direct():
    world = 4
    res = {r: rest for r, *rest in _spawn(_segment_worker_n, world, ())}
    model = _Chain()
    g = torch.Generator().manual_seed(31)
    for step in range(3):
        x, y = torch.randn(2 * world, 6, generator=g), torch.randint(0, 3, (2 * world,), generator=g)
        for p in model.parameters():
            p.grad = None
        sum(torch.nn.functional.cross_entropy(model(x[2 * r:2 * r + 2]), y[2 * r:2 * r + 2]) for r in range(world)).div(world).backward()
        for k, p in model.named_parameters():
            for r in range(world):
                assert torch.allclose(torch.from_numpy(res[r][0][step][k]), p.grad, atol=1e-6), (step, k, r)
    for r in range(world):
        assert all(e > 0 for e in res[r][1]) and res[r][2]["repairs"] == 0


def _a2a_worker(rank, world, port, q):
    """The shard order of the direct exchange: after all_to_all_single rank j holds shard j of every rank, in RANK order."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gridmm_amd import dist as D
    n = 5                                                # elements per shard
    red = D.GradientReducer([torch.nn.Parameter(torch.zeros(3))], algo="direct")
    flat = torch.arange(world * n, dtype=torch.float32) + 1000.0 * rank         # element e of rank r = 1000 r + e
    red._exchange(flat)                                  # shard j -> rank j, ordered sum, reduced shards back to their places
    got = flat
    raw = torch.empty(world * n)
    dist.all_to_all_single(raw, torch.arange(world * n, dtype=torch.float32) + 1000.0 * rank)
    items = D.shard_indices(19)
    gathered = D.all_gather_objects({"rank": rank, "items": items})
    slow = D.max_over_ranks(0.1 * (rank + 1))
    q.put((rank, got.numpy().copy(), gathered, slow, raw.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_to_all_shard_order_and_eval_sharding_world8():
    world, n = 8, 5
    res = {r: rest for r, *rest in _spawn(_a2a_worker, world, ())}
    for j in range(world):
        want_sum = 1000.0 * sum(range(world)) + world * torch.arange(world * n, dtype=torch.float32).numpy()
        assert (res[j][0] == want_sum).all(), j          # every element back in ITS place with the sum over ranks
        raw = res[j][3].reshape(world, n)
        for r in range(world):                           # the collective itself: row r of rank j = shard j of rank r
            want = 1000.0 * r + torch.arange(j * n, (j + 1) * n, dtype=torch.float32).numpy()
            assert (raw[r] == want).all(), (j, r)
        gathered = res[j][1]
        assert [g["rank"] for g in gathered] == list(range(world))
        assert sorted(i for g in gathered for i in g["items"]) == list(range(19))       # contiguous ceil split, no overlap
        assert abs(res[j][2] - 0.8) < 1e-9
