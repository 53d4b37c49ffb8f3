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


# ---- training exchange: GradientReducer == DDP(find_unused_parameters=True) semantics ------------------------
class _Tiny(torch.nn.Module):
    """Two heads; which one is used depends on the sample ('task'), like the task-mixed pre-training step."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.body = torch.nn.Linear(6, 5)
        self.head_a = torch.nn.Linear(5, 3)
        self.head_b = torch.nn.Linear(5, 3)
        self.never = torch.nn.Linear(5, 1)            # unused on every rank -> grad must stay None
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.3

    def forward(self, x, task):
        h = torch.tanh(self.body(x))
        return (self.head_a if task == "a" else self.head_b)(h)


def _tiny_loss(model, x, y, task):
    return torch.nn.functional.cross_entropy(model(x, task), y, reduction="mean")


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gridmm_amd import dist as D
    torch.manual_seed(100 + rank)
    model = _Tiny()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)                               # deliberately different; broadcast must fix it
    D.broadcast_parameters(model.parameters())
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    _tiny_loss(model, xs, ys, "a" if rank == 0 else "b").backward()      # rank 0 never touches head_b and v.v.
    D.GradientReducer(model.parameters(), bucket_mb=1e-4).reduce()        # tiny buckets: exercise several
    out = {k: (None if p.grad is None else p.grad.detach().numpy().copy()) for k, p in model.named_parameters()}
    q.put((rank, out))          # plain numpy: no shared-memory tensor hand-off racing the worker's exit
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_reducer_matches_single_process_mean_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: mean over ranks of the per-rank mean losses
    model = _Tiny()
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
    (0.5 * (_tiny_loss(model, x[:4], y[:4], "a") + _tiny_loss(model, x[4:], y[4:], "b"))).backward()
    for k, p in model.named_parameters():
        for r in range(world):
            got = outs[r][k]
            if p.grad is None:
                assert got is None, k                      # same set of grad-less parameters
            else:
                assert got is not None and torch.allclose(torch.from_numpy(got), p.grad, atol=1e-6), k


def _task_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gridmm_amd.pretrain_loop import TaskSampler
    s = TaskSampler(("mlm", "mrc", "sap"), (1, 1, 1), device="cpu", seed=100 + rank)   # different local seeds
    q.put((rank, [s.next_task() for _ in range(12)]))
    dist.barrier()
    dist.destroy_process_group()


def test_task_sampler_all_ranks_train_the_same_task_world2():
    """data/loader.py:50-58: rank 0's multinomial draw is broadcast (the 1-int exchange of every step)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_task_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[0] == outs[1] and len(set(outs[0])) > 1


def _overlap_worker(rank, world, port, q):
    """Several steps with the reducer constructed BEFORE backward (hooks fill the buckets and launch them during
    backward once the task's used-set is known), alternating tasks, plus one step whose announced key is wrong."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gridmm_amd import dist as D
    model = _Tiny()
    red = D.GradientReducer(model.parameters(), bucket_mb=1e-4)          # several buckets
    g = torch.Generator().manual_seed(11)
    outs, early = [], []
    plan = ["a", "b", "a", "b", "a", ("b", "a")]                          # last step: announces b, runs a
    for step, task in enumerate(plan):
        announce, run = task if isinstance(task, tuple) else (task, task)
        x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
        xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
        for p in model.parameters():
            p.grad = None
        red.expect(announce)
        _tiny_loss(model, xs, ys, run).backward()
        early.append(sum(b["work"] is not None for b in red.buckets))     # buckets launched before reduce()
        red.reduce()
        outs.append({k: (None if p.grad is None else p.grad.detach().numpy().copy()) for k, p in model.named_parameters()})
    q.put((rank, outs, early))        # plain numpy: no shared-memory tensor hand-off at process exit
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_reducer_hooks_overlap_and_task_switching_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (o, e) for r, o, e in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _Tiny()
    g = torch.Generator().manual_seed(11)
    for step, task in enumerate(["a", "b", "a", "b", "a", "a"]):
        x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
        for p in model.parameters():
            p.grad = None
        (0.5 * (_tiny_loss(model, x[:4], y[:4], task) + _tiny_loss(model, x[4:], y[4:], task))).backward()
        for k, p in model.named_parameters():
            for r in range(world):
                got = res[r][0][step][k]
                if p.grad is None:
                    assert got is None, (step, k)
                else:
                    assert got is not None and torch.allclose(torch.from_numpy(got), p.grad, atol=1e-6), (step, k)
    for r in range(world):
        early = res[r][1]
        assert early[0] == 0 and early[1] == 0          # first sight of each task: nothing to predict from
        assert early[2] > 0 and early[3] > 0 and early[4] > 0   # known used-sets: buckets go out during backward


def _mixed_worker(rank, world, port, algo, payload, q):
    """ranks that use DIFFERENT parameter subsets for the same announced key, several steps, early launches on:
    the collective order must not depend on the rank (it hung / aborted with rank-dependent launches)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gridmm_amd import dist as D
    model = _Tiny()
    red = D.GradientReducer(model.parameters(), bucket_mb=1e-4, overlap=True, algo=algo, payload=payload)
    g = torch.Generator().manual_seed(23)
    outs = []
    for step in range(4):
        x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
        xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
        for p in model.parameters():
            p.grad = None
        red.expect("x")
        # step 2 swaps the heads on both ranks (a deviation from the remembered path of key "x" on every rank)
        task = ("a" if rank == 0 else "b") if step != 2 else ("b" if rank == 0 else "a")
        _tiny_loss(model, xs, ys, task).backward()
        red.reduce()
        outs.append({k: (None if p.grad is None else p.grad.detach().numpy().copy()) for k, p in model.named_parameters()})
    q.put((rank, outs, dict(red.stats)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,payload", [("ring", "fp32"), ("direct", "fp32"), ("direct", "bf16")])
def test_gradient_reducer_mixed_usage_multi_step_world2(algo, payload):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mixed_worker, args=(r, world, port, algo, payload, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (o, st) for r, o, st in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _Tiny()
    g = torch.Generator().manual_seed(23)
    tol = 1e-6 if payload == "fp32" else 2.0 ** -7
    for step in range(4):
        x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
        for p in model.parameters():
            p.grad = None
        t0, t1 = ("a", "b") if step != 2 else ("b", "a")
        (0.5 * (_tiny_loss(model, x[:4], y[:4], t0) + _tiny_loss(model, x[4:], y[4:], t1))).backward()
        for k, p in model.named_parameters():
            for r in range(world):
                got = res[r][0][step][k]
                if p.grad is None:
                    assert got is None, (step, k)
                else:
                    assert got is not None, (step, k)
                    err = (torch.from_numpy(got) - p.grad).abs().max().item()
                    assert err <= tol * max(1.0, p.grad.abs().max().item()), (step, k, err)
    assert res[0][1]["launched_early"] > 0 and res[1][1]["launched_early"] > 0      # overlap stayed on


# ---- segmented backward + mark_ready: the host protocol of the captured multi-rank training step ---------------------
class _Chain(torch.nn.Module):
    """Three blocks with hostsync boundaries between them and a weight shared by the first and the last block (the MLM
    decoder / word-embedding tie of the pre-training model): backward in three segments."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.tied = torch.nn.Parameter(torch.randn(6, 6, generator=g) * 0.3)
        self.l1 = torch.nn.Linear(6, 6)
        self.l2 = torch.nn.Linear(6, 6)
        self.l3 = torch.nn.Linear(6, 3)
        for p in (*self.l1.parameters(), *self.l2.parameters(), *self.l3.parameters()):
            p.data = torch.randn(p.shape, generator=g) * 0.3

    def forward(self, x):
        from gridmm_amd import hostsync as hs
        h0 = hs.boundary(2, torch.tanh(self.l1(x @ self.tied)))          # consumed by block 2 AND by the head below
        h1 = hs.boundary(1, torch.tanh(self.l2(h0)))
        return self.l3(h1 @ self.tied + h0)


def _segment_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib
    from gridmm_amd import dist as D, hostsync as hs
    model = _Chain()
    red = D.GradientReducer(model.parameters(), bucket_mb=1e-4)          # several buckets
    g = torch.Generator().manual_seed(31)
    outs, early = [], []
    seg_final = None
    for step in range(4):
        x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
        xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
        for p in model.parameters():
            p.grad = None
        red.expect("k")
        if step == 0:                         # an ordinary eager step first: the key's used-set becomes known
            torch.nn.functional.cross_entropy(model(xs), ys).backward()
        else:
            # what GraphedTrainStep does: backward in segments with the hooks in capture mode (copy into the slot + log),
            # then -- per segment, as after each graph replay -- mark_ready() for the parameters that segment FINISHED
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
            assert n == 3 and len(logs) == 3
            last = {i: k for k, log in enumerate(logs) for i in log}
            seg_final = [[i for i, kk in last.items() if kk == k] for k in range(n)]
            for p in model.parameters():      # (a replay only runs kernels: the host sees no gradients until mark_ready)
                p.grad = None
            for k in range(n):
                red.mark_ready(seg_final[k])
                if k == 0:
                    early.append(sum(b["work"] is not None for b in red.buckets))
        red.reduce()
        outs.append({k: (None if p.grad is None else p.grad.detach().numpy().copy()) for k, p in model.named_parameters()})
    q.put((rank, outs, early, seg_final, dict(red.stats)))
    dist.barrier()
    dist.destroy_process_group()


def test_segmented_backward_feeds_the_reducer_like_an_eager_backward_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_segment_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: rest for r, *rest in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _Chain()
    g = torch.Generator().manual_seed(31)
    for step in range(4):
        x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
        for p in model.parameters():
            p.grad = None
        (0.5 * (torch.nn.functional.cross_entropy(model(x[:4]), y[:4]) +
                torch.nn.functional.cross_entropy(model(x[4:]), y[4:]))).backward()
        for k, p in model.named_parameters():
            for r in range(world):
                got = res[r][0][step][k]
                assert got is not None and torch.allclose(torch.from_numpy(got), p.grad, atol=1e-6), (step, k)
    names = [k for k, _ in model.named_parameters()]
    for r in range(world):
        outs, early, seg_final, stats = res[r]
        final = {names[i]: k for k, idx in enumerate(seg_final) for i in idx}
        assert final["tied"] == 2 and final["l3.weight"] == 0 and final["l2.weight"] == 1 and final["l1.weight"] == 2
        assert all(e > 0 for e in early), early        # buckets left after the FIRST segment, before the others ran
        assert stats["repairs"] == 0
