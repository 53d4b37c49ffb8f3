"""(run by tests/test_hip_train_graph.py, one subprocess per case, on the runtime's DEFAULT graph settings)
GPU: the pre-training step as one hipGraph (train_graph.GraphedTrainStep) against the eager step it captures
(pretrain_loop.PreTrainer.train_step; reference loop pretrain_src/train_r2r.py:231-303): same losses, same gradient norms,
same parameters after several optimizer steps (warm-up lr schedule and AdamW bias correction advancing per replay),
for each of the three tasks; with dropout on, replays draw new masks and two identically seeded runs agree."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TASKS = ("mlm", "mrc", "sap")


def _setup(drop, fp32_grid_proj=True, layers=2, seed_off=0):
    from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from gridmm_amd.synthetic import batch_to, make_pretrain_batch
    from gridmm_amd.vilmodel import default_config
    dev = torch.device("cuda")
    cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=list(TASKS), image_prob_size=1000, obj_prob_size=0,
                         num_l_layers=layers, num_pano_layers=1, num_x_layers=2, hidden_dropout_prob=drop,
                         attention_probs_dropout_prob=drop)
    torch.manual_seed(0)
    model = GlocalTextPathCMTPreTraining(cfg).to(dev)
    if fp32_grid_proj:
        # The reference keeps grid_proj (weights, gradients, AdamW state) in fp16 (vilmodel.py:664): its second moment
        # underflows and the update becomes m / eps, right at the fp16 rounding threshold of the weights.  One-ulp flips
        # driven by the 1e-6 summation-order noise of the atomics then split the loss trajectories of two otherwise
        # identical runs (eager vs eager as much as graph vs eager; tools/dbg_determinism.py: 1e-4 ... 3e-3 after 5-8
        # steps, 1e-5 with an fp32 grid_proj).  Trajectory comparisons use fp32; case_fp16_grid_proj covers the fp16 path.
        model.bert.grid_proj.float()
    batches = {t: batch_to(make_pretrain_batch(np.random.RandomState(i + seed_off), 4, t, max_steps=3, L=40, vocab=30000,
                                               image_prob_size=1000, n_pts=(588, 588 * 2)), dev)
               for i, t in enumerate(TASKS)}
    return model, batches


def _sync(ta, tb):
    """tb <- ta: parameters and AdamW moments (the step counters advance identically by construction).  Rounds 1-2 needed
    this: the backward's float atomics made gradients run-dependent at 1e-7 and the model's discontinuities (arg-max token
    routing, the fp16 grid_proj at its rounding threshold) split two runs after a few steps.  Round 3 removed the atomics
    (case_trajectory compares un-synchronised trajectories bit for bit); these per-step comparisons stay as they were."""
    with torch.no_grad():
        for pa, pb in zip(ta.model.parameters(), tb.model.parameters()):
            pb.copy_(pa)
            sa, sb = ta.optimizer.state.get(pa), tb.optimizer.state.get(pb)
            if sa:
                assert sa["step"] == sb["step"]
                sb["exp_avg"].copy_(sa["exp_avg"])
                sb["exp_avg_sq"].copy_(sa["exp_avg_sq"])


def _compare_step(ta, tb, eager, graphed, tag):
    _sync(ta, tb)
    la, na = eager()
    lb, nb = graphed()
    assert torch.isfinite(lb).all()
    assert torch.allclose(la, lb, rtol=2e-5, atol=2e-5), (tag, float((la - lb).abs().max()))
    assert abs(float(na) - float(nb)) <= 5e-5 * float(na), (tag, float(na), float(nb))
    for (n, pa), pb in zip(ta.model.named_parameters(), tb.model.parameters()):
        tol = 6.2e-5 if pa.dtype == torch.float16 else 2e-6      # fp16: one ulp at |w| < 0.0625
        d = float((pa.float() - pb.float()).abs().max())
        assert d <= tol, (tag, n, d)


def case_equals_eager(task, segments=None, layers=2):
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    model, batches = _setup(0.0, layers=layers)
    ma, mb = copy.deepcopy(model), copy.deepcopy(model)
    ta, tb = PreTrainer(ma, default_opts(warmup_steps=10)), PreTrainer(mb, default_opts(warmup_steps=10))
    for _ in range(2):                                   # what the graphed object runs while it is built: record + warm-up
        ta.train_step(batches[task], task)
    g = GraphedTrainStep(tb, batches[task], task, segments=segments)
    if segments:
        assert len(g.graphs) >= 4 and sum(len(x) for x in g.seg_final) == len(g.params), [len(x) for x in g.seg_final]
    assert tb.global_step == ta.global_step == 2
    for it in range(5):                                  # lr warm-up and AdamW bias correction advance with every replay
        _compare_step(ta, tb, lambda: ta.train_step(batches[task], task), g, (task, it))
    assert tb.global_step == ta.global_step == 7
    # eager use after replays: the replayed weights (packed-weight caches re-validated), clean gradients (the capture's
    # gradient buffers must not be left in p.grad, where an eager backward would accumulate into them)
    _compare_step(ta, tb, lambda: ta.train_step(batches[task], task), lambda: tb.train_step(batches[task], task), (task, "eager"))
    _compare_step(ta, tb, lambda: ta.train_step(batches[task], task), g, (task, "again"))


def _dist2_worker(rank, world, port):
    """One rank of case_dist2 (both ranks on this box's one GPU, gloo): eager exchange-overlapped steps (trainer A) against
    the segmented captured step (trainer B) -- same losses, norms and parameters, and buckets that leave BEFORE the last
    backward segment has been launched."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    model, batches = _setup(0.0, layers=5, seed_off=10 * rank)          # same weights (seed 0), different data per rank
    ma, mb = copy.deepcopy(model), copy.deepcopy(model)
    kw = dict(bucket_mb=8, algo="ring")
    ta, tb = PreTrainer(ma, default_opts(warmup_steps=10), reducer_kw=kw), PreTrainer(mb, default_opts(warmup_steps=10), reducer_kw=kw)
    assert len(tb.reducer.buckets) >= 4
    for task in ("sap", "mlm"):
        for _ in range(2):
            ta.train_step(batches[task], task)
        g = GraphedTrainStep(tb, batches[task], task)
        assert g.segmented and len(g.graphs) >= 4
        for it in range(4):
            _compare_step(ta, tb, lambda: ta.train_step(batches[task], task), g, (task, it, rank))
            assert g.launched_after_segment[-1] > 0, g.launched_after_segment
            # overlap: at least one bucket was handed to the exchange while later segments were still to be launched
            assert g.launched_after_segment[-2] > 0, g.launched_after_segment
        # ranks agree after the exchange
        flat = torch.cat([p.detach().float().reshape(-1) for p in mb.parameters()])
        other = flat.clone()
        dist.broadcast(other, 0)
        assert torch.equal(flat, other)
    dist.barrier()
    dist.destroy_process_group()


def case_rccl1():
    """The multi-rank code paths against the REAL backend ("nccl" = RCCL) with ONE rank (RCCL refuses two ranks on one
    device; GRIDMM_DIST_FORCE makes a one-rank group count as distributed): every exchange algorithm and payload as RCCL
    calls on the side stream, the eager step with buckets launched from the backward hooks, and the segmented captured step
    (graph capture next to a live RCCL communicator and its watchdog thread).  World size 1 => the mean over ranks is the
    rank's own gradient: everything must equal the plain single-process training step."""
    import socket
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", GRIDMM_DIST_FORCE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from gridmm_amd import dist as D
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    assert D.is_dist()
    # 1. the exchange itself, every algorithm / payload, on RCCL
    red = D.GradientReducer([torch.nn.Parameter(torch.zeros(1000, device="cuda"))], bucket_mb=1)
    for algo in ("ring", "rsag", "direct"):
        for payload in ("fp32", "bf16"):
            red.algo, red.payload = algo, payload
            flat = torch.randn(4096, device="cuda")
            want = flat.clone() if payload == "fp32" else flat.to(torch.bfloat16).float()
            if algo == "direct" and payload == "bf16":
                want = want.to(torch.bfloat16).float()
            red._exchange(flat, {})
            torch.cuda.synchronize()
            assert torch.allclose(flat, want, rtol=1e-2 if payload == "bf16" else 0, atol=0), (algo, payload)
    # 2. eager overlapped steps and the segmented captured step == each other, step by step
    model, batches = _setup(0.0, layers=5)
    for algo in ("ring", "direct"):
        ma, mb = copy.deepcopy(model), copy.deepcopy(model)     # (a reducer's hooks stay on its parameters: fresh copies)
        kw = dict(bucket_mb=8, algo=algo)
        ta, tb = PreTrainer(ma, default_opts(warmup_steps=10), reducer_kw=kw), PreTrainer(mb, default_opts(warmup_steps=10), reducer_kw=kw)
        assert tb.reducer._host_async() is False and len(tb.reducer.buckets) >= 4
        task = "sap" if algo == "ring" else "mlm"
        for _ in range(2):
            ta.train_step(batches[task], task)
        g = GraphedTrainStep(tb, batches[task], task)
        assert g.segmented and len(g.graphs) >= 4 and (g.node_types is None or set(g.node_types) == {"kernel"})
        for it in range(4):
            _compare_step(ta, tb, lambda: ta.train_step(batches[task], task), g, (algo, task, it))
            assert g.launched_after_segment[-2] > 0, g.launched_after_segment
        assert ta.reducer.stats["launched_early"] > 0
    dist.barrier()
    dist.destroy_process_group()


def case_dist2():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_dist2_worker, args=(r, 2, port)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0, p.exitcode


def case_alternate_full():
    """The scenario that used to die with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on the default runtime (pre-recorded
    graph packets): full-size eager steps of one trainer alternating with graph replays of another, no synchronisation in
    between.  The graphs hold kernel nodes only now (train_graph.py: KERNEL NODES ONLY)."""
    from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.synthetic import batch_to, make_pretrain_batch
    from gridmm_amd.train_graph import GraphedTrainStep
    from gridmm_amd.vilmodel import default_config
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0"
    dev = torch.device("cuda")
    cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=list(TASKS), image_prob_size=1000, obj_prob_size=0,
                         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    m0 = GlocalTextPathCMTPreTraining(cfg).to(dev)
    m0.bert.grid_proj.float()
    batches = {t: batch_to(make_pretrain_batch(np.random.RandomState(i), 32, t, max_steps=5, L=80, vocab=30000,
                                               image_prob_size=1000, n_pts=(588 * 3, 588 * 5)), dev) for i, t in enumerate(TASKS)}
    ma, mb = copy.deepcopy(m0), copy.deepcopy(m0)
    ta, tb = PreTrainer(ma, default_opts(warmup_steps=20)), PreTrainer(mb, default_opts(warmup_steps=20))
    graphs = {}
    for t in TASKS:
        for _ in range(2):
            ta.train_step(batches[t], t)
        graphs[t] = GraphedTrainStep(tb, batches[t], t)
        assert graphs[t].node_types is None or set(graphs[t].node_types) == {"kernel"}, graphs[t].node_types
    for i in range(12):
        t = TASKS[i % 3]
        la, na = ta.train_step(batches[t], t)
        lb, nb = graphs[t]()
        assert torch.equal(la, lb) and float(na) == float(nb), (i, t)      # no atomics on the path: bit-identical
    torch.cuda.synchronize()


def case_nonkernel():
    """A graph with a memcpy node is refused on the default runtime."""
    from gridmm_amd import autograd as ag
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    model, batches = _setup(0.0)
    tr = PreTrainer(model, default_opts(warmup_steps=10))
    ag.kernel_copy = lambda x: x.clone()                  # what the aggregation's forward did before round 4
    try:
        GraphedTrainStep(tr, batches["sap"], "sap")
    except RuntimeError as e:
        assert "non-kernel nodes" in str(e) and "memcpy" in str(e), str(e)
    else:
        raise AssertionError("a graph with memcpy nodes was accepted")


def case_two_graphs():
    """Two tasks, two graphs, one trainer: building the second graph runs eager steps after the first capture."""
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    model, batches = _setup(0.0)
    ma, mb = copy.deepcopy(model), copy.deepcopy(model)
    ta, tb = PreTrainer(ma, default_opts(warmup_steps=10)), PreTrainer(mb, default_opts(warmup_steps=10))
    graphs = {}
    for t in ("mlm", "sap"):
        for _ in range(2):
            ta.train_step(batches[t], t)
        graphs[t] = GraphedTrainStep(tb, batches[t], t)
    for i in range(6):
        t = ("mlm", "sap")[i % 2]
        _compare_step(ta, tb, lambda: ta.train_step(batches[t], t), graphs[t], (t, i))


def case_fp16_grid_proj():
    """The reference's fp16 grid_proj: fp16 gradients and AdamW state through the device-side lr / step-size words."""
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    model, batches = _setup(0.0, fp32_grid_proj=False)
    assert model.bert.grid_proj.weight.dtype == torch.float16
    ma, mb = copy.deepcopy(model), copy.deepcopy(model)
    ta, tb = PreTrainer(ma, default_opts(warmup_steps=10)), PreTrainer(mb, default_opts(warmup_steps=10))
    for _ in range(2):
        ta.train_step(batches["sap"], "sap")
    g = GraphedTrainStep(tb, batches["sap"], "sap")
    w0 = mb.bert.grid_proj.weight.detach().clone()
    for it in range(4):
        _compare_step(ta, tb, lambda: ta.train_step(batches["sap"], "sap"), g, ("fp16", it))
    assert not torch.equal(w0, mb.bert.grid_proj.weight)         # ... and it does train


def case_full_size():
    """The full-size pre-training twin at bench.py's training shape (161 M parameters, B = 32, L = 80, five steps of native
    12x49x768 observations): two replays of the sap graph against the eager step, each from a common state."""
    from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.synthetic import batch_to, make_pretrain_batch
    from gridmm_amd.train_graph import GraphedTrainStep
    from gridmm_amd.vilmodel import default_config
    dev = torch.device("cuda")
    cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=list(TASKS), image_prob_size=1000, obj_prob_size=0,
                         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = GlocalTextPathCMTPreTraining(cfg).to(dev)
    batch = batch_to(make_pretrain_batch(np.random.RandomState(2), 32, "sap", max_steps=5, L=80, vocab=30000,
                                         image_prob_size=1000, n_pts=(588 * 3, 588 * 5)), dev)
    ma, mb = copy.deepcopy(model), model
    ta, tb = PreTrainer(ma, default_opts(warmup_steps=100)), PreTrainer(mb, default_opts(warmup_steps=100))
    for _ in range(2):
        ta.train_step(batch, "sap")
    g = GraphedTrainStep(tb, batch, "sap")
    for it in range(2):
        _compare_step(ta, tb, lambda: ta.train_step(batch, "sap"), g, ("full", it))


def case_trajectory():
    """Graph replays against eager steps LEFT ALONE (no re-synchronisation of the two trainers): with no float atomics on
    the training path the two trajectories are bit-identical -- losses, gradient norms and parameters after 9 steps of the
    three tasks cycling, the reference's fp16 grid_proj included (round 2 had to compare every step from a common state)."""
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    model, batches = _setup(0.0, fp32_grid_proj=False)
    ma, mb = copy.deepcopy(model), copy.deepcopy(model)
    ta, tb = PreTrainer(ma, default_opts(warmup_steps=10)), PreTrainer(mb, default_opts(warmup_steps=10))
    graphs = {}
    for t in TASKS:
        for _ in range(2):
            ta.train_step(batches[t], t)
        graphs[t] = GraphedTrainStep(tb, batches[t], t)
    for i in range(9):
        t = TASKS[i % 3]
        la, na = ta.train_step(batches[t], t)
        lb, nb = graphs[t]()
        assert torch.equal(la, lb) and torch.equal(na, nb), (i, t, float((la - lb).abs().max()), float(na), float(nb))
    for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        assert torch.equal(pa, pb), n


def case_dropout():
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    model, batches = _setup(0.1)
    runs = []
    for _ in range(2):
        m = copy.deepcopy(model)
        tr = PreTrainer(m, default_opts(warmup_steps=10, learning_rate=0.0))     # lr 0: only the masks differ between replays
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        g = GraphedTrainStep(tr, batches["sap"], "sap")
        runs.append([g()[0].clone() for _ in range(3)])
    a, b = runs
    assert all(torch.isfinite(x).all() for x in a)
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[1], a[2])          # new dropout masks every replay
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-5)                       # same seeds -> same masks


if __name__ == "__main__":
    case = sys.argv[1]
    if case == "dropout":
        case_dropout()
    elif case == "fp16_grid_proj":
        case_fp16_grid_proj()
    elif case == "full_size":
        case_full_size()
    elif case == "two_graphs":
        case_two_graphs()
    elif case == "trajectory":
        case_trajectory()
    elif case.startswith("segments_"):
        # the backward as one graph per autograd segment (the multi-rank form), single process: 5 text layers -> a cut
        # inside the text encoder as well
        case_equals_eager(case.split("_", 1)[1], segments=True, layers=5)
    elif case == "dist2":
        case_dist2()
    elif case == "rccl1":
        case_rccl1()
    elif case == "alternate_full":
        case_alternate_full()
    elif case == "nonkernel":
        case_nonkernel()
    else:
        case_equals_eager(case)
    print("ok", case)
