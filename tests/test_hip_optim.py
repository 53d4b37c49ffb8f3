"""GPU: fused clip + AdamW step (gridmm_amd.optim) against the algorithm of pretrain_src/optim/adamw.py:56-112
restated with torch ops, and the pre-training step loop."""
import json
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_adamw(p, g, state, lr, b1, b2, eps, wd, decay_first):
    state["step"] += 1
    state["m"].mul_(b1).add_(g, alpha=1 - b1)
    state["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = state["v"].sqrt().add_(eps * (math.sqrt(1 - b2 ** state["step"]) if decay_first else 1.0))   # torch vs HF eps
    step_size = lr * math.sqrt(1 - b2 ** state["step"]) / (1 - b1 ** state["step"])
    if decay_first and wd > 0:
        p.mul_(1 - lr * wd)
    p.addcdiv_(state["m"], denom, value=-step_size)
    if not decay_first and wd > 0:
        p.add_(p, alpha=-lr * wd)


@pytest.mark.parametrize("decay_first", [False, True])
def test_fused_clip_adamw_matches_reference_algorithm(decay_first):
    from gridmm_amd.optim import AdamW
    g = torch.Generator().manual_seed(0)
    shapes = [(768, 768), (3072,), (5, 7), (1,)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    ref = [p.detach().double().clone() for p in params]
    st = [dict(step=0, m=torch.zeros_like(r), v=torch.zeros_like(r)) for r in ref]
    opt = AdamW([{"params": params[:2], "weight_decay": 0.01}, {"params": params[2:], "weight_decay": 0.0}],
                lr=1e-3, betas=(0.9, 0.98), decay_first=decay_first)
    for it in range(4):
        grads = [torch.randn(s, generator=g).cuda() * (10.0 if it % 2 else 0.01) for s in shapes]
        for p, gr in zip(params, grads):
            p.grad = gr.clone()
        v0 = params[0]._version
        norm = opt.step(max_grad_norm=5.0)
        assert params[0]._version > v0                          # caches keyed on the version counter will re-pack
        total = math.sqrt(sum(float((gr.double() ** 2).sum()) for gr in grads))
        assert abs(float(norm) - total) < 1e-4 * total
        coef = min(1.0, 5.0 / (total + 1e-6))
        for r, s, gr, wd in zip(ref, st, grads, (0.01, 0.01, 0.0, 0.0)):
            _ref_adamw(r, gr.double() * coef, s, 1e-3, 0.9, 0.98, 1e-6, wd, decay_first)
        for p, r in zip(params, ref):
            assert float((p.detach().double() - r).abs().max()) < 2e-6


def test_torch_adamw_semantics_for_the_finetune_optimizer():
    """decay_first=True reproduces torch.optim.AdamW (agent_base.py:131 'adamW') step for step."""
    from gridmm_amd.optim import AdamW
    g = torch.Generator().manual_seed(3)
    p1 = torch.nn.Parameter(torch.randn(300, 257, generator=g).cuda())
    p2 = torch.nn.Parameter(p1.detach().clone())
    mine = AdamW([p1], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, decay_first=True)
    ref = torch.optim.AdamW([p2], lr=1e-3)
    for _ in range(5):
        gr = torch.randn(300, 257, generator=g).cuda()
        p1.grad, p2.grad = gr.clone(), gr.clone()
        mine.step()
        ref.step()
        assert float((p1.detach() - p2.detach()).abs().max()) < 2e-6


def test_fp16_parameter_keeps_fp16_state():
    from gridmm_amd.optim import AdamW
    p = torch.nn.Parameter(torch.randn(64, 64, generator=torch.Generator().manual_seed(1)).cuda().half())
    before = p.detach().clone()
    opt = AdamW([p], lr=1e-2)
    p.grad = torch.randn(64, 64, generator=torch.Generator().manual_seed(2)).cuda().half()
    opt.step()
    assert opt.state[p]["exp_avg"].dtype == torch.float16
    d = (p.detach().float() - before.float())
    assert torch.isfinite(d).all() and float(d.abs().max()) > 1e-3   # first Adam step moves every entry by ~lr


def test_pretraining_steps_reduce_loss_and_follow_schedule():
    from conftest import load_golden
    from oracle import gen_golden
    from gridmm_amd.pretrain_loop import PreTrainer, TaskSampler, default_opts
    from gridmm_amd.synthetic import batch_to
    import test_hip_pretrain as TP
    fx = load_golden("pretrain_reduced.npz")
    model = TP._model(fx)
    opts = default_opts(learning_rate=5e-5, warmup_steps=2, num_train_steps=40)
    tr = PreTrainer(model, opts)
    sampler = TaskSampler(opts.tasks, opts.mix_ratio, seed=0)
    seq = [sampler.next_task() for _ in range(30)]
    assert set(seq) == {"mlm", "mrc", "sap"}
    batches = {t: batch_to(gen_golden.pretrain_batch(t), "cuda") for t in opts.tasks}
    first, last = {}, {}
    for it in range(18):
        task = opts.tasks[it % 3]
        losses, norm = tr.train_step(batches[task], task)
        assert torch.isfinite(losses).all() and torch.isfinite(norm)
        first.setdefault(task, float(losses.mean()))
        last[task] = float(losses.mean())
    assert tr.global_step == 18
    assert abs(tr.optimizer.param_groups[0]["lr"] - 5e-5 * (40 - 18) / (40 - 2)) < 1e-12     # warmup_linear
    assert sum(last.values()) < sum(first.values()), (first, last)       # task-mixed steps on three fixed batches


def test_optimizer_trajectory_matches_reference_golden():
    """tests/golden/optim_reduced.npz: the reference's build_optimizer / get_lr_sched / AdamW.step (pretrain_src/optim/
    misc.py:12-37, sched.py:17-30, adamw.py:56-112) driven as train_r2r.py:266-296 over 7 steps -- decay / no-decay
    groups by parameter name, warm-up then linear decay, clipping active on some steps only, a parameter without
    gradients, and an fp16 parameter with fp16 optimizer state (incl. a step whose g^2 underflows fp16)."""
    from conftest import load_golden
    from oracle import gen_golden as GG
    from gridmm_amd.optim import build_optimizer, get_lr_sched
    from types import SimpleNamespace
    fx = load_golden("optim_reduced.npz")
    o = json.loads(str(fx["cfg"]))
    model = GG.OptimToy().cuda()
    names = json.loads(str(fx["names"]))
    assert [n for n, _ in model.named_parameters()] == names
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(fx["init." + n]).to(p.dtype))
    opts = SimpleNamespace(optim="adamw", learning_rate=o["learning_rate"], betas=o["betas"], weight_decay=o["weight_decay"],
                           warmup_steps=o["warmup_steps"], num_train_steps=o["num_train_steps"])
    opt = build_optimizer(model, opts)
    groups = [[n for n, p in model.named_parameters() if any(p is q for q in g["params"])] for g in opt.param_groups]
    assert groups == json.loads(str(fx["decay"]))
    assert [g["weight_decay"] for g in opt.param_groups] == [o["weight_decay"], 0.0]
    for step in range(1, o["steps"] + 1):
        lr = get_lr_sched(step, opts)
        assert abs(lr - float(fx["lr"][step - 1])) < 1e-15
        for g in opt.param_groups:
            g["lr"] = lr
        for n, p in model.named_parameters():
            if not n.startswith("unused"):
                p.grad = GG.optim_toy_grad(n, p.shape, step).to(p.dtype).cuda()
        norm = opt.step(max_grad_norm=o["grad_norm"])
        opt.zero_grad()
        want_norm = float(fx["grad_norm"][step - 1])
        assert abs(float(norm) - want_norm) < 2e-4 * want_norm, (step, float(norm), want_norm)
        for n, p in model.named_parameters():
            want = torch.from_numpy(fx["step%d.%s" % (step, n)])
            got = p.detach().float().cpu()
            if p.dtype == torch.float16:     # fp16 parameter / state: within two fp16 roundings of the reference
                assert float((got - want).abs().max()) <= 2.0 ** -9 * float(want.abs().max()), (step, n)
            else:
                assert float((got - want).abs().max()) < 2e-6, (step, n, float((got - want).abs().max()))
    assert all(torch.equal(p.detach().float().cpu(), torch.from_numpy(fx["init." + n]))
               for n, p in model.named_parameters() if n.startswith("unused"))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_single_tensor_entry_points_equal_the_multi_tensor_step(dtype):
    """gridmm_grad_sumsq / gridmm_adamw_step (one tensor per call; lr / step_size / eps as arguments or, `dyn`, from
    device memory) against AdamW.step (the multi-tensor table) on the same tensor: the same update (the two gradient norms are summed
    in different orders, so the clip scale may differ in its last bit); arguments vs `dyn`: identical bits."""
    import ctypes
    from gridmm_amd import _lib
    from gridmm_amd.optim import AdamW
    from gridmm_amd.ops import _p, _stream
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    p0 = (torch.randn(1000, 37, generator=g) * 0.05).to(dtype).cuda()
    grads = [(torch.randn(1000, 37, generator=g) * s).to(dtype).cuda() for s in (0.01, 3.0, 1e-4)]
    pa = torch.nn.Parameter(p0.clone())
    opt = AdamW([pa], lr=2e-4, betas=(0.9, 0.98), weight_decay=0.01)
    pb, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pc, mc, vc = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    b1, b2, eps, wd, max_norm = 0.9, 0.98, 1e-6, 0.01, 1.0
    for t, gr in enumerate(grads, 1):
        pa.grad = gr.clone()
        na = opt.step(max_grad_norm=max_norm)
        ss = 2e-4 * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        sumsq = torch.zeros(1, device="cuda")      # once for both variants: its atomics sum in a run-dependent order
        assert lib.gridmm_grad_sumsq(_p(gr), gr.numel(), int(dtype == torch.float16), _p(sumsq), _stream()) == 0
        for (p, mm, vv, use_dyn) in ((pb, m, v, False), (pc, mc, vc, True)):
            dyn = torch.tensor([2e-4, ss, eps], device="cuda") if use_dyn else None
            lr_a, ss_a, eps_a = (123.0, 456.0, 789.0) if use_dyn else (2e-4, ss, eps)          # ignored when dyn is given
            assert lib.gridmm_adamw_step(_p(p), _p(gr), _p(mm), _p(vv), p.numel(), int(dtype == torch.float16), lr_a, b1,
                                         b2, eps_a, wd, ss_a, 0, _p(sumsq), max_norm, _p(dyn), _stream()) == 0
            assert abs(float(sumsq.sqrt()) - float(na)) <= 1e-5 * float(na)
        torch.cuda.synchronize()
        tol = 1e-7 if dtype == torch.float32 else 6.2e-5    # fp16: one ulp at |w| < 0.0625 .. 0.125
        assert float((pa.detach().float() - pb.float()).abs().max()) <= tol
        assert torch.equal(pb, pc) and torch.equal(m, mc) and torch.equal(v, vc)
