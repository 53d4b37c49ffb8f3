"""GPU: gridmm_attention_rows_train / gridmm_attention_rows_bwd (csrc/attention_train.hip) -- the attention of the
differentiable path on the bf16 matrix pipe (3-term split, K / V or Q / dO staged in LDS by a loader wave) -- against torch
fp64 autograd of the reference's formula (map_nav_src/models/vilmodel.py:95-157: softmax(QK^T / 8 + mask), dropout on the
probabilities, P V) and, for dropout, against the exact-fp32 kernels with the SAME seed (same counter-hash mask)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda")


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _run(q_src, kv_src, cols, kmask, heads, dy, p=0.0, seed=0, shift=False):
    """Forward + backward through the C-ABI on strided column blocks of fused projections.  Returns (out, lse2, dq_src, dkv_src).
    shift: the K / V planes are relative to row 0 of their episode and the kernels get V[row 0] as vbar (what
    gridmm_linear_planes_shift produces on the differentiable path)."""
    from gridmm_amd import _lib, ops
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    H = heads * 64
    qc, kc, vc = cols
    B, Sq = q_src.shape[:2]
    Sk = kv_src.shape[1]
    H_ = heads * 64
    vbar, vbs = ctypes.c_void_p(0), 0
    if shift:
        row0 = kv_src[:, :1].clone()
        if kv_src is q_src:
            row0[..., cols[0]:cols[0] + H_] = 0            # the q block stays as it is
        shifted = kv_src - row0
        sh = row0[:, 0].contiguous()
        vbar, vbs = ctypes.c_void_p(sh.data_ptr() + 4 * cols[2]), sh.stride(0)
        qa = ops.split_rows(shifted if kv_src is q_src else q_src)
        ka = qa if kv_src is q_src else ops.split_rows(shifted)
    else:
        qa, ka = ops.split_rows(q_src), (ops.split_rows(kv_src) if kv_src is not q_src else None)
        ka = qa if ka is None else ka
    Sqp = (Sq + 15) // 16 * 16
    out = torch.empty(B, Sq, H, device=q_src.device)
    oh, ol = ops._planes_like(out.shape, out.device)
    lse = torch.full((B, heads, Sqp), float("nan"), device=q_src.device)
    km = None if kmask is None else kmask.view(torch.uint8)
    Wq, Wk = q_src.shape[-1], kv_src.shape[-1]

    def off(t, c):
        return ctypes.c_void_p(t.data_ptr() + 2 * c)
    rc = lib.gridmm_attention_rows_train(off(qa.hi, qc), off(qa.lo, qc), Sq * Wq, Wq, off(ka.hi, kc), off(ka.lo, kc), Sk * Wk, Wk,
                                         off(ka.hi, vc), off(ka.lo, vc), Sk * Wk, Wk, _p(km), Sk if km is not None else 0, _p(out),
                                         Sq * H, H, _p(oh), _p(ol), Sq * H, H, _p(lse), Sqp, vbar, vbs, B, heads, Sq, Sk, 0.125, float(p),
                                         seed, None, st)
    assert rc == 0, rc
    need = lib.gridmm_attention_rows_bwd_workspace(B, heads, Sq)
    ws = torch.empty(need, dtype=torch.uint8, device=q_src.device)
    dq_src = torch.zeros_like(q_src)
    dkv_src = dq_src if kv_src is q_src else torch.zeros_like(kv_src)

    def foff(t, c):
        return ctypes.c_void_p(t.data_ptr() + 4 * c)
    rc = lib.gridmm_attention_rows_bwd(off(qa.hi, qc), off(qa.lo, qc), Sq * Wq, Wq, off(ka.hi, kc), off(ka.lo, kc), Sk * Wk, Wk,
                                       off(ka.hi, vc), off(ka.lo, vc), Sk * Wk, Wk, _p(km), Sk if km is not None else 0, _p(out),
                                       Sq * H, H, _p(dy), Sq * H, H, _p(lse), vbar, vbs, _p(ws), need, foff(dq_src, qc), Sq * Wq, Wq,
                                       foff(dkv_src, kc), Sk * Wk, Wk, foff(dkv_src, vc), Sk * Wk, Wk, B, heads, Sq, Sk, Sqp,
                                       0.125, float(p), seed, None, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out, (oh, ol), lse, dq_src, dkv_src


def _ref(q, k, v, kmask, heads, keep=None, p=0.0):
    B, Sq, H = q.shape
    Sk = k.shape[1]
    qh = q.view(B, Sq, heads, 64).transpose(1, 2)
    kh = k.view(B, Sk, heads, 64).transpose(1, 2)
    vh = v.view(B, Sk, heads, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / 8.0
    if kmask is not None:
        s = s.masked_fill(~kmask[:, None, None, :], -float("inf"))
    pr = torch.softmax(s, -1)
    pr = torch.nan_to_num(pr, nan=0.0)                   # fully masked rows: zero output (the kernels' convention)
    if keep is not None:
        pr = pr * keep.to(pr.dtype) / (1.0 - p)
    return (pr @ vh).transpose(1, 2).reshape(B, Sq, H), s


@pytest.mark.parametrize("shift", [False, True])
@pytest.mark.parametrize("B,Sq,Sk,heads,same", [(2, 57, 296, 12, False), (3, 216, 216, 12, True), (2, 80, 80, 2, True),
                                                (1, 17, 45, 4, False), (2, 130, 33, 3, False), (1, 300, 300, 1, True)])
def test_forward_and_backward_match_fp64(B, Sq, Sk, heads, same, shift):
    dev = _dev()
    H = heads * 64
    g = torch.Generator().manual_seed(Sq * 3 + Sk)
    lens = torch.randint(max(1, Sk // 3), Sk + 1, (B,), generator=g)
    lens[0] = Sk
    kmask = (torch.arange(Sk)[None] < lens[:, None]).to(dev)
    if B > 1:
        kmask[1, 3:9] = False                            # holes inside the valid range too
    dy = torch.randn(B, Sq, H, generator=g).to(dev)
    if same:
        qkv = torch.randn(B, Sq, 3 * H, generator=g).to(dev)
        out, (oh, ol), lse, dsrc, _ = _run(qkv, qkv, (0, H, 2 * H), kmask, heads, dy, shift=shift)
        qd = qkv.double().requires_grad_()
        yd, s = _ref(qd[..., :H], qd[..., H:2 * H], qd[..., 2 * H:], kmask, heads)
        yd.backward(dy.double())
        assert _rel(out, yd) < 2e-5
        assert _rel(dsrc, qd.grad) < 4e-5, _rel(dsrc, qd.grad)
    else:
        q = torch.randn(B, Sq, H, generator=g).to(dev)
        kv = torch.randn(B, Sk, 4 * H, generator=g).to(dev)
        out, (oh, ol), lse, dq, dkv = _run(q, kv, (0, 2 * H, 3 * H), kmask, heads, dy, shift=shift)
        qd, kvd = q.double().requires_grad_(), kv.double().requires_grad_()
        yd, s = _ref(qd, kvd[..., 2 * H:3 * H], kvd[..., 3 * H:], kmask, heads)
        yd.backward(dy.double())
        assert _rel(out, yd) < 2e-5
        assert _rel(dq, qd.grad) < 4e-5, _rel(dq, qd.grad)
        assert _rel(dkv, kvd.grad) < 4e-5, _rel(dkv, kvd.grad)
        assert float(dkv[..., :2 * H].abs().max()) == 0.0          # the other layers' column blocks are not touched
    # planes of the output and the saved statistic
    assert float((oh.float() + ol.float() - out).abs().max()) < 1e-4
    if not shift:        # (with shifted K the statistic belongs to the shifted scores s - <q, k_row0>: the probabilities are the same)
        want = torch.logsumexp(s.detach(), -1) / torch.log(torch.tensor(2.0, dtype=torch.float64))       # log2 domain
        assert float((lse[:, :, :Sq].double() - want).abs().max()) < 1e-4


def test_fully_masked_episode_gives_zero_output_and_zero_gradients():
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(2, 20, 3 * 128, generator=g).to(dev)
    kmask = torch.ones(2, 20, dtype=torch.bool, device=dev)
    kmask[1] = False
    dy = torch.ones(2, 20, 128, device=dev)
    out, _, lse, d, _ = _run(qkv, qkv, (0, 128, 256), kmask, 2, dy)
    assert torch.isfinite(out).all() and torch.isfinite(d).all()
    assert float(out[1].abs().max()) == 0.0 and float(d[1].abs().max()) == 0.0
    assert float(lse[1, :, :20].min()) > 1e29


@pytest.mark.parametrize("shift", [False, True])
@pytest.mark.parametrize("B,Sq,Sk,heads", [(2, 57, 296, 12), (2, 100, 100, 4)])
def test_dropout_mask_equals_the_fp32_kernels_and_gradients_match_fp64(B, Sq, Sk, heads, shift):
    """Dropout on the probabilities: the keep-mask is the counter hash of (seed, b, h, q, k) restated on the host
    (autograd.attention_dropout_mask, the same restatement the fp32 kernels are pinned by)."""
    from gridmm_amd import autograd as ag
    dev = _dev()
    H, p, seed = heads * 64, 0.1, 123456789
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, Sq, H, generator=g).to(dev)
    kv = torch.randn(B, Sk, 2 * H, generator=g).to(dev)
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    kmask = (torch.arange(Sk)[None] < lens[:, None]).to(dev)
    dy = torch.randn(B, Sq, H, generator=g).to(dev)
    out, _, lse, dq, dkv = _run(q, kv, (0, 0, H), kmask, heads, dy, p=p, seed=seed, shift=shift)
    keep = torch.from_numpy(ag.attention_dropout_mask(seed, B, heads, Sq, Sk, p)).to(dev)
    qd, kvd = q.double().requires_grad_(), kv.double().requires_grad_()
    yd, _ = _ref(qd, kvd[..., :H], kvd[..., H:], kmask, heads, keep=keep, p=p)
    yd.backward(dy.double())
    assert _rel(out, yd) < 2e-5, _rel(out, yd)
    assert _rel(dq, qd.grad) < 4e-5 and _rel(dkv, kvd.grad) < 4e-5
    out2, _, _, dq2, dkv2 = _run(q, kv, (0, 0, H), kmask, heads, dy, p=p, seed=seed, shift=shift)
    assert torch.equal(out, out2) and torch.equal(dq, dq2) and torch.equal(dkv, dkv2)      # run-to-run bit equality


def test_shift_removes_the_common_component_error():
    """K / V rows that share a large common component (LayerNorm bias, type embeddings -- the normal case): without the shift the
    bf16x3 products carry its rounding error into dS = P o (dP - delta) and the q / k gradients are an order of magnitude less
    accurate than with it (the reason gridmm_linear_planes_shift exists; measured on the full pre-training model by
    tools/dbg_pretrain_grad_errors.py)."""
    dev = _dev()
    B, S, heads = 2, 96, 4
    H = heads * 64
    g = torch.Generator().manual_seed(3)
    common = 20.0 * torch.randn(1, 1, 3 * H, generator=g)
    qkv = (torch.randn(B, S, 3 * H, generator=g) + common).to(dev)
    qkv[..., :H] -= common[..., :H].to(dev)              # (queries without it: the scores stay O(1))
    qkv[..., H:2 * H] = 0.05 * qkv[..., H:2 * H] + 0.95 * common[..., H:2 * H].to(dev) * 0.05
    kmask = torch.ones(B, S, dtype=torch.bool, device=dev)
    dy = torch.randn(B, S, H, generator=g).to(dev)
    qd = qkv.double().requires_grad_()
    yd, _ = _ref(qd[..., :H], qd[..., H:2 * H], qd[..., 2 * H:], kmask, heads)
    yd.backward(dy.double())
    errs = {}
    for shift in (False, True):
        out, _, _, d, _ = _run(qkv, qkv, (0, H, 2 * H), kmask, heads, dy, shift=shift)
        errs[shift] = (_rel(d[..., :2 * H], qd.grad[..., :2 * H]), _rel(out, yd))
    assert errs[True][0] < 1e-4 and errs[True][1] < 2e-5, errs
    assert errs[True][0] < 0.3 * errs[False][0], errs
