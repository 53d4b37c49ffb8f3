"""GPU: backward kernels (gridmm_amd.autograd) against torch autograd of the same op in fp32/fp64.

Tolerances: GEMM-backed ops run on MFMA bf16 with the 3-term hi/lo split (~1e-5 relative); LayerNorm /
GELU / attention backward are fp32.  Stated per test.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("M,K,N,bias,res", [(300, 768, 768, True, False), (57, 768, 3072, True, True),
                                            (130, 3072, 768, True, False), (40, 7, 768, True, False),
                                            (64, 14, 64, True, False), (33, 768, 1, True, False),
                                            (196 * 2, 64, 96, False, False), (5, 5, 32, True, False)])
def test_linear_backward(M, K, N, bias, res):
    from gridmm_amd import autograd as ag
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + K)
    x = torch.randn(3, M, K, generator=g).to(dev).requires_grad_()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).requires_grad_()
    b = (torch.randn(N, generator=g) * 0.1).to(dev).requires_grad_() if bias else None
    r = torch.randn(3, M, N, generator=g).to(dev).requires_grad_() if res else None
    dy = torch.randn(3, M, N, generator=g).to(dev)
    y = ag.linear(x, w, b, r)
    y.backward(dy)
    got = [t.grad.clone() for t in (x, w, b, r) if t is not None]
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    bd = b.detach().double().requires_grad_() if bias else None
    rd = r.detach().double().requires_grad_() if res else None
    yd = torch.nn.functional.linear(xd, wd, bd)
    if res:
        yd = yd + rd
    yd.backward(dy.double())
    want = [t.grad for t in (xd, wd, bd, rd) if t is not None]
    assert _rel(y, yd) < 2e-5
    for a, e in zip(got, want):
        assert a.shape == e.shape
        assert _rel(a, e) < 3e-5, (a.shape, _rel(a, e))


@pytest.mark.parametrize("M,H,res", [(37, 768, True), (1000, 768, False), (9, 64, True), (130, 1024, False)])
def test_layernorm_backward(M, H, res):
    from gridmm_amd import autograd as ag
    dev = _dev()
    g = torch.Generator().manual_seed(M + H)
    x = torch.randn(2, M, H, generator=g).to(dev).requires_grad_()
    r = torch.randn(2, M, H, generator=g).to(dev).requires_grad_() if res else None
    ln = torch.nn.LayerNorm(H, eps=1e-12).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(H, generator=g))
        ln.bias.copy_(0.1 * torch.randn(H, generator=g))
    dy = torch.randn(2, M, H, generator=g).to(dev)
    y = ag.layer_norm(x, ln, r)
    y.backward(dy)
    got = [x.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()] + ([r.grad.clone()] if res else [])
    ln.zero_grad()
    lnd = torch.nn.LayerNorm(H, eps=1e-12).to(dev).double()
    lnd.load_state_dict({k: v.double() for k, v in ln.state_dict().items()})
    xd = x.detach().double().requires_grad_()
    rd = r.detach().double().requires_grad_() if res else None
    yd = lnd(xd + rd if res else xd)
    yd.backward(dy.double())
    want = [xd.grad, lnd.weight.grad, lnd.bias.grad] + ([rd.grad] if res else [])
    assert _rel(y, yd) < 1e-5
    for a, e in zip(got, want):
        assert _rel(a, e) < 2e-5, _rel(a, e)


def test_activation_backward():
    from gridmm_amd import autograd as ag
    dev = _dev()
    x = (torch.randn(7, 33, 3072, generator=torch.Generator().manual_seed(1)) * 2).to(dev).requires_grad_()
    dy = torch.randn(7, 33, 3072, generator=torch.Generator().manual_seed(2)).to(dev)
    for fn, ref in ((ag.gelu, lambda t: t * 0.5 * (1.0 + torch.erf(t / math.sqrt(2.0)))), (ag.relu, torch.relu)):
        x.grad = None
        y = fn(x)
        y.backward(dy)
        xd = x.detach().double().requires_grad_()
        yd = ref(xd)
        yd.backward(dy.double())
        assert _rel(y, yd) < 1e-6
        assert _rel(x.grad, xd.grad) < 1e-6


def _ref_attention(q, k, v, kmask, heads):
    B, Sq, H = q.shape
    Sk = k.shape[1]
    qh = q.view(B, Sq, heads, 64).transpose(1, 2)
    kh = k.view(B, Sk, heads, 64).transpose(1, 2)
    vh = v.view(B, Sk, heads, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / 8.0
    if kmask is not None:
        s = s.masked_fill(~kmask[:, None, None, :], -float("inf"))
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Sq, H)


@pytest.mark.parametrize("B,Sq,Sk,heads,same", [(2, 57, 296, 12, False), (3, 216, 216, 12, True),
                                                (2, 80, 80, 2, True), (1, 17, 45, 4, False)])
def test_attention_backward(B, Sq, Sk, heads, same):
    from gridmm_amd import autograd as ag
    dev = _dev()
    H = heads * 64
    g = torch.Generator().manual_seed(Sq * 3 + Sk)
    lens = torch.randint(max(1, Sk // 3), Sk + 1, (B,), generator=g)
    lens[0] = Sk
    kmask = (torch.arange(Sk)[None] < lens[:, None]).to(dev)
    dy = torch.randn(B, Sq, H, generator=g).to(dev)
    if same:
        qkv = torch.randn(B, Sq, 3 * H, generator=g).to(dev).requires_grad_()
        y = ag.self_attention(qkv, kmask, heads)
        y.backward(dy)
        qd = qkv.detach().double().requires_grad_()
        yd = _ref_attention(qd[..., :H], qd[..., H:2 * H], qd[..., 2 * H:], kmask, heads)
        yd.backward(dy.double())
        assert _rel(y, yd) < 1e-5
        assert _rel(qkv.grad, qd.grad) < 2e-5, _rel(qkv.grad, qd.grad)
    else:
        q = torch.randn(B, Sq, H, generator=g).to(dev).requires_grad_()
        kv = torch.randn(B, Sk, 4 * H, generator=g).to(dev).requires_grad_()   # two layers' [k|v]; use layer 1
        y = ag.cross_attention(q, kv, kmask, heads, kv_col=2 * H)
        y.backward(dy)
        qd, kvd = q.detach().double().requires_grad_(), kv.detach().double().requires_grad_()
        yd = _ref_attention(qd, kvd[..., 2 * H:3 * H], kvd[..., 3 * H:], kmask, heads)
        yd.backward(dy.double())
        assert _rel(y, yd) < 1e-5
        assert _rel(q.grad, qd.grad) < 2e-5
        assert _rel(kv.grad, kvd.grad) < 2e-5
        assert float(kv.grad[..., :2 * H].abs().max()) == 0.0


def test_attention_backward_fully_masked_row_batch():
    """An episode whose keys are all masked produces zero output and zero gradients (no NaN)."""
    from gridmm_amd import autograd as ag
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(2, 20, 3 * 128, generator=g).to(dev).requires_grad_()
    kmask = torch.ones(2, 20, dtype=torch.bool, device=dev)
    kmask[1] = False
    y = ag.self_attention(qkv, kmask, 2)
    y.sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(qkv.grad).all()
    assert float(y[1].abs().max()) == 0.0 and float(qkv.grad[1].abs().max()) == 0.0


@pytest.mark.parametrize("B,D,L,n_obs", [(2, 768, 40, 3), (3, 512, 80, 2), (2, 256, 20, 2),
                                          (2, 512, 200, 2), (2, 768, 144, 1),     # L > 128: streamed text fragments
                                          (2, 512, 96, 1), (2, 512, 112, 1), (2, 512, 128, 1), (1, 768, 96, 1),
                                          (2, 768, 128, 1), (2, 256, 70, 1),   # every wave-assignment case (Lt 5..8)
                                          (2, 512, 80, 12), (1, 768, 64, 5),     # thousands of points: the chunked gather pass
                                          (2, 512, 80, 30), (1, 512, 80, 60)])   # deep memories: 16 / 32 chunks per episode
def test_grid_aggregate_backward(B, D, L, n_obs):
    """d cells / d text_fts against torch autograd of the reference formulation (vilmodel.py:797-807)."""
    from gridmm_amd import autograd as ag, ops
    dev = _dev()
    rs = np.random.RandomState(B * 100 + D)
    cap = 588 * n_obs
    n_pts = rs.randint(cap // 2, cap + 1, size=B)
    slab = torch.from_numpy((rs.standard_normal((B, cap, D)) * 0.35).astype(np.float16)).to(dev)
    ids = rs.randint(-1, 196, size=(B, cap)).astype(np.int16)
    ids[:, :50] = rs.randint(0, 4, size=(B, 50))   # a few crowded cells
    for b in range(B):
        ids[b, n_pts[b]:] = -1
    ids_t = torch.from_numpy(ids).to(dev)
    perm = torch.empty(B, cap, dtype=torch.int32, device=dev)
    cell_start = torch.empty(B, 198, dtype=torch.int32, device=dev)
    ops.grid_sort_ids(ids_t, torch.from_numpy(n_pts.astype(np.int32)).to(dev), perm, cell_start)
    text = (torch.from_numpy(rs.standard_normal((B, L, D)).astype(np.float32)) * 0.3).to(dev).requires_grad_()
    dcells = torch.from_numpy(rs.standard_normal((B, 196, D)).astype(np.float32)).to(dev)
    cells, occ = ag.grid_aggregate(text, slab, perm, cell_start)
    cells.backward(dcells)

    td = text.detach().double().requires_grad_()
    ref = torch.zeros(B, 196, D, dtype=torch.float64, device=dev)
    for b in range(B):
        x = slab[b].double()
        w = (x @ td[b].t()).max(-1)[0]
        for c in range(196):
            sel = ids_t[b] == c
            if sel.any():
                ref[b, c] = (torch.softmax(w[sel], 0)[:, None] * x[sel]).sum(0)
    ref.backward(dcells.double())
    assert _rel(cells, ref) < 1e-4
    assert _rel(text.grad, td.grad) < 1e-3, _rel(text.grad, td.grad)


def test_attention_probability_dropout_forward_and_backward():
    """Dropout on softmax(QK^T) (vilmodel.py:143): the kernels' hash mask, restated on the host, applied in a torch
    reference -> same output and gradients; keep rate ~ 1 - p; p = 0 is the identity."""
    from gridmm_amd import autograd as ag
    dev = _dev()
    B, S, heads, p = 2, 75, 3, 0.25
    H = heads * 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B, S, 3 * H, generator=g).to(dev).requires_grad_()
    kmask = torch.ones(B, S, dtype=torch.bool, device=dev)
    kmask[1, 60:] = False
    dy = torch.randn(B, S, H, generator=g).to(dev)
    torch.manual_seed(1234)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # what _Attention.forward will draw
    torch.manual_seed(1234)
    y = ag.self_attention(qkv, kmask, heads, dropout_p=p)
    y.backward(dy)
    keep = torch.from_numpy(ag.attention_dropout_mask(seed, B, heads, S, S, p)).to(dev)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
    qd = qkv.detach().double().requires_grad_()
    q, k, v = (qd[..., i * H:(i + 1) * H].view(B, S, heads, 64).transpose(1, 2) for i in range(3))
    s = (q @ k.transpose(-1, -2) / 8.0).masked_fill(~kmask[:, None, None, :], -float("inf"))
    pr = torch.softmax(s, -1) * keep / (1 - p)
    yd = (pr @ v).transpose(1, 2).reshape(B, S, H)
    yd.backward(dy.double())
    assert _rel(y, yd) < 1e-5
    assert _rel(qkv.grad, qd.grad) < 2e-5


@pytest.mark.parametrize("M,N,K", [(6912, 768, 7), (1824, 768, 14), (37, 768, 7), (300, 64, 5), (1, 8, 16)])
def test_skinny_linear_matches_fp64(M, N, K):
    """gridmm_linear_skinny / _bwd (the K <= 16 position / angle embedding Linears of the differentiable path,
    map_nav_src/models/vilmodel.py:454-470, 538-552, 640-655) vs fp64: forward, dW, db; two runs bit-identical."""
    import torch
    from gridmm_amd import autograd as ag
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    w = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.3).to(dev))
    b = torch.nn.Parameter(torch.randn(N, generator=g).to(dev))
    dy = torch.randn(M, N, generator=g).to(dev)
    outs = []
    for _ in range(2):
        w.grad = b.grad = None
        y = ag.linear(x, w, b)
        y.backward(dy)
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), w.grad.clone(), b.grad.clone()))
    y, dw, db = outs[0]
    assert all(torch.equal(a, c) for a, c in zip(outs[0], outs[1]))
    yr = x.double() @ w.detach().double().t() + b.detach().double()
    assert float((y.double() - yr).abs().max()) <= 1e-5 * max(1.0, float(yr.abs().max()))
    dwr, dbr = dy.double().t() @ x.double(), dy.double().sum(0)
    assert float((dw.double() - dwr).abs().max()) <= 2e-5 * max(1.0, float(dwr.abs().max()))
    assert float((db.double() - dbr).abs().max()) <= 2e-5 * max(1.0, float(dbr.abs().max()))


@pytest.mark.parametrize("M,K", [(1824, 768), (57, 768), (1, 64), (130, 256)])
def test_rowdot_linear_matches_fp64(M, K):
    """gridmm_rowdot / _bwd (nn.Linear(K, 1): the last layer of the heads' ClsPrediction, vilmodel.py:437-446) vs fp64:
    forward, dX, dw, db; two runs bit-identical."""
    import torch
    from gridmm_amd import autograd as ag
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_()
    w = torch.nn.Parameter((torch.randn(1, K, generator=g) * 0.3).to(dev))
    b = torch.nn.Parameter(torch.randn(1, generator=g).to(dev))
    dy = torch.randn(M, 1, generator=g).to(dev)
    outs = []
    for _ in range(2):
        w.grad = b.grad = x.grad = None
        y = ag.linear(x, w, b)
        assert y.shape == (M, 1)
        y.backward(dy)
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()))
    assert all(torch.equal(a, c) for a, c in zip(outs[0], outs[1]))
    y, dx, dw, db = outs[0]
    xd, wd = x.detach().double(), w.detach().double()
    assert float((y.double() - (xd @ wd.t() + b.detach().double())).abs().max()) <= 1e-4
    assert float((dx.double() - dy.double() @ wd).abs().max()) <= 1e-5
    assert float((dw.double() - dy.double().t() @ xd).abs().max()) <= 2e-5 * max(1.0, float((dy.double().t() @ xd).abs().max()))
    assert float((db.double() - dy.double().sum()).abs().max()) <= 1e-4
