"""GPU: the training row-wise kernels (csrc/train_rowops.hip) -- backward of the logit fusion, backward of the cell
compaction, hidden-state dropout -- against torch autograd over the plain-torch restatements of the same ops."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fusion_inputs(B, G, V, seed):
    g = torch.Generator().manual_seed(seed)
    g_raw, l_raw, gr_raw = torch.randn(B, G, generator=g), torch.randn(B, V, generator=g), torch.randn(B, G, generator=g)
    f_raw = torch.randn(B, generator=g)
    gm = torch.rand(B, G, generator=g) < 0.85
    gm[:, 0] = True
    gv = (torch.rand(B, G, generator=g) < 0.3) & gm
    gv[:, 0] = False
    vn = torch.rand(B, V, generator=g) < 0.6
    vn[:, 0] = True
    con = torch.full((B, G), -2, dtype=torch.int32)
    cv = torch.zeros(B, V, dtype=torch.uint8)
    for b in range(B):
        cands = [k for k in range(1, V) if vn[b, k]]
        free = list(cands)
        for k in cands[: len(cands) // 3]:
            cv[b, k] = 1                      # candidates that are already-visited nodes
            free.remove(k)
        for j in range(1, G):
            if gm[b, j] and not gv[b, j]:
                con[b, j] = free.pop() if (free and torch.rand(1, generator=g).item() < 0.5) else -1
    return g_raw, l_raw, gr_raw, f_raw, gm, gv, vn, con, cv


@pytest.mark.parametrize("with_fuse", [True, False])
def test_fuse_logits_forward_backward_match_torch_autograd(with_fuse):
    from gridmm_amd import vilmodel_train as VT
    B, G, V = 6, 11, 9
    g_raw, l_raw, gr_raw, f_raw, gm, gv, vn, con, cv = _fusion_inputs(B, G, V, 3)
    if not with_fuse:
        f_raw = None
    wts = [torch.randn(B, n) for n in (G, V, G, G)]      # a smooth probe: sum of softmax-free weighted finite logits

    def run(dev):
        leaves = [t.clone().to(dev).requires_grad_(True) for t in (g_raw, l_raw, gr_raw)] + \
                 ([f_raw.clone().to(dev).requires_grad_(True)] if with_fuse else [None])
        outs = VT.fuse_logits(leaves[0], leaves[1], leaves[2], leaves[3], gm.to(dev), gv.to(dev), vn.to(dev), con.to(dev), cv.to(dev))
        loss = 0
        for o, w in zip(outs, wts):
            fin = torch.isfinite(o)
            loss = loss + (torch.where(fin, o, torch.zeros_like(o)) * w.to(dev)).sum()
        loss.backward()
        return [o.detach().cpu() for o in outs], [None if l is None else l.grad.cpu() for l in leaves]

    want_o, want_g = run("cpu")            # plain torch expression (CPU branch of VT.fuse_logits)
    got_o, got_g = run("cuda")             # gridmm_fuse_logits + gridmm_fuse_logits_bwd
    for a, w in zip(got_o, want_o):
        f = torch.isfinite(w)
        assert torch.equal(f, torch.isfinite(a)) and (a[f] - w[f]).abs().max() < 1e-6
    for a, w in zip(got_g, want_g):
        if w is None:
            assert a is None
        else:
            assert (a - w).abs().max() < 1e-5, (a, w)


def test_cells_compact_backward_scatters_rows_back():
    from gridmm_amd import autograd as ag
    torch.manual_seed(0)
    B, H = 4, 768
    proj = torch.randn(B, 196, H, device="cuda", requires_grad=True)
    pos = torch.randn(B, 196, H, device="cuda", requires_grad=True)
    occ = (torch.rand(B, 196, device="cuda") < 0.5).to(torch.uint8)
    occ[1] = 0
    occ[2] = 1
    out, mask = ag.cells_compact(proj, pos, occ)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    # torch restatement: occupied cells first in cell order, zeros behind
    p2, q2 = proj.detach().clone().requires_grad_(True), pos.detach().clone().requires_grad_(True)
    x = p2 + q2
    ob = occ.bool()
    order = torch.argsort((~ob).to(torch.uint8), dim=1, stable=True)
    xs = x.gather(1, order.unsqueeze(-1).expand(-1, -1, H)) * (torch.arange(196, device="cuda")[None] < ob.sum(1)[:, None]).unsqueeze(-1)
    (xs * w).sum().backward()
    assert torch.equal(out, xs.detach())
    assert torch.equal(proj.grad, p2.grad) and torch.equal(pos.grad, q2.grad)
    assert int(mask[1].sum()) == 0 and int(mask[2].sum()) == 196


def test_dropout_mask_is_the_documented_hash_and_the_backward_reuses_it():
    from gridmm_amd import autograd as ag
    torch.manual_seed(11)
    x = torch.randn(37, 768, device="cuda", requires_grad=True)
    p = 0.1
    state = torch.random.get_rng_state()
    y = ag.dropout(x, p)
    torch.random.set_rng_state(state)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())              # the seed the call drew
    keep = torch.from_numpy(ag.dropout_mask(seed, x.numel(), p)).view_as(x).cuda()
    assert torch.equal(y != 0, keep | (x.detach() == 0) & False) or torch.equal((y != 0), keep & (x.detach() != 0))
    assert torch.allclose(y[keep], x.detach()[keep] / (1 - p), rtol=1e-6)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
    w = torch.randn_like(x)
    (y * w).sum().backward()
    assert torch.allclose(x.grad, torch.where(keep, w / (1 - p), torch.zeros_like(w)), rtol=1e-6)
    y2 = ag.dropout(x, p)                                            # a new call draws a new seed
    assert not torch.equal(y2 != 0, y != 0)


@pytest.mark.parametrize("M,H,res", [(57, 768, True), (1824, 768, True), (300, 768, False), (33, 64, True)])
def test_layernorm_with_fused_dropout_equals_dropout_then_layernorm(M, H, res):
    """gridmm_layernorm_dropout / _bwd (LN(dropout(x) + r) in one launch each way) against gridmm_dropout followed by
    gridmm_layernorm and their backward kernels: the same seed gives the same mask, outputs and all four gradients are
    bit-identical."""
    from gridmm_amd import autograd as ag
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(M + H)
    x0, r0 = torch.randn(2, M, H, generator=g), torch.randn(2, M, H, generator=g)
    ln = torch.nn.LayerNorm(H, eps=1e-12).to(dev)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(H, generator=g))
        ln.bias.copy_(torch.randn(H, generator=g))
    dy = torch.randn(2, M, H, generator=g).to(dev)
    outs = []
    for fused in (False, True):
        torch.manual_seed(123)
        x = x0.to(dev).requires_grad_()
        r = r0.to(dev).requires_grad_() if res else None
        ln.zero_grad(set_to_none=True)
        if fused:
            y = ag.layer_norm(x, ln, residual=r, dropout_p=0.1)
        else:
            y = ag.layer_norm(ag.dropout(x, 0.1), ln, residual=r)
        y.backward(dy)
        outs.append([y.detach().clone(), x.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()] + ([r.grad.clone()] if res else []))
    for a, b in zip(*outs):
        assert torch.equal(a, b), float((a - b).abs().max())
    assert float((outs[0][1] == 0).float().mean()) > 0.05          # a real mask: ~10 % of x's gradient is dropped
