"""GPU: the fused row-wise stages and the grouped / row-mapped GEMM (round 3) vs plain PyTorch fp32/fp64 references of
the same ops, through the C-ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda")


def _ops():
    from gridmm_amd import ops
    return ops


def test_linear_reads_a_sub_sequence_in_place(dev):
    """A = rows [S, S+L) of every episode of a (B, S+L, H) plane buffer (batched row map) == the gathered copy."""
    ops = _ops()
    torch.manual_seed(0)
    B, S, L, H, N = 5, 23, 17, 768, 512
    full = torch.randn(B, S + L, H, device=dev)
    buf = ops.Act(None, *ops._planes_like((B, S + L, H), dev))
    sub = ops.split_rows(full[:, S:].contiguous(), out=(buf.hi[:, S:], buf.lo[:, S:]))
    head = ops.split_rows(full[:, :S].contiguous(), out=(buf.hi[:, :S], buf.lo[:, :S]))
    w = torch.randn(N, H, device=dev) * 0.05
    b = torch.randn(N, device=dev) * 0.1
    pw = ops.PackedLinear(w, b)
    for act, lo, hi in ((sub, S, S + L), (head, 0, S)):
        got = ops.linear(ops.Act(None, act.hi, act.lo), pw).f32
        want = ops.linear(full[:, lo:hi].contiguous(), pw).f32
        assert got.shape == want.shape and torch.equal(got, want)
        ref = full[:, lo:hi].double() @ w.double().t() + b.double()
        assert (got.double() - ref).abs().max() < 2e-4
    # the whole buffer as one uniform sequence sees both parts
    allrows = ops.linear(buf, pw).f32
    assert torch.equal(allrows[:, S:], ops.linear(full[:, S:].contiguous(), pw).f32)


def test_layernorm_planes_into_a_longer_sequence(dev):
    ops = _ops()
    torch.manual_seed(1)
    B, S, L, H = 4, 19, 6, 768
    x = torch.randn(B, S, H, device=dev)
    g, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
    buf = ops._planes_like((B, S + L, H), dev)
    buf[0].zero_(); buf[1].zero_()
    y = ops.layernorm(x, g, b, 1e-12, planes_out=(buf[0][:, :S], buf[1][:, :S]))
    plain = ops.layernorm(x, g, b, 1e-12, want_planes=True)
    assert torch.equal(y.f32, plain.f32)
    assert torch.equal(buf[0][:, :S], plain.hi) and torch.equal(buf[1][:, :S], plain.lo)
    assert not buf[0][:, S:].any() and not buf[1][:, S:].any()          # the tail rows are untouched


@pytest.mark.parametrize("B,G,V", [(32, 20, 37), (3, 7, 9)])
def test_grouped_gemm_equals_single_launches(dev, B, G, V):
    ops = _ops()
    torch.manual_seed(2)
    H, Sq, S, L = 768, G + V, 196 + G, 11
    qa = ops.split_rows(torch.randn(B, Sq, H, device=dev))
    ctx = torch.randn(B, S + L, H, device=dev)
    kv = ops.split_rows(ctx)
    w_gl, b_gl = torch.randn(2 * H, H, device=dev) * 0.05, torch.randn(2 * H, device=dev) * 0.1
    w_gr, b_gr = torch.randn(H, H, device=dev) * 0.05, torch.randn(H, device=dev) * 0.1
    w_f, b_f = torch.randn(H, 2 * H, device=dev) * 0.05, torch.randn(H, device=dev) * 0.1
    pw_gl, pw_gr, pw_f = ops.PackedLinear(w_gl, b_gl), ops.PackedLinear(w_gr, b_gr), ops.PackedLinear(w_f, b_f)
    h_gl = torch.empty(B * Sq, 2 * H, device=dev)
    h_gr = torch.empty(B * G, H, device=dev)
    fa, fb = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    ops.linear_grouped([
        ops.gemm_problem(qa.hi, qa.lo, H, B * Sq, pw_gl, h_gl, act=ops.ACT_RELU),
        ops.gemm_problem(kv.hi, kv.lo, H, B * G, pw_gr, h_gr, act=ops.ACT_RELU, a_rpb=G, a_bs=(S + L) * H, a_off=196 * H),
        ops.gemm_problem(qa.hi, qa.lo, Sq * H, B, pw_f, fa, K=H, bias=False),
        ops.gemm_problem(qa.hi, qa.lo, Sq * H, B, pw_f, fb, K=H, bias=False, a_off=G * H, w_col0=H)])
    want_gl = ops.linear(qa, pw_gl, act=ops.ACT_RELU).f32.view(B * Sq, 2 * H)
    assert (h_gl - want_gl).abs().max() < 1e-5 * max(1.0, want_gl.abs().max().item())
    want_gr = ops.linear(ctx[:, 196:S].contiguous(), pw_gr, act=ops.ACT_RELU).f32.view(B * G, H)
    assert (h_gr - want_gr).abs().max() < 1e-5 * max(1.0, want_gr.abs().max().item())
    x = qa.f32
    ref = torch.cat([x[:, 0], x[:, G]], 1).double() @ w_f.double().t()
    assert ((fa + fb).double() - ref).abs().max() < 2e-4


def test_cells_embed_equals_compaction_plus_position_embedding(dev):
    ops = _ops()
    torch.manual_seed(3)
    B, H, G = 6, 768, 9
    S = 196 + G
    proj = torch.randn(B, 196, H, device=dev)
    pos = torch.randn(B, 196, 5, device=dev)
    occ = (torch.rand(B, 196, device=dev) < 0.6).to(torch.uint8)
    occ[0] = 1; occ[1] = 0; occ[1, 77] = 1
    lin, ln = nn.Linear(5, H).to(dev), nn.LayerNorm(H, eps=1e-12).to(dev)
    with torch.no_grad():
        ln.weight.normal_(); ln.bias.normal_()
        pos_emb = F.layer_norm(pos @ lin.weight.t() + lin.bias, (H,), ln.weight, ln.bias, 1e-12)
        want = torch.empty(B, S, H, device=dev).fill_(7.0)
        want_mask = torch.zeros(B, S, dtype=torch.uint8, device=dev)
        n1, c1 = ops.cells_compact(proj, pos_emb.contiguous(), occ, want, want_mask)
        got = torch.empty(B, S, H, device=dev).fill_(7.0)
        masks = torch.zeros(B, S + 13, dtype=torch.uint8, device=dev)
        gm = (torch.rand(B, G, device=dev) < 0.5).to(torch.uint8)
        n2, c2 = ops.cells_embed(proj, pos, lin, ln, occ, got, masks, tail_mask=gm)
    assert torch.equal(n1, n2) and torch.equal(c1, c2)
    assert torch.equal(masks[:, :196], want_mask[:, :196]) and torch.equal(masks[:, 196:S], gm)
    assert not masks[:, S:].any()
    assert (got[:, :196] - want[:, :196]).abs().max() < 2e-5
    assert torch.equal(got[:, 196:], want[:, 196:])                       # rows behind the cells are not written


def test_node_embed_matches_linear_layernorm_adds(dev):
    ops = _ops()
    torch.manual_seed(4)
    B, G, V, L, H = 5, 8, 11, 13, 768
    S = 196 + G
    gpos, vpos = torch.randn(B, G, 7, device=dev), torch.randn(B, V, 14, device=dev)
    gimg, vimg = torch.randn(B, G, H, device=dev), torch.randn(B, V, H, device=dev)
    steps = torch.randint(0, 20, (B, G), device=dev)
    table = torch.randn(20, H, device=dev)
    glin, gln = nn.Linear(7, H).to(dev), nn.LayerNorm(H, eps=1e-12).to(dev)
    vlin, vln = nn.Linear(14, H).to(dev), nn.LayerNorm(H, eps=1e-12).to(dev)
    map_embeds = torch.zeros(B, S, H, device=dev)
    q = torch.zeros(B, G + V, H, device=dev)
    qp = ops._planes_like((B, G + V, H), dev)
    qp[0].zero_(); qp[1].zero_()
    gm, vm, tm = [(torch.rand(B, n, device=dev) < 0.6).to(torch.uint8) for n in (G, V, L)]
    kvm = torch.full((B, S + L), 9, dtype=torch.uint8, device=dev)
    qm = torch.empty(B, G + V, dtype=torch.uint8, device=dev)
    with torch.no_grad():
        ops.node_embed([ops.embed_seg(gpos, glin, gln, gimg, map_embeds[:, 196:], table=table, idx=steps),
                        ops.embed_seg(vpos, vlin, vln, vimg, q[:, G:], planes=(qp[0][:, G:], qp[1][:, G:]))],
                       H, gm, vm, tm, kvm, 196, qm)
        wg = F.layer_norm(gpos @ glin.weight.t() + glin.bias, (H,), gln.weight, gln.bias, 1e-12) + gimg + table[steps]
        wv = F.layer_norm(vpos @ vlin.weight.t() + vlin.bias, (H,), vln.weight, vln.bias, 1e-12) + vimg
    assert (map_embeds[:, 196:] - wg).abs().max() < 3e-5 and not map_embeds[:, :196].any()
    assert (q[:, G:] - wv).abs().max() < 3e-5 and not q[:, :G].any()
    rec = qp[0][:, G:].float() + qp[1][:, G:].float()
    assert (rec - q[:, G:]).abs().max() <= 2.0 ** -15 * q.abs().max() and not qp[0][:, :G].any()
    assert torch.equal(kvm[:, 196:S], gm) and torch.equal(kvm[:, S:], tm) and (kvm[:, :196] == 9).all()
    assert torch.equal(qm, torch.cat([gm, vm], 1))


@pytest.mark.parametrize("with_obj,with_fuse", [(False, True), (True, True), (False, False)])
def test_nav_heads_equals_ln_dot_plus_fuse_logits(dev, with_obj, with_fuse):
    ops = _ops()
    torch.manual_seed(5)
    B, G, V, H = 7, 9, 12, 768
    Sq = G + V
    nh = 3 if with_obj else 2
    h_gl = torch.randn(B * Sq, nh * H, device=dev).clamp_min(0)
    h_gr = torch.randn(B * G, H, device=dev).clamp_min(0)
    fa, fb, fbias = torch.randn(B, H, device=dev), torch.randn(B, H, device=dev), torch.randn(H, device=dev)
    nets = [nn.Sequential(nn.Linear(H, H), nn.ReLU(), nn.LayerNorm(H, eps=1e-12), nn.Linear(H, 1)).to(dev) for _ in range(5)]
    for n in nets:
        with torch.no_grad():
            n[2].weight.normal_(); n[2].bias.normal_()
    gm = (torch.rand(B, G, device=dev) < 0.8).to(torch.uint8)
    gv = (torch.rand(B, G, device=dev) < 0.3).to(torch.uint8)
    vn = (torch.rand(B, V, device=dev) < 0.6).to(torch.uint8)
    vo = (torch.rand(B, V, device=dev) < 0.5).to(torch.uint8) if with_obj else None
    con = torch.randint(-2, V, (B, G), dtype=torch.int32, device=dev)
    cv = (torch.rand(B, V, device=dev) < 0.3).to(torch.uint8)
    tails = [ops.cls_tail(nets[0]) if with_fuse else None, ops.cls_tail(nets[1]), ops.cls_tail(nets[2]),
             ops.cls_tail(nets[3]), ops.cls_tail(nets[4]) if with_obj else None]
    with torch.no_grad():
        got = ops.nav_heads(h_gl, fa if with_fuse else None, fb if with_fuse else None, fbias if with_fuse else None, h_gr,
                            tails, gm, gv, vn, vo, con, cv, G, V)

        def tail(net, x):
            return ops.ln_dot(x.contiguous(), net[2].weight, net[2].bias, net[2].eps, net[3].weight.view(-1), net[3].bias)
        hv = h_gl.view(B, Sq, nh * H)
        g_raw = tail(nets[1], hv[:, :G, :H])
        l_raw = tail(nets[2], hv[:, G:, H:2 * H])
        gr_raw = tail(nets[3], h_gr.view(B, G, H))
        f_raw = tail(nets[0], (fa + fb + fbias).clamp_min(0)) if with_fuse else None
        want = ops.fuse_logits(g_raw, l_raw, gr_raw, f_raw, gm, gv, vn, con, cv)
    for a, w in zip(got[:4], want):
        f = torch.isfinite(w)
        assert torch.equal(f, torch.isfinite(a))
        assert (a[f] - w[f]).abs().max() < 2e-5 if f.any() else True
    if with_obj:
        o_raw = tail(nets[4], hv[:, G:, 2 * H:])
        w = o_raw.masked_fill(vo == 0, -float("inf"))
        f = torch.isfinite(w)
        assert torch.equal(f, torch.isfinite(got[4])) and (got[4][f] - w[f]).abs().max() < 2e-5
    else:
        assert got[4] is None
