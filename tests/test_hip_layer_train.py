"""GPU: gridmm_xattn_layer_train_fwd / gridmm_xattn_layer_bwd (one cross-modal layer of the differentiable
path, forward and the WHOLE backward as one C call each) against the op-by-op autograd path over the same kernels
(gridmm_amd/vilmodel_train.py with FUSED_XLAYER off).  Same kernels, same order, same tiles -> outputs and every gradient
must be bit-identical; the op-by-op path itself is pinned against the reference's gradients in tests/test_hip_pretrain.py.
Reference layer: map_nav_src/models/vilmodel.py:399-427, pretrain_src/model/vilmodel.py:404-415."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer(inter=256, lang2visn=False, seed=0):
    from gridmm_amd.vilmodel import GraphLXRTXLayer, default_config
    cfg = default_config(intermediate_size=inter, use_lang2visn_attn=lang2visn)
    torch.manual_seed(seed)
    layer = GraphLXRTXLayer(cfg).cuda()
    for p in layer.parameters():
        torch.nn.init.normal_(p, std=0.05)
    return layer


def _model(p_hidden, p_attn, training=True):
    cfg = types.SimpleNamespace(hidden_dropout_prob=p_hidden, attention_probs_dropout_prob=p_attn)
    return types.SimpleNamespace(heads=12, config=cfg, training=training)


def _run(fused, layer, model, x, kv_all, kv_col, ctx_mask, self_mask, dy, lang=False, seed=11):
    from gridmm_amd import vilmodel_train as VT
    VT.FUSED_XLAYER = fused
    try:
        for p in layer.parameters():
            p.grad = None
        x = x.clone().requires_grad_()
        kv_all = kv_all.clone().requires_grad_()
        torch.manual_seed(seed)                      # the dropout seeds come from torch's CPU generator, in call order
        if lang:
            y = VT.lang2visn_layer(model, layer, x, self_mask, kv_all, ctx_mask)      # kv_all = the vision tokens here
        else:
            H = x.shape[-1]
            kv = kv_all if kv_col is None else kv_all
            y = VT.x_layer(model, layer, kv, ctx_mask, x, self_mask, kv_col=0 if kv_col is None else kv_col)
        y.backward(dy)
        torch.cuda.synchronize()
        grads = {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None}
        return y.detach().clone(), x.grad.clone(), kv_all.grad.clone(), grads
    finally:
        VT.FUSED_XLAYER = True


@pytest.mark.parametrize("B,Sq,Sk,kv_cols,kv_col,p", [(3, 57, 120, 2, 0, 0.0), (2, 37, 296, 8, 2 * 768 * 2, 0.0), (4, 20, 64, 2, 0, 0.1),
                                                       (2, 216, 80, 4, 2 * 768, 0.1)])
def test_fused_layer_equals_op_by_op_bitwise(B, Sq, Sk, kv_cols, kv_col, p):
    H = 768
    layer = _layer()
    model = _model(p, p)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, Sq, H, device="cuda", generator=g)
    kv_all = torch.randn(B, Sk, kv_cols * H, device="cuda", generator=g)
    ctx_mask = torch.arange(Sk, device="cuda")[None] < torch.tensor([Sk - 7 * b for b in range(B)], device="cuda")[:, None]
    self_mask = torch.arange(Sq, device="cuda")[None] < torch.tensor([Sq - 3 * b for b in range(B)], device="cuda")[:, None]
    dy = torch.randn(B, Sq, H, device="cuda", generator=g)
    a = _run(False, layer, model, x, kv_all, kv_col, ctx_mask, self_mask, dy)
    b = _run(True, layer, model, x, kv_all, kv_col, ctx_mask, self_mask, dy)
    assert torch.isfinite(b[0]).all()
    assert torch.equal(a[0], b[0]), float((a[0] - b[0]).abs().max())
    assert torch.equal(a[1], b[1]), float((a[1] - b[1]).abs().max())
    assert torch.equal(a[2], b[2]), float((a[2] - b[2]).abs().max())
    assert set(a[3]) == set(b[3]) and len(a[3]) == 22
    for n in a[3]:
        assert torch.equal(a[3][n], b[3][n]), (n, float((a[3][n] - b[3][n]).abs().max()))
    if p > 0:                                        # another seed -> other masks (the dropout is really applied)
        c = _run(True, layer, model, x, kv_all, kv_col, ctx_mask, self_mask, dy, seed=12)
        assert not torch.equal(b[0], c[0])


def test_fused_lang2visn_layer_equals_op_by_op_bitwise():
    """The text-side twin of the pre-training model (forward_lang2visn, vilmodel.py:416-427): the text attends to the vision
    tokens through the SAME visual_attention weights, then lang_self_att + lang FFN."""
    H, B, L, S = 768, 3, 40, 77
    layer = _layer(lang2visn=True)
    model = _model(0.0, 0.0)
    g = torch.Generator(device="cuda").manual_seed(5)
    lang = torch.randn(B, L, H, device="cuda", generator=g)
    visn = torch.randn(B, S, H, device="cuda", generator=g)
    lang_mask = torch.arange(L, device="cuda")[None] < torch.tensor([L, L - 5, L - 11], device="cuda")[:, None]
    visn_mask = torch.arange(S, device="cuda")[None] < torch.tensor([S, S - 9, S - 30], device="cuda")[:, None]
    dy = torch.randn(B, L, H, device="cuda", generator=g)
    a = _run(False, layer, model, lang, visn, None, visn_mask, lang_mask, dy, lang=True)
    b = _run(True, layer, model, lang, visn, None, visn_mask, lang_mask, dy, lang=True)
    for i in range(3):
        assert torch.equal(a[i], b[i]), (i, float((a[i] - b[i]).abs().max()))
    assert set(a[3]) == set(b[3])
    for n in a[3]:
        assert torch.equal(a[3][n], b[3][n]), (n, float((a[3][n] - b[3][n]).abs().max()))


def test_fused_layer_matches_fp64_torch_reference():
    """Independent of the op-by-op path: the fused layer against a plain fp64 torch restatement of the layer."""
    import torch.nn.functional as F
    H, B, Sq, Sk = 768, 2, 30, 50
    layer = _layer(inter=128)
    model = _model(0.0, 0.0)
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(B, Sq, H, device="cuda", generator=g) * 0.5
    kv = torch.randn(B, Sk, 2 * H, device="cuda", generator=g) * 0.5
    cm = torch.arange(Sk, device="cuda")[None] < torch.tensor([Sk, Sk - 13], device="cuda")[:, None]
    sm = torch.arange(Sq, device="cuda")[None] < torch.tensor([Sq, Sq - 4], device="cuda")[:, None]
    dy = torch.randn(B, Sq, H, device="cuda", generator=g)
    y, dx, dkv, grads = _run(True, layer, model, x, kv, 0, cm, sm, dy)

    L64 = _layer(inter=128).double()
    L64.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    x64, kv64 = x.double().requires_grad_(), kv.double().requires_grad_()

    def attn(q, k, v, mask):
        sh = lambda t: t.view(t.shape[0], t.shape[1], 12, 64).transpose(1, 2)        # noqa: E731
        s = sh(q) @ sh(k).transpose(-1, -2) / 8.0
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
        return (torch.softmax(s, -1) @ sh(v)).transpose(1, 2).reshape(q.shape)
    xa, sa = L64.visual_attention, L64.visn_self_att
    q = xa.att.query(x64)
    a1 = xa.output.LayerNorm(xa.output.dense(attn(q, kv64[..., :H], kv64[..., H:], cm)) + x64)
    s = sa.self
    a2 = sa.output.LayerNorm(sa.output.dense(attn(s.query(a1), s.key(a1), s.value(a1), sm)) + a1)
    y64 = L64.visn_output.LayerNorm(L64.visn_output.dense(F.gelu(L64.visn_inter.dense(a2))) + a2)
    y64.backward(dy.double())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-12))    # noqa: E731
    assert rel(y, y64.detach()) < 2e-5
    assert rel(dx, x64.grad) < 1e-4 and rel(dkv, kv64.grad) < 1e-4
    scale = max(float(p.grad.abs().max()) for p in L64.parameters() if p.grad is not None)
    for n, p in L64.named_parameters():
        if p.grad is not None:      # (the key bias has an exactly-zero gradient: softmax is shift invariant -- absolute floor)
            err = float((grads[n].double() - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-4 * scale)
            assert err < 2e-4, (n, err)


@pytest.mark.parametrize("B,S,p", [(3, 57, 0.0), (2, 80, 0.1), (4, 37, 0.1), (1, 216, 0.0)])
def test_fused_bert_layer_equals_op_by_op_bitwise(B, S, p):
    """The layer C calls with KV = NULL (a BertLayer: self attention + feed forward; text and panorama encoders,
    map_nav_src/models/vilmodel.py:214-231) against vilmodel_train.bert_layer's op-by-op form: same kernels in the same
    order -> output, input gradient and every parameter gradient bit-identical, with and without dropout, ragged masks."""
    from gridmm_amd import vilmodel_train as VT
    from gridmm_amd.vilmodel import BertLayer, default_config
    H = 768
    torch.manual_seed(5)
    layer = BertLayer(default_config(intermediate_size=256)).cuda()
    for q in layer.parameters():
        torch.nn.init.normal_(q, std=0.05)
    model = _model(p, p)
    g = torch.Generator(device="cuda").manual_seed(B + S)
    x0 = torch.randn(B, S, H, device="cuda", generator=g)
    dy = torch.randn(B, S, H, device="cuda", generator=g)
    lens = torch.randint(1, S + 1, (B,), generator=torch.Generator().manual_seed(S))
    lens[0] = S
    mask = (torch.arange(S)[None] < lens[:, None]).cuda()
    outs = []
    for fused in (True, False):
        VT.FUSED_BERT_LAYER = fused
        try:
            for q in layer.parameters():
                q.grad = None
            x = x0.clone().requires_grad_()
            torch.manual_seed(21)                    # the dropout seeds come from torch's CPU generator, in call order
            y = VT.bert_layer(model, layer, x, mask)
            y.backward(dy)
            torch.cuda.synchronize()
            outs.append((y.detach().clone(), x.grad.clone(), {n: q.grad.clone() for n, q in layer.named_parameters()}))
        finally:
            VT.FUSED_BERT_LAYER = True
    (y1, dx1, g1), (y2, dx2, g2) = outs
    assert torch.isfinite(y1).all()
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2)
    assert set(g1) == set(g2) and len(g1) == 16
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n


@pytest.mark.parametrize("B,S,p,layers", [(3, 37, 0.0, 2), (2, 216, 0.1, 1), (4, 57, 0.1, 2), (1, 20, 0.0, 1)])
def test_fused_pre_ln_layer_equals_op_by_op_bitwise(B, S, p, layers):
    """gridmm_preln_layer_train_fwd / _bwd (the pre-LayerNorm layers of the panorama and grid encoders,
    map_nav_src/models/transformer.py:170-182) against vilmodel_train.pre_ln_encoder's op-by-op form: output, input gradient
    and all parameter gradients bit-identical, with and without dropout (4 masks per layer), ragged key masks."""
    from gridmm_amd import vilmodel_train as VT
    from gridmm_amd.vilmodel import PreLNEncoder, default_config
    H = 768
    torch.manual_seed(7)
    enc = PreLNEncoder(default_config(intermediate_size=256), layers).cuda()
    for q in enc.parameters():
        torch.nn.init.normal_(q, std=0.05)
    model = _model(p, p)
    g = torch.Generator(device="cuda").manual_seed(B * 7 + S)
    x0 = torch.randn(B, S, H, device="cuda", generator=g)
    dy = torch.randn(B, S, H, device="cuda", generator=g)
    lens = torch.randint(1, S + 1, (B,), generator=torch.Generator().manual_seed(S + 1))
    lens[0] = S
    mask = (torch.arange(S)[None] < lens[:, None]).cuda()
    outs = []
    for fused in (True, False):
        VT.FUSED_PRELN_LAYER = fused
        try:
            for q in enc.parameters():
                q.grad = None
            x = x0.clone().requires_grad_()
            torch.manual_seed(33)                    # the dropout seeds come from torch's CPU generator, in call order
            y = VT.pre_ln_encoder(model, enc, x, mask)
            y.backward(dy)
            torch.cuda.synchronize()
            outs.append((y.detach().clone(), x.grad.clone(), {n: q.grad.clone() for n, q in enc.named_parameters()}))
        finally:
            VT.FUSED_PRELN_LAYER = True
    (y1, dx1, g1), (y2, dx2, g2) = outs
    assert torch.isfinite(y1).all()
    assert torch.equal(y1, y2), float((y1 - y2).abs().max())
    assert torch.equal(dx1, dx2), float((dx1 - dx2).abs().max())
    assert set(g1) == set(g2) and len(g1) == 12 * layers + 2
    for n in g1:
        assert torch.equal(g1[n], g2[n]), (n, float((g1[n] - g2[n]).abs().max()))


def test_dropout_add_equals_dropout_then_add():
    """gridmm_dropout_add: r + dropout(x) in one pass == gridmm_dropout followed by an fp32 add, bit for bit; the planes are
    the split of that sum; p = 0 is a plain add."""
    from gridmm_amd import autograd as ag, ops
    lib = ag._lib.load()
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(37, 768, device="cuda", generator=g)
    r = torch.randn(37, 768, device="cuda", generator=g)
    for p in (0.0, 0.1, 0.5):
        y = torch.empty_like(x)
        hi, lo = ops._planes_like(x.shape, x.device)
        ag._lib.check(lib.gridmm_dropout_add(ag._p(x), ag._p(r), ag._p(y), ag._p(hi), ag._p(lo), x.numel(), p, 1234, None,
                                             ag._stream()), "gridmm_dropout_add")
        d = torch.empty_like(x)
        if p > 0:
            ag._lib.check(lib.gridmm_dropout(ag._p(x), ag._p(d), x.numel(), p, 1234, None, ag._stream()), "gridmm_dropout")
        else:
            d.copy_(x)
        want = r + d
        torch.cuda.synchronize()
        assert torch.equal(y, want), p
        ref = ops.split_rows(want)
        assert torch.equal(hi, ref.hi) and torch.equal(lo, ref.lo)
