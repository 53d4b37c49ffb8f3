"""GPU: each HIP kernel (through the C-ABI) vs a plain PyTorch fp32 reference of the same op."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda")


def _ops():
    from gridmm_amd import ops
    return ops


@pytest.mark.parametrize("M,N,K", [(37, 768, 768), (6912, 2304, 768), (1824, 768, 3072), (300, 768, 5),
                                   (64, 768, 14), (129, 130, 7), (9472, 1536, 768), (32, 768, 1536),
                                   (6272, 768, 512)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_bf16x3_matches_fp32(dev, M, N, K, act):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N + K + act)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if act != 1 else None
    pw = ops.PackedLinear(w, b)
    out = ops.linear(x, pw, act=act, residual=r, want_planes=(N % 8 == 0))
    y = out.f32
    ref = x.double() @ w.double().t() + b.double()
    if act == 1:
        ref = ref * 0.5 * (1 + torch.erf(ref / math.sqrt(2)))
    if act == 2:
        ref = ref.clamp_min(0)
    if r is not None:
        ref = ref + r.double()
    err = (y.double() - ref).abs().max().item()
    scale = (x.double().abs() @ w.double().abs().t()).max().item()
    assert err <= 4e-5 * scale + 1e-6, (err, scale)     # ~2^-16 relative to the absolute-value product
    # and clearly better than a single bf16 term would be
    assert err <= 2e-4 * max(1.0, ref.abs().max().item())
    # the bf16 hi/lo planes emitted for the next GEMM reproduce the fp32 result to ~2^-16
    if out.hi is not None:
        rec = out.hi.float() + out.lo.float()
        assert (rec - y).abs().max().item() <= 2.0 ** -15 * max(1e-3, y.abs().max().item())


@pytest.mark.parametrize("cfg", [66, 71, 75, 76])
@pytest.mark.parametrize("M,N,K", [(6912, 768, 3072), (1824, 768, 768), (191, 200, 64), (193, 132, 128), (1000, 764, 1536),
                                   (6912, 768, 768), (50, 4, 192), (6912, 3072, 768)])
def test_round6_tiles_match_fp64_and_the_reference_tile_bit_for_bit(dev, cfg, M, N, K):
    """The tiles pick_cfg gained in round 6 -- 192x128 with one 16-wave workgroup per CU (66) and the register-pipelined
    128x64 loops (71, 75: fragments of sub-step t + 1 read under the MFMAs of t, DMA pieces issued between the MFMAs) --
    and the two-per-CU 192x128 form (76: uneven DMA piece split, 16-row epilogue passes) -- forced through gridmm_linear_planes_cfg: ragged M / N, one k-step, long contractions, with bias + residual + plane output.
    Same products in the same order as every other tile, so the result must EQUAL the 64x64 tile's (cfg 4) bit for bit; run
    repeatedly, because a misplaced wait in a DMA ring shows up as a rare wrong tile, not as a constant error."""
    import ctypes
    from gridmm_amd import _lib
    ops = _ops()
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 5 * K + cfg)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    r = torch.randn(M, N, generator=g).to(dev)
    pw = ops.PackedLinear(w, b)
    a = ops.split_rows(x)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(c):
        y = torch.full((M, N), float("nan"), device=dev)
        hi = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        lo = torch.zeros_like(hi)
        rc = lib.gridmm_linear_planes_cfg(a.hi.data_ptr(), a.lo.data_ptr(), K, pw.hi.data_ptr(), pw.lo.data_ptr(), pw.Kp,
                                          b.data_ptr(), r.data_ptr(), N, y.data_ptr(), N, hi.data_ptr(), lo.data_ptr(), N,
                                          M, N, K, 1, c, st)
        assert rc == 0, rc
        return y, hi, lo
    y0, h0, l0 = run(4)
    ref = x.double() @ w.double().t() + b.double()
    ref = ref * 0.5 * (1 + torch.erf(ref / math.sqrt(2))) + r.double()
    scale = (x.double().abs() @ w.double().abs().t()).max().item()
    assert (y0.double() - ref).abs().max().item() <= 4e-5 * scale + 1e-6
    for _ in range(12):
        y, h, l = run(cfg)
        assert torch.equal(y, y0) and torch.equal(h, h0) and torch.equal(l, l0)


def test_round6_heuristic_picks_the_new_tiles_for_the_step_shapes(dev):
    """pick_cfg is internal; what can be observed is that the heuristic's result equals the forced tile's for the shapes the
    rule is meant for -- and that results do not depend on the tile at all (bit-identical by construction)."""
    import ctypes
    from gridmm_amd import _lib
    ops = _ops()
    lib = _lib.load()
    for (M, N, K) in [(6912, 768, 3072), (6912, 768, 768), (1824, 768, 3072), (1824, 768, 768)]:
        x = torch.randn(M, K, device=dev)
        pw = ops.PackedLinear(torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev))
        a = ops.split_rows(x)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        outs = []
        for c in (0, 15):
            y = torch.empty(M, N, device=dev)
            assert lib.gridmm_linear_planes_cfg(a.hi.data_ptr(), a.lo.data_ptr(), K, pw.hi.data_ptr(), pw.lo.data_ptr(), pw.Kp,
                                                pw.bias.data_ptr(), None, 0, y.data_ptr(), N, None, None, 0, M, N, K, 0, c, st) == 0
            outs.append(y)
        assert torch.equal(outs[0], outs[1])


def test_linear_strided_input_and_output(dev):
    ops = _ops()
    x = torch.randn(4, 50, 1024, device=dev)[..., :768]        # row stride 1024
    w = torch.randn(96, 768, device=dev) * 0.05
    pw = ops.PackedLinear(w, None)
    y = ops.linear(x, pw).f32
    assert torch.allclose(y, x @ w.t(), atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("M,H", [(5, 768), (6912, 768), (33, 64), (7, 1024)])
def test_layernorm_variants(dev, M, H):
    ops = _ops()
    x, r, a = torch.randn(M, H, device=dev) * 3 + 1, torch.randn(M, H, device=dev), torch.randn(M, H, device=dev)
    g, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
    table = torch.randn(11, H, device=dev)
    idx = torch.randint(0, 11, (M,), device=dev)
    for eps in (1e-12, 1e-5):
        y = ops.layernorm(x, g, b, eps).f32
        assert torch.allclose(y, F.layer_norm(x, (H,), g, b, eps), atol=2e-5, rtol=1e-5)
        ya = ops.layernorm(x, g, b, eps, residual=r, add1=a, table=table, idx=idx, want_planes=(H % 8 == 0))
        y = ya.f32
        ref = F.layer_norm(x + r, (H,), g, b, eps) + a + table[idx]
        assert torch.allclose(y, ref, atol=3e-5, rtol=1e-5)
        if ya.hi is not None:
            assert ((ya.hi.float() + ya.lo.float()) - y).abs().max() <= 2.0 ** -15 * y.abs().max()
    w, b0 = torch.randn(H, device=dev), torch.randn(1, device=dev)
    d = ops.ln_dot(x, g, b, 1e-12, w, b0)
    ref = F.layer_norm(x, (H,), g, b, 1e-12) @ w + b0
    assert torch.allclose(d, ref, atol=2e-4, rtol=1e-5)
    z = ops.layernorm(torch.zeros(3, H, device=dev), g, b, 1e-5).f32      # zero rows -> beta, no NaN
    assert torch.allclose(z, b.expand(3, H))


@pytest.mark.parametrize("B,Sq,Sk", [(2, 16, 16), (3, 57, 296), (2, 216, 216), (1, 5, 37), (2, 216, 80)])
def test_attention_matches_fp32_softmax(dev, B, Sq, Sk):
    ops = _ops()
    Hh = 12
    g = torch.Generator().manual_seed(B + Sq + Sk)
    qkv = torch.randn(B, max(Sq, Sk), 3 * 768, generator=g).to(dev)
    q, k, v = qkv[:, :Sq, :768], qkv[:, :Sk, 768:1536], qkv[:, :Sk, 1536:]          # strided views
    lens = torch.randint(1, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    mask = (torch.arange(Sk)[None] < lens[:, None])
    mask[-1, 0] = False                                                            # hole at the front too
    if not mask[-1].any():
        mask[-1, -1] = True
    mask = mask.to(dev)
    oa = ops.attention(q, k, v, mask, want_f32=True, want_planes=True)
    o = oa.f32
    assert ((oa.hi.float() + oa.lo.float()) - o).abs().max() <= 2.0 ** -15 * o.abs().max()
    def heads(t):
        return t.reshape(B, -1, Hh, 64).permute(0, 2, 1, 3).double()
    s = heads(q) @ heads(k).transpose(-1, -2) / 8.0
    s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ heads(v)).permute(0, 2, 1, 3).reshape(B, Sq, 768)
    assert (o.double() - ref).abs().max().item() < 2e-5
    # additive -10000 convention gives the same numbers in fp32 (vilmodel.py:136)
    s2 = (heads(q) @ heads(k).transpose(-1, -2) / 8.0).float() + (1.0 - mask[:, None, None, :].float()) * -10000.0
    ref2 = (torch.softmax(s2, -1) @ heads(v).float()).permute(0, 2, 1, 3).reshape(B, Sq, 768)
    assert (o - ref2).abs().max().item() < 2e-5


def test_copy_rows_and_cells_compact_quirk(dev):
    ops = _ops()
    B, H, G = 3, 768, 5
    proj, pos = torch.randn(B, 196, H, device=dev), torch.randn(B, 196, H, device=dev)
    occ = torch.zeros(B, 196, dtype=torch.uint8, device=dev)
    occ[0, torch.randperm(196)[:150]] = 1
    occ[1, [3, 17, 18, 95, 96, 150, 195]] = 1
    occ[2, :] = 1
    out = torch.full((B, 196 + G, H), 7.0, device=dev)
    mask = torch.full((B, 196 + G), 9, dtype=torch.uint8, device=dev)
    n, cmax = ops.cells_compact(proj, pos, occ, out, mask)
    # python restatement of vilmodel.py:813-823
    m = occ.clone().long().cpu()
    emb = torch.zeros(B, int(m.sum(1).max()), H)
    cells = (proj + pos).cpu()
    for b in range(B):
        mm = m[b]
        emb[b, :mm.sum()] = cells[b][mm == 1]
        m[b, :mm.sum()] = 1
        m[b, mm.sum():] = 0
    C = emb.shape[1]
    assert int(cmax) == C and n.tolist() == occ.sum(1).tolist()
    assert torch.equal(mask[:, :C].cpu().long(), m[:, :C])
    assert (mask[:, C:196] == 0).all() and (mask[:, 196:] == 9).all()
    assert torch.equal(out[:, :C].cpu(), emb)
    assert (out[:, 196:] == 7.0).all()
    src = torch.randn(B, G, H, device=dev)
    ops.copy_rows(src, out, 196)
    assert torch.equal(out[:, 196:], src)


@pytest.mark.parametrize("B,Sq,Sk", [(2, 16, 32), (3, 57, 296), (2, 216, 216), (1, 5, 37), (2, 216, 80)])
def test_attention_bf16x3_planes_matches_fp32_softmax(dev, B, Sq, Sk):
    """The hot-path attention (MFMA bf16 3-term split, V transposed per head) vs an fp64 softmax reference."""
    ops = _ops()
    Hh = 12
    g = torch.Generator().manual_seed(B * 3 + Sq + Sk)
    qkv = torch.randn(B, max(Sq, Sk), 3 * 768, generator=g).to(dev)
    a = ops.split_rows(qkv)
    lens = torch.randint(1, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    mask = (torch.arange(Sk)[None] < lens[:, None])
    mask[-1, 0] = False
    if not mask[-1].any():
        mask[-1, -1] = True
    mask = mask.to(dev)
    sl = lambda c0, n: (a.hi[:, :n, c0:c0 + 768], a.lo[:, :n, c0:c0 + 768])
    out = ops.attention_planes(sl(0, Sq), sl(768, Sk), sl(1536, Sk), mask, want_f32=True, want_planes=True)
    o = out.f32
    def heads(t):
        return t.reshape(B, -1, Hh, 64).permute(0, 2, 1, 3).double()
    q, k, v = qkv[:, :Sq, :768], qkv[:, :Sk, 768:1536], qkv[:, :Sk, 1536:]
    s = heads(q) @ heads(k).transpose(-1, -2) / 8.0
    s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ heads(v)).permute(0, 2, 1, 3).reshape(B, Sq, 768)
    err = (o.double() - ref).abs().max().item()
    assert err < 2e-4, err          # ~2^-16 relative on scores of magnitude ~10 (N(0,1) q.k over 64 dims)
    assert ((out.hi.float() + out.lo.float()) - o).abs().max() <= 2.0 ** -15 * o.abs().max()


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 5, 6, 9, 24])
@pytest.mark.parametrize("B,Sq,Sk", [(2, 16, 32), (3, 57, 296), (2, 216, 216), (1, 5, 37), (2, 216, 80), (2, 57, 57), (1, 40, 512),
                                     (2, 90, 700), (1, 300, 1100)])   # RxR: 250-300 instruction tokens + map + a long graph
def test_attention_rows_matches_fp32_softmax(dev, B, Sq, Sk, cfg):
    """gridmm_attention_rows (K / V staged row-major in LDS, transpose reads) vs an fp64 softmax reference: ragged
    masks, a masked first key, whole 32-key tiles masked out, key counts across the 64 / 128-row chunk boundaries, and
    separate K / V buffers with different row strides."""
    ops = _ops()
    Hh = 12
    g = torch.Generator().manual_seed(B * 3 + Sq + Sk)
    qb = torch.randn(B, Sq, 768, generator=g).to(dev)
    kvb = torch.randn(B, Sk, 4 * 768, generator=g).to(dev)      # [pad | K | pad | V]: strided column slices
    qa, kva = ops.split_rows(qb), ops.split_rows(kvb)
    lens = torch.randint(1, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    mask = (torch.arange(Sk)[None] < lens[:, None])
    mask[-1, 0] = False
    if Sk > 100:
        mask[0, 32:96] = False          # two fully masked 32-key tiles in the middle
    if not mask[-1].any():
        mask[-1, -1] = True
    mask = mask.to(dev)
    ksl = (kva.hi[..., 768:1536], kva.lo[..., 768:1536])
    vsl = (kva.hi[..., 2304:], kva.lo[..., 2304:])
    out = ops.attention_rows((qa.hi, qa.lo), ksl, vsl, mask, want_f32=True, want_planes=True, cfg=cfg)
    o = out.f32
    def heads(t):
        return t.reshape(B, -1, Hh, 64).permute(0, 2, 1, 3).double()
    s = heads(qb) @ heads(kvb[..., 768:1536]).transpose(-1, -2) / 8.0
    s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ heads(kvb[..., 2304:])).permute(0, 2, 1, 3).reshape(B, Sq, 768)
    err = (o.double() - ref).abs().max().item()
    assert err < 2e-4, err
    assert ((out.hi.float() + out.lo.float()) - o).abs().max() <= 2.0 ** -15 * o.abs().max()


def test_attention_rows_fully_masked_row_is_zero(dev):
    ops = _ops()
    x = ops.split_rows(torch.randn(2, 20, 3 * 768, device=dev))
    mask = torch.ones(2, 20, dtype=torch.bool, device=dev)
    mask[1] = False
    sl = lambda c0: (x.hi[..., c0:c0 + 768], x.lo[..., c0:c0 + 768])
    out = ops.attention_rows(sl(0), sl(768), sl(1536), mask, want_f32=True)
    assert torch.isfinite(out.f32).all() and (out.f32[1] == 0).all() and out.f32[0].abs().max() > 0


def test_xattn_layer_entry_point_equals_the_eleven_calls(dev):
    """gridmm_xattn_layer_fwd (one C call per GraphLXRTXLayer) against the same layer issued kernel by kernel from
    Python (ops.TIMER forces that path): bit-identical outputs, for a cross-attention over a separate context with a
    ragged mask and a shared K/V buffer read at a column offset."""
    import numpy as np
    ops = _ops()
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    torch.manual_seed(0)
    model = GlocalTextPathNavCMT(default_config(num_l_layers=1, num_pano_layers=1, num_x_layers=2, intermediate_size=256,
                                                vocab_size=100)).eval().to(dev)
    layer = model.local_encoder.encoder.x_layers[1]
    with torch.no_grad():                    # (non-trivial gamma / beta)
        for ln in (layer.visual_attention.output.LayerNorm, layer.visn_self_att.output.LayerNorm, layer.visn_output.LayerNorm):
            ln.weight.copy_(1.0 + 0.2 * torch.randn(768, device=dev))
            ln.bias.copy_(0.2 * torch.randn(768, device=dev))
    B, Sq, Sk, H = 3, 57, 100, 768
    g = torch.Generator().manual_seed(1)
    x = ops.split_rows(torch.randn(B, Sq, H, generator=g).to(dev))
    kv = ops.split_rows(torch.randn(B, Sk, 4 * H, generator=g).to(dev))          # K/V of layer 1 at column 2H
    cm = (torch.arange(Sk)[None] < torch.tensor([100, 37, 64])[:, None]).to(dev)
    sm = (torch.arange(Sq)[None] < torch.tensor([57, 57, 40])[:, None]).to(dev)
    with torch.no_grad():
        fused = model._x_layer(layer, "t.1", None, model._u8(cm), x, model._u8(sm), kv=(kv, 2 * H))    # one C call
        ops.TIMER = ops.KernelTimer()
        try:
            split = model._x_layer(layer, "t.1", None, model._u8(cm), x, model._u8(sm), kv=(kv, 2 * H))   # eleven Python calls
        finally:
            ops.TIMER = None
    torch.cuda.synchronize()
    assert torch.equal(fused.f32, split.f32) and torch.equal(fused.hi, split.hi) and torch.equal(fused.lo, split.lo)
    assert torch.isfinite(fused.f32).all() and float(fused.f32.abs().max()) > 0.1
