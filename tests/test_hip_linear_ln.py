"""GPU: gridmm_linear_planes_ln (GEMM + residual + LayerNorm in one launch, rendezvous of a row block's column tiles)
against gridmm_linear_planes followed by gridmm_layernorm (map_nav_src/models/vilmodel.py:156-168, 196-209)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda")


@pytest.fixture(autouse=True)
def _fused_on():
    """The fused form is opt-in (GRIDMM_LN_FUSE / ops.LN_FUSE: measured slower than the two launches in the step)."""
    from gridmm_amd import ops
    keep, ops.LN_FUSE = ops.LN_FUSE, True
    yield
    ops.LN_FUSE = keep


def _case(dev, M, N, K, seed, mean_shift=0.0):
    from gridmm_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = ops.split_rows(torch.randn(M, K, generator=g).to(dev))
    pw = ops.PackedLinear((torch.randn(N, K, generator=g) * 0.05).to(dev), (torch.randn(N, generator=g) * 0.1).to(dev))
    r = (torch.randn(M, N, generator=g) + mean_shift).to(dev)
    gamma = (1.0 + 0.1 * torch.randn(N, generator=g)).to(dev)
    beta = (0.1 * torch.randn(N, generator=g)).to(dev)
    return ops, x, pw, r, gamma, beta


@pytest.mark.parametrize("M,N,K", [(1824, 768, 768), (1824, 768, 3072), (6912, 768, 768), (100, 768, 768), (4224, 768, 3072),
                                   (57, 768, 768), (1824, 512, 768)])
def test_fused_equals_gemm_then_layernorm(dev, M, N, K):
    ops, x, pw, r, gamma, beta = _case(dev, M, N, K, M + N + K)
    h = ops.linear(x, pw, residual=r).f32
    want = ops.layernorm(h, gamma, beta, 1e-12, want_planes=True)
    got = ops.linear_ln(x, pw, gamma, beta, 1e-12, residual=r, want_pre=True)
    assert got is not None, "the fused form must take this shape"
    y, pre = got
    torch.cuda.synchronize()
    assert torch.equal(pre, h)                                   # same GEMM arithmetic, element by element
    assert float((y.f32 - want.f32).abs().max()) < 5e-6
    assert float((y.hi.float() + y.lo.float() - y.f32).abs().max()) < 1e-4
    sync = ops.ln_sync(dev)
    assert int(sync.abs().sum()) == 0                            # the counters are back at zero
    # run-to-run: the merge order of the tile statistics is fixed
    y2, _ = ops.linear_ln(x, pw, gamma, beta, 1e-12, residual=r)
    torch.cuda.synchronize()
    assert torch.equal(y2.f32, y.f32) and torch.equal(y2.hi, y.hi) and torch.equal(y2.lo, y.lo)


def test_fused_rows_with_a_large_common_offset(dev):
    """Rows whose mean is 50x their spread: the tile statistics are (mean, squared deviations from the TILE mean) merged
    by Chan's formula -- no E[x^2] - mean^2 cancellation."""
    ops, x, pw, r, gamma, beta = _case(dev, 1824, 768, 768, 5, mean_shift=50.0)
    h = ops.linear(x, pw, residual=r).f32
    ref = torch.nn.functional.layer_norm(h.double(), (768,), gamma.double(), beta.double(), 1e-12)
    y, _ = ops.linear_ln(x, pw, gamma, beta, 1e-12, residual=r)
    assert float((y.f32.double() - ref).abs().max()) < 2e-5


def test_fused_planes_through_the_row_map_and_without_fp32(dev):
    ops, x, pw, r, gamma, beta = _case(dev, 32 * 57, 768, 768, 9)
    B, S, Sp = 32, 57, 80
    hi, lo = ops._planes_like((B, Sp, 768), dev)
    hi.zero_(); lo.zero_()
    y, _ = ops.linear_ln(x, pw, gamma, beta, 1e-12, residual=r, want_f32=False, planes_out=(hi[:, :S], lo[:, :S]))
    want = ops.layernorm(ops.linear(x, pw, residual=r).f32, gamma, beta, 1e-12, want_planes=True)
    torch.cuda.synchronize()
    assert y.f32 is None
    got = (hi[:, :S].float() + lo[:, :S].float()).reshape(B * S, 768)
    assert float((got - want.f32).abs().max()) < 1e-4
    assert float(hi[:, S:].abs().max()) == 0.0                   # rows outside the map are untouched


def test_fused_launches_in_a_hipgraph_and_back_to_back(dev):
    """200 fused launches back to back in one captured graph, replayed three times: the counters of a row block are
    reused by the next launch as soon as the previous one has left them at zero."""
    ops, x, pw, r, gamma, beta = _case(dev, 1824, 768, 768, 11)
    want, _ = ops.linear_ln(x, pw, gamma, beta, 1e-12, residual=r)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.linear_ln(x, pw, gamma, beta, 1e-12, residual=r)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(200):
            y, _ = ops.linear_ln(x, pw, gamma, beta, 1e-12, residual=r)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y.f32, want.f32) and int(ops.ln_sync(dev).abs().sum()) == 0


def test_shapes_the_fused_form_refuses(dev):
    ops, x, pw, r, gamma, beta = _case(dev, 20000, 768, 768, 13)          # 128x128 tiles: 157 x 6 workgroups > the device holds
    assert ops.linear_ln(x, pw, gamma, beta, 1e-12, residual=r) is None
    ops, x, pw, r, gamma, beta = _case(dev, 256, 100, 768, 14)            # N not a multiple of the tile width
    assert ops.linear_ln(x, pw, gamma, beta, 1e-12, residual=r) is None
