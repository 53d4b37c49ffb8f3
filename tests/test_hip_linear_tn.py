"""GPU: gridmm_linear_planes_tn -- the weight gradient dW = dY^T X straight from ROW-major planes (hardware transpose reads,
no transposed copies; backward of nn.Linear, map_nav_src/r2r/agent_base.py:199 / pretrain_src/train_r2r.py:262) -- against
fp64, and against the transposed-planes path of rounds 1-3 (same products, same k order inside a 32-row step)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda")


@pytest.mark.parametrize("M,N,K", [(1824, 768, 768), (6912, 768, 3072), (100, 768, 768), (2560, 1000, 768), (57, 64, 64),
                                   (300, 2304, 768), (9472, 1536, 768), (33, 8, 40), (4096, 3072, 768)])
def test_tn_gemm_matches_fp64(dev, M, N, K):
    from gridmm_amd import autograd as ag
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    x = torch.randn(M, K, generator=g).to(dev)
    dy = (torch.randn(M, N, generator=g) * 0.1).to(dev)
    xh, xl, _, Mp, _ = ag.split_rows_pad(x)
    yh, yl, db, Mp2, rows = ag.split_rows_pad(dy, want_colsum=True)
    assert Mp == Mp2 == (M + 31) // 32 * 32
    assert float(xh[M:].float().abs().max() if Mp > M else 0.0) == 0.0          # the pad rows are zero
    dw = ag._gemm_tn_rows((yh, yl), (xh, xl), N, K, M)
    ref = dy.double().t() @ x.double()
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((dw.double() - ref).abs().max()) <= 2e-5 * max(1.0, scale)
    assert float((db.double() - dy.double().sum(0)).abs().max()) <= 1e-4
    assert float((rows.hi.float() + rows.lo.float() - dy).abs().max()) < 1e-4


@pytest.mark.parametrize("M", [1, 20, 33, 57, 1824 - 5, 300])
def test_tn_gemm_on_unpadded_planes_of_any_row_count(dev, M):
    """Planes with EXACTLY M rows (what LayerNorm / GELU / attention emit): the last 32-row step clamps its reads to row M - 1
    and zeroes the copies on one side; the buffer behind the planes is poisoned to prove nothing past row M - 1 is read."""
    from gridmm_amd import autograd as ag, ops
    N, K = 768, 192
    g = torch.Generator().manual_seed(M)
    pool = torch.full((4, M + 40, 768), float("nan"), dtype=torch.bfloat16, device=dev)     # NaN everywhere around the planes
    x, dy = torch.randn(M, K, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    xs, ys = ops.split_rows(x), ops.split_rows(dy)
    xh, xl = pool[0, :M, :K], pool[1, :M, :K]
    yh, yl = pool[2, :M], pool[3, :M]
    xh.copy_(xs.hi); xl.copy_(xs.lo); yh.copy_(ys.hi); yl.copy_(ys.lo)
    lib = ag._lib.load()
    dw = torch.empty(N, K, device=dev)
    ag._lib.check(lib.gridmm_linear_planes_tn(ag._p(yh), ag._p(yl), 768, ag._p(xh), ag._p(xl), 768, ag._p(dw), None, M, N, K, 1,
                                              ag._stream()), "gridmm_linear_planes_tn")
    ref = dy.double().t() @ x.double()
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all()
    assert float((dw.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_tn_gemm_equals_the_transposed_planes_path(dev):
    """Both paths multiply the same bf16 hi / lo values with fp32 accumulation over the same 32-row k-steps: the results
    agree to accumulation-order noise (the order of the rows INSIDE a k-step differs)."""
    from gridmm_amd import autograd as ag
    M, N, K = 1824, 768, 768
    g = torch.Generator().manual_seed(5)
    x, dy = torch.randn(M, K, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    xh, xl, _, Mp, _ = ag.split_rows_pad(x)
    yh, yl, _, _, _ = ag.split_rows_pad(dy)
    a = ag._gemm_tn_rows((yh, yl), (xh, xl), N, K, M)
    th, tl, _, Mp2, _ = ag.transpose_split(x)
    uh, ul, _, _, _ = ag.transpose_split(dy)
    b = ag._gemm_tn((uh, ul), (th, tl), N, K, M, Mp2, dy)
    torch.cuda.synchronize()
    assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) * 1e-2
    # run-to-run: fixed summation order
    a2 = ag._gemm_tn_rows((yh, yl), (xh, xl), N, K, M)
    assert torch.equal(a, a2)


def test_linear_backward_through_the_tn_path_matches_torch(dev):
    from gridmm_amd import autograd as ag
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 57, 768, generator=g).to(dev).requires_grad_()
    w = (torch.randn(1536, 768, generator=g) * 0.05).to(dev).requires_grad_()
    b = torch.randn(1536, generator=g).to(dev).requires_grad_()
    assert ag.TN_GEMM
    y = ag.linear(x, w, b)
    (y * torch.cos(y)).sum().backward()
    got = (x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    yd = torch.nn.functional.linear(xd, wd, bd)
    (yd * torch.cos(yd)).sum().backward()
    for a, r in zip(got, (xd.grad, wd.grad, bd.grad)):
        assert float((a.double() - r).abs().max()) <= 3e-5 * max(1.0, float(r.abs().max()))


@pytest.mark.parametrize("M,N,K", [(1824, 768, 768), (6912, 3072, 768), (57, 64, 64), (300, 2304, 768), (33, 8, 40),
                                   (1000, 1000, 768), (100, 768, 3072)])
def test_bias_gradient_from_the_weight_gradient_gemm(dev, M, N, K):
    """gridmm_linear_planes_tn_db: db = the column sums of dY computed by the GEMM itself from the planes (all-ones operand, one
    partial per contraction range, summed in order) vs fp64; dW is untouched by the extra MFMAs (bit-identical to the plain
    call); two runs bit-identical."""
    from gridmm_amd import autograd as ag
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    dy = (torch.randn(M, N, generator=g) * 0.1 + 0.05).to(dev)
    xh, xl, _, _, _ = ag.split_rows_pad(x)
    yh, yl, _, _, _ = ag.split_rows_pad(dy)
    dw_ref = ag._gemm_tn_rows((yh, yl), (xh, xl), N, K, M)
    dw, db = ag._gemm_tn_rows((yh, yl), (xh, xl), N, K, M, want_db=True)
    dw2, db2 = ag._gemm_tn_rows((yh, yl), (xh, xl), N, K, M, want_db=True)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_ref) and torch.equal(dw, dw2) and torch.equal(db, db2)
    ref = dy.double().sum(0)
    assert float((db.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_grouped_weight_gradients_equal_the_single_launches_bitwise(dev):
    """gridmm_linear_planes_tn_grouped: the weight (+ bias) gradients of a layer's Linears in one grouped GEMM launch per tile
    class + one summing launch == gridmm_linear_planes_tn_db problem by problem, bit for bit (both tile classes, split and
    unsplit contractions, with and without a bias gradient)."""
    import ctypes
    from gridmm_amd import autograd as ag

    class Prob(ctypes.Structure):
        _fields_ = [("A_hi", ctypes.c_void_p), ("A_lo", ctypes.c_void_p), ("lda", ctypes.c_int),
                    ("B_hi", ctypes.c_void_p), ("B_lo", ctypes.c_void_p), ("ldb", ctypes.c_int),
                    ("C", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
                    ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("splits", ctypes.c_int),
                    ("db_ws", ctypes.c_void_p), ("db", ctypes.c_void_p)]
    lib = ag._lib.load()
    M = 1824
    shapes = [(768, 768, True), (2304, 768, True), (768, 3072, False), (3072, 768, True), (64, 64, True), (8, 40, False)]
    g = torch.Generator().manual_seed(9)
    keep, probs, want = [], (Prob * len(shapes))(), []
    for i, (N, K, bias) in enumerate(shapes):
        x = torch.randn(M, K, generator=g).to(dev)
        dy = (torch.randn(M, N, generator=g) * 0.1).to(dev)
        xh, xl, _, Mp, _ = ag.split_rows_pad(x)
        yh, yl, _, _, _ = ag.split_rows_pad(dy)
        if bias:
            dw_ref, db_ref = ag._gemm_tn_rows((yh, yl), (xh, xl), N, K, M, want_db=True)
        else:
            dw_ref, db_ref = ag._gemm_tn_rows((yh, yl), (xh, xl), N, K, M), None
        splits = lib.gridmm_linear_planes_tn_splits(M, N, K)
        dw = torch.full((N, K), float("nan"), device=dev)
        db = torch.full((N,), float("nan"), device=dev) if bias else None
        ws = torch.empty(splits, N, K, device=dev) if splits > 1 else None
        dbw = torch.empty(splits, N, device=dev) if bias else None
        keep += [xh, xl, yh, yl, dbw, dw, db, ws]
        probs[i] = Prob(yh.data_ptr(), yl.data_ptr(), N, xh.data_ptr(), xl.data_ptr(), K, dw.data_ptr(),
                        ws.data_ptr() if ws is not None else None, M, N, K, splits,
                        dbw.data_ptr() if bias else None, db.data_ptr() if bias else None)
        want.append((dw_ref, db_ref, dw, db))
    ag._lib.check(lib.gridmm_linear_planes_tn_grouped(ctypes.byref(probs), len(shapes), ag._stream()),
                  "gridmm_linear_planes_tn_grouped")
    torch.cuda.synchronize()
    for dw_ref, db_ref, dw, db in want:
        assert torch.equal(dw, dw_ref)
        if db_ref is not None:
            assert torch.equal(db, db_ref)
