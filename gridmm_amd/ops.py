"""Thin torch-tensor wrappers over the C-ABI (include/gridmm.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; all
arithmetic on the hot path happens inside libgridmm_hip.so.  Every wrapper
raises if a tensor is not on the GPU - there is no fallback.
"""
import ctypes
import math

import os
import torch

from . import _lib

ACT_NONE, ACT_GELU, ACT_RELU, ACT_QUICKGELU = 0, 1, 2, 3
N_CELLS = 196


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the capture stream inside a hipGraph capture).  The raw
    accessor costs ~0.5 us; torch.cuda.current_stream() builds a Stream object per call (~10 us: 8 ms of a fine-tune iteration's
    ~800 kernel calls went there)."""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_CUR_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-launch HIP-event timing (bench.py roofline leg).  Events are recorded on the
    stream the kernels are launched on (torch's current stream == the stream passed to the C-ABI)."""

    def __init__(self):
        self.records = []  # (name, work, start_event, end_event)

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def end(self, name, work, start):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        self.records.append((name, work, start, e))

    def summary(self):
        """name -> dict(calls, ms, work) after a device synchronize."""
        out = {}
        for name, work, s, e in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "work": 0.0})
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
            d["work"] += work
        return out


TIMER = None  # set to a KernelTimer() to time launches
LAST_AGGREGATE_RC = None


def _timed(name, work, fn):
    if TIMER is None:
        return fn()
    s = TIMER.begin()
    r = fn()
    TIMER.end(name, work, s)
    return r


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise _lib.GridmmLibraryError("gridmm ops need GPU tensors (got %s); no CPU fallback exists" % t.device)
    return ctypes.c_void_p(t.data_ptr())


def _rows2d(t):
    """View (..., H) with contiguous last dim as (M, H) + row stride; requires a uniform row stride."""
    if t.dim() > 1 and t.is_contiguous():          # the common case, without the stride walk below
        H = t.shape[-1]
        return (t.numel() // H if H else 0), H, H
    if t.stride(-1) != 1:
        raise ValueError("last dim must be contiguous")
    if t.dim() == 1:
        return 1, t.shape[0], t.shape[0]
    H = t.shape[-1]
    rs = t.stride(-2)
    M = 1
    for d in range(t.dim() - 1):
        M *= t.shape[d]
    # leading dims must fold onto a single stride
    exp = rs
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise ValueError("tensor rows are not uniformly strided: %s %s" % (tuple(t.shape), t.stride()))
        exp *= t.shape[d]
    return M, H, rs


def _rows_map(t):
    """(M, H, row stride, rows_per_batch, batch stride) of a row-contiguous tensor: uniform rows -> rpb = 0; a (B, R, H)
    view whose episodes are not R rows apart (a sub-sequence of a longer padded sequence) -> the batched row map the
    *_map entry points take."""
    try:
        M, H, ld = _rows2d(t)
        return M, H, ld, 0, 0
    except ValueError:
        if t.dim() == 3 and t.stride(2) == 1:
            return t.shape[0] * t.shape[1], t.shape[2], t.stride(1), t.shape[1], t.stride(0)
        raise


class PackedLinear:
    """bf16 hi/lo planes of a Linear weight (N, K) zero-padded to Kp = roundup(K, 32), plus fp32 bias."""

    __slots__ = ("hi", "lo", "bias", "N", "K", "Kp", "_tiled")

    def tiled(self):
        """The planes as [Np / 16][Kp / 32][16][32] blocks (rows zero-padded to Np = roundup(N, 16)): every 1-KiB DMA piece of
        a BK = 32 tile is one contiguous KiB instead of 16 half cache lines (GRIDMM_WT=0 turns the tiled planes off)."""
        t = getattr(self, "_tiled", None)
        if t is None:
            Np = (self.N + 15) // 16 * 16
            out = []
            for p in (self.hi, self.lo):
                q = torch.zeros(Np, self.Kp, dtype=p.dtype, device=p.device)
                q[:self.N] = p
                out.append(q.view(Np // 16, 16, self.Kp // 32, 32).permute(0, 2, 1, 3).contiguous())
            t = self._tiled = tuple(out)
        return t

    def __init__(self, weight, bias=None):
        lib = _lib.load()
        w = weight.detach().to(torch.float32).contiguous()
        self.N, self.K = w.shape
        self.Kp = (self.K + 31) // 32 * 32
        self.hi = torch.empty(self.N, self.Kp, dtype=torch.bfloat16, device=w.device)
        self.lo = torch.empty_like(self.hi)
        _lib.check(lib.gridmm_split_weight(_p(w), _p(self.hi), _p(self.lo), self.N, self.K, self.Kp, _stream()),
                   "gridmm_split_weight")
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous()
        self._tiled = None


def _is_uniform(t):
    try:
        _rows2d(t)
        return True
    except ValueError:
        return False


def uniform_rows(t):
    """Return t if its rows fold onto one stride, else a packed copy (device-side gridmm_copy_rows)."""
    if _is_uniform(t):
        return t
    if t.dim() == 3 and t.stride(2) == 1 and t.dtype == torch.float32 and t.shape[2] % 4 == 0:
        out = torch.empty(t.shape, dtype=t.dtype, device=t.device)
        return copy_rows(t, out, 0)
    return t.contiguous()


class Act:
    """An activation in up to two device representations: fp32 (residual / LayerNorm / attention
    input) and bf16 hi/lo planes (A operand of the next MFMA GEMM, written by the producer kernel)."""

    __slots__ = ("f32", "hi", "lo")

    def __init__(self, f32=None, hi=None, lo=None):
        self.f32, self.hi, self.lo = f32, hi, lo

    @property
    def shape(self):
        return (self.f32 if self.f32 is not None else self.hi).shape

    @property
    def device(self):
        return (self.f32 if self.f32 is not None else self.hi).device


def _planes_like(shape, device):
    """hi / lo planes as the two halves of ONE allocation: a pair moves with a single copy_rows launch."""
    buf = torch.empty((2,) + tuple(shape), dtype=torch.bfloat16, device=device)
    return buf[0], buf[1]


def split_rows(x, out=None):
    """fp32 (..., K) -> Act with bf16 hi/lo planes (K % 8 == 0).  out = (hi, lo): destination plane views, possibly rows of
    a longer padded sequence (batched row map)."""
    lib = _lib.load()
    x = uniform_rows(x)
    M, K, ldx = _rows2d(x)
    assert K % 8 == 0
    if out is None:
        hi, lo = _planes_like(x.shape, x.device)
        ldp, rpb, bs = K, 0, 0
    else:
        hi, lo = out
        assert hi.shape == x.shape and hi.stride() == lo.stride() and hi.dtype == torch.bfloat16
        _, _, ldp, rpb, bs = _rows_map(hi)
    _timed("split_rows", 0.0, lambda: _lib.check(
        lib.gridmm_split_rows_map(_p(x), ldx, _p(hi), _p(lo), ldp, rpb, bs, M, K, _stream()), "gridmm_split_rows"))
    return Act(x, hi, lo)


def linear(x, pw, act=ACT_NONE, residual=None, out=None, want_f32=True, want_planes=False, planes_out=None, allow_tiled=True,
           plane_shift=None):
    """act(x @ W^T + b) (+ residual) -> Act.  x: Act or fp32 tensor (..., K).

    K % 32 == 0 (every hidden-size GEMM): gridmm_linear_planes -- A as bf16 planes (taken from the
    producer, or split here for external fp32 inputs), LDS-DMA pipeline.  Otherwise (K = 5 / 7 / 14
    position features): gridmm_linear with the split done in the kernel.
    """
    lib = _lib.load()
    a = x if isinstance(x, Act) else Act(x)
    shape = a.shape
    K = shape[-1]
    if K != pw.K:
        raise ValueError("linear: bad input %s for weight (%d,%d)" % (tuple(shape), pw.N, pw.K))
    if residual is not None:
        residual = uniform_rows(residual)
    oshape = tuple(shape[:-1]) + (pw.N,)
    dev = a.device
    if (K % 32 == 0) and (pw.N % 4 == 0):
        if a.hi is None:
            a = split_rows(a.f32)
        try:
            M, _, lda, rpb, bs = _rows_map(a.hi)
        except ValueError:
            a = split_rows(a.f32)
            M, _, lda, rpb, bs = _rows_map(a.hi)
        c = out if out is not None else (torch.empty(oshape, dtype=torch.float32, device=dev) if want_f32 else None)
        hi = lo = None
        if planes_out is not None:           # (hi, lo): contiguous destination planes of the result's shape
            hi, lo = planes_out
            assert tuple(hi.shape) == oshape and hi.is_contiguous() and lo.is_contiguous() and hi.dtype == torch.bfloat16
        elif want_planes:
            hi, lo = _planes_like(oshape, dev)
        ldc = _rows2d(c)[2] if c is not None else 0
        ldr = _rows2d(residual)[2] if residual is not None else 0
        def call():
            if plane_shift is not None:      # (tab (E, N) fp32, rows per episode, first shifted column): gridmm_linear_planes_shift
                tab, rpb_s, c0 = plane_shift
                assert rpb == 0 and hi is not None and tab.is_contiguous() and tab.shape[-1] == pw.N
                return _lib.check(lib.gridmm_linear_planes_shift(
                    _p(a.hi), _p(a.lo), lda, _p(pw.hi), _p(pw.lo), pw.Kp, _p(pw.bias), _p(residual), ldr, _p(c), ldc, _p(hi), _p(lo),
                    pw.N, _p(tab), int(rpb_s), int(c0), M, pw.N, K, act, _stream()), "gridmm_linear_planes_shift")
            if WT and allow_tiled and getattr(pw, "_tiled", False) is not False:   # (inference weights, packed once; the training
                                                                                   # path re-packs per step and passes allow_tiled=False)
                th, tl = pw.tiled()
                rc = lib.gridmm_linear_planes_map(_p(a.hi), _p(a.lo), lda, rpb, bs, _p(th), _p(tl), pw.Kp, _lib.W_TILED, _p(pw.bias),
                                                  _p(residual), ldr, _p(c), ldc, _p(hi), _p(lo), pw.N, M, pw.N, K, act, _stream())
                if rc != -2:
                    return _lib.check(rc, "gridmm_linear_planes (tiled W)")
            _lib.check(
                lib.gridmm_linear_planes_map(_p(a.hi), _p(a.lo), lda, rpb, bs, _p(pw.hi), _p(pw.lo), pw.Kp, _lib.W_ROWMAJOR,
                                             _p(pw.bias), _p(residual), ldr, _p(c), ldc, _p(hi), _p(lo), pw.N, M, pw.N, K, act,
                                             _stream()),
                "gridmm_linear_planes")
        _timed("linear", 2.0 * M * pw.N * K, call)
        return Act(c, hi, lo)
    xf = uniform_rows(a.f32)
    M, _, lda = _rows2d(xf)
    if xf.dtype != torch.float32:
        raise ValueError("linear: fp32 input expected")
    c = out if out is not None else torch.empty(oshape, dtype=torch.float32, device=dev)
    ldc = _rows2d(c)[2]
    ldr = _rows2d(residual)[2] if residual is not None else 0
    _timed("linear_small", 2.0 * M * pw.N * K, lambda: _lib.check(
        lib.gridmm_linear(_p(xf), lda, _p(pw.hi), _p(pw.lo), pw.Kp, _p(pw.bias), _p(residual), ldr,
                          _p(c), ldc, M, pw.N, K, act, _stream()), "gridmm_linear"))
    return split_rows(c) if want_planes else Act(c)


def layernorm(x, gamma, beta, eps, residual=None, add1=None, table=None, idx=None, out=None,
              want_f32=True, want_planes=False, planes_out=None):
    """LN(x (+ residual)) * gamma + beta (+ add1) (+ table[idx]) -> Act (fp32 and/or bf16 planes).  planes_out = (hi, lo):
    destination plane views (possibly rows of a longer padded sequence: batched row map)."""
    lib = _lib.load()
    if isinstance(x, Act):
        x = x.f32
    x = uniform_rows(x)
    residual = None if residual is None else uniform_rows(residual)
    add1 = None if add1 is None else uniform_rows(add1)
    M, H, ldx = _rows2d(x)
    final = None
    if out is not None and not _is_uniform(out):
        final, out = out, None
    if out is None and (want_f32 or final is not None):
        out = torch.empty(*x.shape, dtype=torch.float32, device=x.device)
    ldy = _rows2d(out)[2] if out is not None else 0
    hi = lo = None
    ldp, rpb, bs = H, 0, 0
    if planes_out is not None:
        hi, lo = planes_out
        assert hi.shape == x.shape and hi.stride() == lo.stride() and hi.dtype == torch.bfloat16
        _, _, ldp, rpb, bs = _rows_map(hi)
    elif want_planes:
        hi, lo = _planes_like(x.shape, x.device)
    ldr = ld1 = 0
    if residual is not None:
        _, _, ldr = _rows2d(residual)
    if add1 is not None:
        _, _, ld1 = _rows2d(add1)
    if idx is not None:
        idx = idx.reshape(-1).to(torch.int64).contiguous()
        assert idx.numel() == M
    _timed("layernorm", 0.0, lambda: _lib.check(
        lib.gridmm_layernorm_map(_p(x), ldx, _p(residual), ldr, _p(gamma), _p(beta), float(eps), _p(out), ldy,
                                 _p(add1), ld1, _p(table), _p(idx), _p(hi), _p(lo), ldp, rpb, bs, M, H, _stream()),
        "gridmm_layernorm"))
    if final is not None:
        copy_rows(out, final, 0)
        out = final
    return Act(out, hi, lo)


# Tiled weight planes for the BK = 32 tiles (PackedLinear.tiled): every 1-KiB DMA piece of W is one contiguous KiB.  Measured
# in the B = 32 step: -10..-15 us with the Python-issued GEMMs alone (profiles/r4_tiled_weights.txt).  GRIDMM_WT=0: row-major only.
WT = bool(int(os.environ.get("GRIDMM_WT", "1")))


def attention(q, k, v, kmask, heads=12, scale=None, want_f32=False, want_planes=True):
    """q (B,Sq,H*64) / k,v (B,Sk,H*64) fp32, possibly strided views into fused QKV buffers; kmask (B,Sk)
    uint8/bool.  Returns an Act (bf16 planes for the output projection by default)."""
    lib = _lib.load()
    B, Sq, HD = q.shape
    Sk = k.shape[1]
    assert HD == heads * 64 and k.shape[2] == HD and v.shape[2] == HD
    for t in (q, k, v):
        assert t.stride(2) == 1 and t.dtype == torch.float32
    if scale is None:
        scale = 1.0 / math.sqrt(64.0)
    out = torch.empty(B, Sq, HD, dtype=torch.float32, device=q.device) if want_f32 else None
    hi = lo = None
    if want_planes:
        hi, lo = _planes_like((B, Sq, HD), q.device)
    if kmask is not None:
        if kmask.dtype == torch.bool:
            kmask = kmask.view(torch.uint8)
        assert kmask.shape == (B, Sk) and kmask.stride(1) == 1
    _timed("attention", 4.0 * B * Sq * Sk * HD, lambda: _lib.check(lib.gridmm_attention(
        _p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1), _p(v), v.stride(0), v.stride(1),
        _p(kmask), kmask.stride(0) if kmask is not None else 0, _p(out), Sq * HD, HD, _p(hi), _p(lo), Sq * HD, HD,
        B, heads, Sq, Sk, float(scale), _stream()), "gridmm_attention"))
    return Act(out, hi, lo)


def attention_planes(q, k, v, kmask, heads=12, scale=None, want_f32=False, want_planes=True):
    """bf16x3 attention.  q, k, v: (hi, lo) pairs of bf16 plane views (B,S,H*64) -- typically slices of the
    fused QKV GEMM's plane output.  Returns an Act (planes for the output projection by default)."""
    lib = _lib.load()
    qh, ql = q
    kh, kl = k
    vh, vl = v
    B, Sq, HD = qh.shape
    Sk = kh.shape[1]
    assert HD == heads * 64 and kh.shape[2] == HD and vh.shape[2] == HD
    for t in (qh, ql, kh, kl, vh, vl):
        assert t.dtype == torch.bfloat16 and t.stride(2) == 1
    assert qh.stride() == ql.stride() and kh.stride() == kl.stride() and vh.stride() == vl.stride()
    if scale is None:
        scale = 1.0 / math.sqrt(64.0)
    Skp = (Sk + 31) // 32 * 32
    dev = qh.device
    th = torch.empty(B, heads, Skp // 32, 64, 32, dtype=torch.bfloat16, device=dev)   # re-tiled V (see attention.hip)
    tl = torch.empty_like(th)
    _timed("transpose_v", 0.0, lambda: _lib.check(
        lib.gridmm_transpose_v(_p(vh), _p(vl), vh.stride(0), vh.stride(1), _p(th), _p(tl), B, heads, Sk, Skp,
                               _stream()), "gridmm_transpose_v"))
    out = torch.empty(B, Sq, HD, dtype=torch.float32, device=dev) if want_f32 else None
    hi = lo = None
    if want_planes:
        hi, lo = _planes_like((B, Sq, HD), dev)
    if kmask is not None:
        if kmask.dtype == torch.bool:
            kmask = kmask.view(torch.uint8)
        assert kmask.shape == (B, Sk) and kmask.stride(1) == 1
    _timed("attention", 4.0 * B * Sq * Sk * HD, lambda: _lib.check(lib.gridmm_attention_planes(
        _p(qh), _p(ql), qh.stride(0), qh.stride(1), _p(kh), _p(kl), kh.stride(0), kh.stride(1), _p(th), _p(tl), Skp,
        _p(kmask), kmask.stride(0) if kmask is not None else 0, _p(out), Sq * HD, HD, _p(hi), _p(lo), Sq * HD, HD,
        B, heads, Sq, Sk, float(scale), _stream()), "gridmm_attention_planes"))
    return Act(out, hi, lo)


def attention_rows(q, k, v, kmask, heads=12, scale=None, want_f32=False, want_planes=True, cfg=0, k2=None, v2=None):
    """bf16x3 attention straight from row-major planes.  q, k, v: (hi, lo) pairs of bf16 plane views (B,S,H*64) --
    typically column slices of the fused QKV / KV GEMM outputs; K and V rows are staged in LDS by the kernel (no
    re-tiling pass).  k2 / v2: (hi, lo) pairs holding the LAST keys of the context in a second buffer (same strides for
    both; gridmm_attention_rows_seg).  Returns an Act (planes for the output projection by default)."""
    lib = _lib.load()
    qh, ql = q
    kh, kl = k
    vh, vl = v
    B, Sq, HD = qh.shape
    S1 = kh.shape[1]
    Sk = S1 + (k2[0].shape[1] if k2 is not None else 0)
    assert HD == heads * 64 and kh.shape[2] == HD and vh.shape == kh.shape
    for t in (qh, ql, kh, kl, vh, vl):
        assert t.dtype == torch.bfloat16 and t.stride(2) == 1
    assert qh.stride() == ql.stride() and kh.stride() == kl.stride() and vh.stride() == vl.stride()
    if scale is None:
        scale = 1.0 / math.sqrt(64.0)
    dev = qh.device
    out = torch.empty(B, Sq, HD, dtype=torch.float32, device=dev) if want_f32 else None
    hi = lo = None
    if want_planes:
        hi, lo = _planes_like((B, Sq, HD), dev)
    if kmask is not None:
        if kmask.dtype == torch.bool:
            kmask = kmask.view(torch.uint8)
        assert kmask.shape == (B, Sk) and kmask.stride(1) == 1
    if k2 is not None:
        (k2h, k2l), (v2h, v2l) = k2, v2
        for t in (k2h, k2l, v2h, v2l):
            assert t.dtype == torch.bfloat16 and t.stride(2) == 1 and t.shape == k2h.shape and t.stride() == k2h.stride()
        _timed("attention", 4.0 * B * Sq * Sk * HD, lambda: _lib.check(lib.gridmm_attention_rows_seg(
            _p(qh), _p(ql), qh.stride(0), qh.stride(1), _p(kh), _p(kl), kh.stride(0), kh.stride(1), _p(vh), _p(vl),
            vh.stride(0), vh.stride(1), S1, _p(k2h), _p(k2l), _p(v2h), _p(v2l), k2h.stride(0), k2h.stride(1), _p(kmask),
            kmask.stride(0) if kmask is not None else 0, _p(out), Sq * HD, HD, _p(hi), _p(lo), Sq * HD, HD, B, heads, Sq, Sk,
            float(scale), _stream()), "gridmm_attention_rows_seg"))
        return Act(out, hi, lo)
    _timed("attention", 4.0 * B * Sq * Sk * HD, lambda: _lib.check(lib.gridmm_attention_rows_cfg(
        _p(qh), _p(ql), qh.stride(0), qh.stride(1), _p(kh), _p(kl), kh.stride(0), kh.stride(1), _p(vh), _p(vl),
        vh.stride(0), vh.stride(1), _p(kmask), kmask.stride(0) if kmask is not None else 0, _p(out), Sq * HD, HD,
        _p(hi), _p(lo), Sq * HD, HD, B, heads, Sq, Sk, float(scale), int(cfg), _stream()), "gridmm_attention_rows"))
    return Act(out, hi, lo)


class _CLinear(ctypes.Structure):
    _fields_ = [("w_hi", ctypes.c_void_p), ("w_lo", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("N", ctypes.c_int),
                ("K", ctypes.c_int), ("Kp", ctypes.c_int), ("wt_hi", ctypes.c_void_p), ("wt_lo", ctypes.c_void_p)]


def _clinear(pw):
    """gridmm_linear_t of a PackedLinear (with its tiled planes when WT is on)."""
    th, tl = pw.tiled() if (WT and pw.Kp % 32 == 0) else (None, None)
    return _CLinear(pw.hi.data_ptr(), pw.lo.data_ptr(), pw.bias.data_ptr() if pw.bias is not None else None, pw.N, pw.K, pw.Kp,
                    th.data_ptr() if th is not None else None, tl.data_ptr() if tl is not None else None)


class _CLn(ctypes.Structure):
    _fields_ = [("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("eps", ctypes.c_float)]


class _CXLayer(ctypes.Structure):
    _fields_ = [(n, _CLinear) for n in ("xq", "xo", "sqkv", "so", "ffn_i", "ffn_o")] + \
               [(n, _CLn) for n in ("x_ln", "s_ln", "f_ln")]


class XLayerWeights:
    """gridmm_xlayer_t of one cross-modal layer: packed Linears (PackedLinear) and LayerNorm modules; keeps them alive."""

    def __init__(self, xq, xo, sqkv, so, ffn_i, ffn_o, x_ln, s_ln, f_ln):
        self.keep = (xq, xo, sqkv, so, ffn_i, ffn_o, x_ln, s_ln, f_ln)
        c = _CXLayer()
        for name, pw in zip(("xq", "xo", "sqkv", "so", "ffn_i", "ffn_o"), self.keep[:6]):
            setattr(c, name, _clinear(pw))
        for name, ln in zip(("x_ln", "s_ln", "f_ln"), self.keep[6:]):
            setattr(c, name, _CLn(ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps)))
        self.c = c
        self.H, self.I = xq.N, ffn_i.N


def xattn_layer(w, x, kv, k_col, v_col, ctx_mask, self_mask, heads=12, planes_out=None, kv2=None):
    """One GraphLXRTXLayer as ONE C call (gridmm_xattn_layer_fwd): x Act (f32 + planes) (B, Sq, H); kv Act planes
    (B, Sk, n*H) holding the context's K / V projections at columns k_col / v_col.  Returns Act(f32 + planes).
    planes_out = (hi, lo): where the output planes go (views, possibly rows of a longer padded sequence).  The scratch is
    allocated per call (stream-ordered by the caching allocator: safe for eager calls, several graphs and streams)."""
    lib = _lib.load()
    B, Sq, H = x.f32.shape
    Sk1 = kv.hi.shape[1]
    Sk = Sk1
    k2 = (None, None, 0, 0, 0, 0)
    if kv2 is not None:     # (Act planes (B, S2, .), k column, v column): the last S2 context rows live in a second buffer
        a2, k2_col, v2_col = kv2
        assert a2.hi.stride(2) == 1 and a2.hi.stride() == a2.lo.stride() and a2.hi.shape[0] == B
        Sk = Sk1 + a2.hi.shape[1]
        k2 = (a2.hi, a2.lo, a2.hi.stride(0), a2.hi.stride(1), int(k2_col), int(v2_col))
    dev = x.f32.device
    assert x.f32.is_contiguous() and x.hi.is_contiguous() and kv.hi.stride(2) == 1 and kv.hi.stride() == kv.lo.stride()
    need = lib.gridmm_xattn_layer_workspace(B, Sq, H, w.I)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    y = torch.empty(B, Sq, H, dtype=torch.float32, device=dev)
    rpb, bs = 0, 0
    if planes_out is not None:
        hi, lo = planes_out
        assert hi.shape == (B, Sq, H) and hi.stride() == lo.stride() and hi.stride(1) == H and hi.stride(2) == 1
        if hi.stride(0) != Sq * H:
            rpb, bs = Sq, hi.stride(0)
    else:
        hi, lo = _planes_like((B, Sq, H), dev)
    cm = ctx_mask.view(torch.uint8) if ctx_mask.dtype == torch.bool else ctx_mask
    sm = self_mask.view(torch.uint8) if self_mask.dtype == torch.bool else self_mask
    assert cm.stride(1) == 1 and sm.stride(1) == 1
    _lib.check(lib.gridmm_xattn_layer_fwd(ctypes.byref(w.c), _p(x.f32), _p(x.hi), _p(x.lo), _p(kv.hi), _p(kv.lo),
                                          kv.hi.stride(0), kv.hi.stride(1), int(k_col), int(v_col), Sk1, _p(k2[0]), _p(k2[1]),
                                          k2[2], k2[3], k2[4], k2[5], _p(cm), cm.stride(0),
                                          _p(sm), sm.stride(0), _p(y), _p(hi), _p(lo), rpb, bs, _p(ws), need,
                                          B, Sq, Sk, heads, _stream()), "gridmm_xattn_layer_fwd")
    return Act(y, hi, lo)


class _CProblem(ctypes.Structure):
    _fields_ = [("A_hi", ctypes.c_void_p), ("A_lo", ctypes.c_void_p), ("lda", ctypes.c_int), ("a_rpb", ctypes.c_int),
                ("a_bs", ctypes.c_int64), ("W_hi", ctypes.c_void_p), ("W_lo", ctypes.c_void_p), ("Kp", ctypes.c_int),
                ("bias", ctypes.c_void_p), ("C", ctypes.c_void_p), ("ldc", ctypes.c_int), ("C_hi", ctypes.c_void_p),
                ("C_lo", ctypes.c_void_p), ("ldp", ctypes.c_int), ("M", ctypes.c_int), ("N", ctypes.c_int),
                ("K", ctypes.c_int), ("act", ctypes.c_int)]


def gemm_problem(a_hi, a_lo, lda, M, pw, out, act=ACT_NONE, a_rpb=0, a_bs=0, a_off=0, w_col0=0, K=None, bias=True):
    """One record of linear_grouped: out (M, N) fp32 = act(A W^T + b); A = plane tensors read from element offset a_off
    with row stride lda (batched row map a_rpb / a_bs optional); W = PackedLinear, contraction over its columns
    [w_col0, w_col0 + K)."""
    K = pw.K if K is None else K
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[-1] == pw.N
    return _CProblem(a_hi.data_ptr() + 2 * a_off, a_lo.data_ptr() + 2 * a_off, int(lda), int(a_rpb), int(a_bs),
                     pw.hi.data_ptr() + 2 * w_col0, pw.lo.data_ptr() + 2 * w_col0, pw.Kp,
                     pw.bias.data_ptr() if (bias and pw.bias is not None) else None, out.data_ptr(), pw.N, None, None, 0,
                     int(M), pw.N, int(K), int(act))


def linear_grouped(problems):
    """Several small plane GEMMs in one launch (gridmm_linear_planes_grouped); problems: list of gemm_problem()."""
    lib = _lib.load()
    arr = (_CProblem * len(problems))(*problems)
    work = sum(2.0 * p.M * p.N * p.K for p in problems)
    _timed("linear", work, lambda: _lib.check(lib.gridmm_linear_planes_grouped(arr, len(problems), _stream()),
                                              "gridmm_linear_planes_grouped"))


def linear_wt(lin):
    """nn.Linear(K, H) weight as the [K][H] fp32 image the fused embedding kernels stage in LDS (cache it per weight version)."""
    return lin.weight.detach().float().t().contiguous()


def cells_embed(proj, pos_fts, lin, ln, occ, out, mask, tail_mask=None, wT=None, c_pad=N_CELLS):
    """cells_compact with the grid position embedding (lin = nn.Linear(K, H), ln = nn.LayerNorm) computed inside; mask:
    (B, >= 196 + n_tail) uint8 view with any row stride; tail_mask (B, n_tail) is copied behind the 196 cell bits.
    Returns (n_cells, cmax) int32 tensors."""
    lib = _lib.load()
    B, S_pad, H = out.shape
    K = pos_fts.shape[-1]
    dev = out.device
    n_cells = torch.empty(B, dtype=torch.int32, device=dev)
    cmax = torch.empty(1, dtype=torch.int32, device=dev)
    assert out.is_contiguous() and mask.stride(1) == 1 and mask.dtype == torch.uint8 and proj.is_contiguous()
    pos_fts = pos_fts.float().contiguous()
    n_tail = 0 if tail_mask is None else tail_mask.shape[1]
    assert tail_mask is None or tail_mask.is_contiguous()
    wT = linear_wt(lin) if wT is None else wT
    assert wT.shape == (K, H) and wT.is_contiguous()
    _timed("cells_embed", 0.0, lambda: _lib.check(lib.gridmm_cells_embed(
        _p(proj), _p(pos_fts), K, _p(wT), _p(lin.bias), _p(ln.weight), _p(ln.bias), float(ln.eps), _p(occ),
        _p(out), _p(mask), mask.stride(0), _p(tail_mask), n_tail, _p(n_cells), _p(cmax), B, H, S_pad, int(c_pad), _stream()),
        "gridmm_cells_embed"))
    return n_cells, cmax


class _CEmbedSeg(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_void_p), ("K", ctypes.c_int), ("W", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("eps", ctypes.c_float), ("add1", ctypes.c_void_p),
                ("ld1", ctypes.c_int), ("table", ctypes.c_void_p), ("idx", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("out_hi", ctypes.c_void_p), ("out_lo", ctypes.c_void_p), ("out_rpb", ctypes.c_int),
                ("out_bs", ctypes.c_int64), ("M", ctypes.c_int)]


def embed_seg(pos, lin, ln, add1, out, table=None, idx=None, planes=None, wT=None):
    """One segment of node_embed: out (B, R, H) fp32 view (rows of a longer sequence allowed) = LN(lin(pos)) + add1
    (+ table[idx]); planes = (hi, lo) views with the SAME strides as out; wT = linear_wt(lin) (cached by the caller)."""
    wT = linear_wt(lin) if wT is None else wT
    keep = [wT]
    pos = pos.float().contiguous()
    add1 = add1.float().contiguous()
    keep += [pos, add1]
    B, R, H = out.shape
    assert out.stride(2) == 1 and out.stride(1) == H and pos.shape[:2] == (B, R) and add1.shape == (B, R, H)
    rpb, bs = (0, 0) if out.stride(0) == R * H else (R, out.stride(0))
    if idx is not None:
        idx = idx.reshape(-1).to(torch.int64).contiguous()
        keep.append(idx)
    hi = lo = None
    if planes is not None:
        hi, lo = planes
        assert hi.stride() == out.stride() and lo.stride() == out.stride()
    assert wT.shape == (pos.shape[-1], H) and wT.is_contiguous()
    seg = _CEmbedSeg(pos.data_ptr(), pos.shape[-1], wT.data_ptr(), lin.bias.data_ptr(), ln.weight.data_ptr(),
                     ln.bias.data_ptr(), float(ln.eps), add1.data_ptr(), H, table.data_ptr() if table is not None else None,
                     idx.data_ptr() if idx is not None else None, out.data_ptr(), hi.data_ptr() if hi is not None else None,
                     lo.data_ptr() if lo is not None else None, rpb, bs, B * R)
    seg._keep = keep
    return seg


def node_embed(segs, H, gmap_m, vp_m, txt_m, kv_masks, kv_col0, q_masks):
    """Position embeddings of the map nodes / candidate views (embed_seg records) + the byte masks of the [cells | nodes |
    txt] context (written from column kv_col0 of kv_masks on) and of the [nodes | views] queries, one launch."""
    lib = _lib.load()
    arr = (_CEmbedSeg * len(segs))(*segs)
    B, G = gmap_m.shape
    V, L = vp_m.shape[1], txt_m.shape[1]
    for m in (gmap_m, vp_m, txt_m, q_masks):
        assert m.dtype == torch.uint8 and m.is_contiguous()
    assert kv_masks.dtype == torch.uint8 and kv_masks.stride(1) == 1 and q_masks.shape == (B, G + V)
    _timed("node_embed", 0.0, lambda: _lib.check(lib.gridmm_node_embed(
        arr, len(segs), H, _p(gmap_m), G, _p(vp_m), V, _p(txt_m), L, _p(kv_masks), kv_masks.stride(0), int(kv_col0),
        _p(q_masks), B, _stream()), "gridmm_node_embed"))


class _CClsTail(ctypes.Structure):
    _fields_ = [("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("eps", ctypes.c_float), ("w", ctypes.c_void_p),
                ("b0", ctypes.c_void_p)]


def cls_tail(net):
    """ClsPrediction.net -> the LayerNorm . Linear(H, 1) tail record (net = [Linear, ReLU, LayerNorm, Linear])."""
    return _CClsTail(net[2].weight.data_ptr(), net[2].bias.data_ptr(), float(net[2].eps), net[3].weight.data_ptr(),
                     net[3].bias.data_ptr() if net[3].bias is not None else None)


def nav_heads(h_gl, fuse_a, fuse_b, fuse_bias, h_grid, tails, gmap_masks, gmap_visited, vp_nav_masks, vp_obj_masks,
              cand_of_node, cand_visited, G, V):
    """LN . w tails of the heads + masking + fusion (gridmm_nav_heads).  tails: [fuse, global, local, grid, object] cls_tail
    records (object may be None).  Returns (global, local, grid, fused, obj or None)."""
    lib = _lib.load()
    B = gmap_masks.shape[0]
    H = h_grid.shape[-1]
    dev = h_gl.device
    outs = [torch.empty(B, G, device=dev), torch.empty(B, V, device=dev), torch.empty(B, G, device=dev),
            torch.empty(B, G, device=dev)]
    obj = torch.empty(B, V, device=dev) if vp_obj_masks is not None else None
    t = [x if x is not None else tails[3] for x in tails]
    if fuse_a is None:
        t[0] = tails[1]
    arr = (_CClsTail * 5)(*t)
    ws = torch.empty(int(lib.gridmm_nav_heads_workspace(B, G, V)), dtype=torch.uint8, device=dev)
    _timed("nav_heads", 0.0, lambda: _lib.check(lib.gridmm_nav_heads(
        _p(h_gl), h_gl.shape[-1], _p(fuse_a), _p(fuse_b), _p(fuse_bias), _p(h_grid), arr, _p(gmap_masks), _p(gmap_visited),
        _p(vp_nav_masks), _p(vp_obj_masks), _p(cand_of_node), _p(cand_visited), _p(outs[0]), _p(outs[1]), _p(outs[2]),
        _p(outs[3]), _p(obj), _p(ws), B, G, V, H, _stream()), "gridmm_nav_heads"))
    return outs[0], outs[1], outs[2], outs[3], obj


def tokens_to_slab(tokens, slot, n_views):
    """tokens (B * n_views, T, D) fp32 (T = 1 class token + patches) -> slot (B, n_views * (T-1), D) fp16 view of the grid
    memory's slab (GridMemoryBatch.next_slot()): the patch tokens land where fill_gridmap expects the new observation."""
    lib = _lib.load()
    N, T, D = tokens.shape
    B = N // n_views
    assert N == B * n_views and tokens.dtype == torch.float32 and tokens.is_contiguous()
    assert slot.dtype == torch.float16 and slot.shape == (B, n_views * (T - 1), D) and slot.stride(1) == D and slot.stride(2) == 1
    _lib.check(lib.gridmm_tokens_to_slab(_p(tokens), T, D, _p(slot), slot.stride(0), B, n_views, _stream()),
               "gridmm_tokens_to_slab")
    return slot


def ln_dot(x, gamma, beta, eps, w, b0, out=None):
    lib = _lib.load()
    if isinstance(x, Act):
        x = x.f32
    x = uniform_rows(x)
    M, H, ldx = _rows2d(x)
    if out is None:
        out = torch.empty(*x.shape[:-1], dtype=torch.float32, device=x.device)
    _lib.check(lib.gridmm_ln_dot(_p(x), ldx, _p(gamma), _p(beta), float(eps), _p(w), _p(b0), _p(out), M, H,
                                 _stream()), "gridmm_ln_dot")
    return out


def copy_planes(src, dst, dst_row0=0):
    """dst.hi/lo[:, dst_row0:dst_row0+rows] = src.hi/lo for Act pairs; one launch when both pairs are the halves of one
    allocation (what _planes_like hands out), else two."""
    def paired(a):   # lo sits exactly B batch strides behind hi (also true for row slices of such a pair)
        return (a.hi.shape == a.lo.shape and a.hi.stride() == a.lo.stride() and a.hi.stride(2) == 1
                and a.hi.untyped_storage().data_ptr() == a.lo.untyped_storage().data_ptr()
                and a.lo.data_ptr() - a.hi.data_ptr() == a.hi.shape[0] * a.hi.stride(0) * a.hi.element_size())
    if paired(src) and paired(dst):
        B, rows, H = src.hi.shape
        s2 = torch.as_strided(src.hi, (2 * B, rows, H), src.hi.stride())
        d2 = torch.as_strided(dst.hi, (2 * B,) + tuple(dst.hi.shape[1:]), dst.hi.stride())
        copy_rows(s2, d2, dst_row0)
    else:
        copy_rows(src.hi, dst.hi, dst_row0)
        copy_rows(src.lo, dst.lo, dst_row0)
    return dst


def copy_rows(src, dst, dst_row0=0):
    """dst[:, dst_row0:dst_row0+rows] = src   for (B, rows, H) fp32 tensors (row-contiguous)."""
    lib = _lib.load()
    if src.dtype == torch.bfloat16:   # bf16 planes move as packed words
        assert dst.dtype == torch.bfloat16 and src.shape[2] % 8 == 0 and src.stride(1) % 2 == 0
        src, dst = src.view(torch.float32), dst.view(torch.float32)
    B, rows, H = src.shape
    assert dst.shape[0] == B and dst.shape[2] == H and src.stride(2) == 1 and dst.stride(2) == 1
    d = dst[:, dst_row0:dst_row0 + rows]
    _lib.check(lib.gridmm_copy_rows(_p(src), src.stride(0), src.stride(1), _p(d), d.stride(0), d.stride(1),
                                    B, rows, H, _stream()), "gridmm_copy_rows")
    return dst


def cells_compact(proj, pos_emb, occ, out, mask):
    """Compact cells into rows [0,196) of out (B,S_pad,H) / mask (B,S_pad); returns (n_cells, cmax) int32 tensors."""
    lib = _lib.load()
    B, S_pad, H = out.shape
    n_cells = torch.empty(B, dtype=torch.int32, device=out.device)
    cmax = torch.empty(1, dtype=torch.int32, device=out.device)
    assert out.is_contiguous() and mask.is_contiguous() and mask.shape == (B, S_pad)
    _lib.check(lib.gridmm_cells_compact(_p(proj), _p(pos_emb), _p(occ), _p(out), _p(mask), _p(n_cells), _p(cmax),
                                        B, H, S_pad, _stream()), "gridmm_cells_compact")
    return n_cells, cmax


def fuse_logits(g_raw, l_raw, grid_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, cand_of_node,
                cand_visited):
    lib = _lib.load()
    B, G = g_raw.shape
    V = l_raw.shape[1]
    dev = g_raw.device
    outs = [torch.empty(B, G, device=dev), torch.empty(B, V, device=dev), torch.empty(B, G, device=dev),
            torch.empty(B, G, device=dev)]
    _lib.check(lib.gridmm_fuse_logits(_p(g_raw), _p(l_raw), _p(grid_raw), _p(fuse_raw), _p(gmap_masks),
                                      _p(gmap_visited), _p(vp_nav_masks), _p(cand_of_node), _p(cand_visited),
                                      _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), B, G, V, _stream()),
               "gridmm_fuse_logits")
    return outs  # global, local, grid, fused


FLAG_VLNCE = 1


def grid_project(depth, x_off, view_cos, view_sin, pose, n_old, hist_x, hist_y, hist_valid, bbox, half_len,
                 pos_fts, active, n_views, ppv, depth_div, flags=0, max_dist=30.0):
    lib = _lib.load()
    B, cap = hist_x.shape
    depth_f32 = int(depth.dtype == torch.float32)
    view_stride = n_views if view_cos.dim() == 2 else 0       # per-episode view tables (VLN-CE) or shared
    _lib.check(lib.gridmm_grid_project(_p(depth), depth_f32, _p(x_off), _p(view_cos), _p(view_sin), view_stride,
                                       _p(pose), _p(n_old), _p(hist_x), _p(hist_y), _p(hist_valid), _p(bbox),
                                       _p(half_len), _p(pos_fts), _p(active), B, n_views, ppv, cap, float(depth_div),
                                       int(flags), float(max_dist), _stream()), "gridmm_grid_project")


def grid_bin(hist_x, hist_y, hist_valid, n_pts, pose, head_cs, half_len, cell_id, perm, cell_start, flags=0,
             workspace=None, slices=1):
    """slices > 1 (+ workspace (B, slices, 17, 197) int32): the multi-workgroup form for deep memories."""
    lib = _lib.load()
    B, cap = hist_x.shape
    if slices > 1 and workspace is not None:
        assert workspace.dtype == torch.int32 and workspace.numel() >= B * slices * 17 * 197
        _lib.check(lib.gridmm_grid_bin_sliced(_p(hist_x), _p(hist_y), _p(hist_valid), _p(n_pts), _p(pose), _p(head_cs),
                                              _p(half_len), _p(cell_id), _p(perm), _p(cell_start), _p(workspace),
                                              int(slices), B, cap, int(flags), _stream()), "gridmm_grid_bin_sliced")
        return
    _lib.check(lib.gridmm_grid_bin(_p(hist_x), _p(hist_y), _p(hist_valid), _p(n_pts), _p(pose), _p(head_cs),
                                   _p(half_len), _p(cell_id), _p(perm), _p(cell_start), B, cap, int(flags),
                                   _stream()), "gridmm_grid_bin")


def grid_sort_ids(cell_id, n_pts, perm, cell_start):
    lib = _lib.load()
    B, cap = cell_id.shape
    _lib.check(lib.gridmm_grid_sort_ids(_p(cell_id), _p(n_pts), _p(perm), _p(cell_start), B, cap, _stream()),
               "gridmm_grid_sort_ids")


def grid_cell_count_max(cell_start, out):
    """out[0] = max over the batch of the number of non-empty cells (cell_start (B, 198) int32 as the binning wrote it)."""
    lib = _lib.load()
    assert cell_start.dtype == torch.int32 and cell_start.is_contiguous() and cell_start.shape[1] == N_CELLS + 2
    _lib.check(lib.gridmm_grid_cell_count_max(_p(cell_start), _p(out), cell_start.shape[0], _stream()),
               "gridmm_grid_cell_count_max")
    return out


def text_fragments(text_fts, out=None):
    """(B, L, D) fp32 -> MFMA B-fragment planes (fp16 hi|lo)."""
    lib = _lib.load()
    B, L, D = text_fts.shape
    Lt = (L + 15) // 16
    frag = out if out is not None else torch.empty(B, 2, Lt, D // 32, 64, 8, dtype=torch.float16, device=text_fts.device)
    assert frag.shape == (B, 2, Lt, D // 32, 64, 8) and frag.is_contiguous() and frag.dtype == torch.float16
    _lib.check(lib.gridmm_text_fragments(_p(text_fts.contiguous()), _p(frag), B, L, D, _stream()),
               "gridmm_text_fragments")
    return frag


def two_pass_aggregation(D, L):
    """Shapes whose aggregation is a relevance pass + an accumulation pass (every other shape reads the slab once already)."""
    return D == 768 or not 33 <= L <= 96


def grid_aggregate_incremental(slab, perm, cell_start, text_frag, L, n_pts, active, n_new, state, n_chunks=None, full=False):
    """grid_aggregate for a device-resident memory on the two-pass shapes, with the relevance pass restricted to the points
    that have no value yet (gridmm_grid_aggregate_incremental: normally the observation just appended) -> (cells, occ), or
    None when the shape is outside the two-pass kernels' range (nothing was launched that matters: call grid_aggregate).
    state: dict owned by the memory with 'hist' (B, cap) f32, 'valid' (B,) int32, 'scratch' uint8, 'rel' (B, cap) f32
    (GridMemoryBatch.relevance_cache).  full: no point has a value yet (first step of an episode): the plain passes + one
    launch that keeps their values (always correct, only cheaper than the general sequence on a cold memory)."""
    lib = _lib.load()
    B, cap, D = slab.shape
    assert slab.dtype == torch.float16 and slab.is_contiguous()
    if n_chunks is None:
        n_chunks = max(1, min(N_CELLS, -(-256 // B)))
        if os.environ.get("GRIDMM_AGG_CHUNKS"):
            n_chunks = int(os.environ["GRIDMM_AGG_CHUNKS"])
    dev = slab.device
    cells = torch.empty(B, N_CELLS, D, dtype=torch.float32, device=dev)
    occ = torch.empty(B, N_CELLS, dtype=torch.uint8, device=dev)
    chunks = torch.empty(int(lib.gridmm_grid_aggregate_workspace(B, D, n_chunks)), dtype=torch.uint8, device=dev)
    status = []

    def launch():
        rc = lib.gridmm_grid_aggregate_incremental(_p(slab), _p(perm), _p(cell_start), _p(text_frag), _p(n_pts), _p(active),
                                                   int(n_new), _p(state["hist"]), _p(state["valid"]), _p(state["scratch"]),
                                                   _p(cells), _p(occ), _p(state["rel"]), _p(chunks), B, cap, D, L, n_chunks,
                                                   int(bool(full)), _stream())
        status.append(rc)
        if rc != -1:                 # GRIDMM_EINVAL: shape outside the range, the caller falls back
            _lib.check(rc, "gridmm_grid_aggregate_incremental")
    if TIMER is not None:
        TIMER.last_aggregate = launch        # (a re-launch recomputes the last observation's values again: idempotent)
    _timed("grid_aggregate", 0.0, launch)
    if status[-1] != 0:
        return None
    global LAST_AGGREGATE_RC
    LAST_AGGREGATE_RC = 2            # the incremental two-pass path
    return cells, occ


def grid_aggregate(slab, perm, cell_start, text_frag, L, n_chunks=None, want_relevance=False, n_points=None,
                   want_amax=False):
    """slab (B,cap,D) fp16 -> cells (B,196,D) fp32, occ (B,196) uint8 [, relevance (B,cap) fp32: w of the point at
    SORTED position p, i.e. of slot perm[b, p]].  n_points: host-known upper bound of the points per episode (defaults to
    the slab capacity); only steers the chunking.  want_amax (training): also returns the arg-max token of every point by
    sorted position, (B,cap) int32 -- or None when the shape ran on the generic kernel, which does not produce it."""
    lib = _lib.load()
    B, cap, D = slab.shape
    assert slab.dtype == torch.float16 and slab.is_contiguous()
    if n_chunks is None:
        # one workgroup per CU (256): the chunks are equal shares of an episode's sorted points (cells that a cut splits are
        # merged from their pieces), so one round of workgroups is balanced at any memory depth
        n_chunks = max(1, min(N_CELLS, -(-256 // B)))
        if os.environ.get("GRIDMM_AGG_CHUNKS"):
            n_chunks = int(os.environ["GRIDMM_AGG_CHUNKS"])
    dev = slab.device
    cells = torch.empty(B, N_CELLS, D, dtype=torch.float32, device=dev)
    occ = torch.empty(B, N_CELLS, dtype=torch.uint8, device=dev)
    # the two-pass paths (D = 768; instructions outside 33..96 tokens at D <= 512) need the relevance buffer as their
    # intermediate: allocated even when not asked for
    if want_relevance or want_amax:
        rel = torch.zeros(B, cap, dtype=torch.float32, device=dev)
    else:                                   # scratch of a two-pass path (only valid positions are written / read)
        rel = torch.empty(B, cap, dtype=torch.float32, device=dev) if (D == 768 or not 33 <= L <= 96) else None
        if os.environ.get("GRIDMM_AGG_FORCE_GENERIC") == "1":
            rel = None                      # tools/bench_agg_long.py: without the intermediate the dispatcher falls back
    chunks = torch.empty(int(lib.gridmm_grid_aggregate_workspace(B, D, n_chunks)), dtype=torch.uint8, device=dev)
    amax = torch.empty(B, cap, dtype=torch.int32, device=dev) if want_amax else None
    status = []

    def launch():
        rc = lib.gridmm_grid_aggregate_train(_p(slab), _p(perm), _p(cell_start), _p(text_frag), _p(cells), _p(occ),
                                             _p(rel), _p(amax), _p(chunks), B, cap, D, L, n_chunks, _stream())
        status.append(rc)
        global LAST_AGGREGATE_RC
        LAST_AGGREGATE_RC = rc       # 0: a pipelined path (one pass, or relevance + accumulation passes); 1: the generic kernel
        _lib.check(min(rc, 0), "gridmm_grid_aggregate_train")
    if TIMER is not None:
        TIMER.last_aggregate = launch        # bench.py re-launches it inside a hipGraph for the device-side duration
    _timed("grid_aggregate", 0.0, launch)
    if want_amax:
        return cells, occ, rel, (amax if status[-1] == 0 else None)
    return (cells, occ, rel) if want_relevance else (cells, occ)
