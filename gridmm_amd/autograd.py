"""Differentiable wrappers over the HIP kernels (training: SURVEY.md §8 rows a11 / a13 / a14).

The reference trains through torch autograd over nn.Linear / LayerNorm / softmax attention / GELU
(fine-tune: map_nav_src/r2r/agent_base.py:164-211; pre-train: pretrain_src/train_r2r.py:231-327).  Here
every one of those ops is a torch.autograd.Function whose forward AND backward are HIP kernels of
libgridmm_hip.so; torch only records the graph and moves/gathers/concatenates tensors.

  linear      y = x W^T + b            fwd / dX: MFMA bf16x3 tile GEMM (gridmm_linear_planes)
                                       dW = dY^T X: the same GEMM over transposed planes (gridmm_transpose_split),
                                       db: column sums from the same transpose pass
  layer_norm  LN(x (+ r)) g + b        gridmm_layernorm / gridmm_layernorm_bwd
  gelu, relu                           gridmm_activation
  attention   softmax(QK^T s + m) V    gridmm_attention_train / gridmm_attention_bwd (exact-fp32 MFMA), optional
                                       dropout on the probabilities from a counter-based hash (same mask fwd / bwd)
  grid_aggregate                       gridmm_grid_aggregate / gridmm_grid_aggregate_bwd (grad w.r.t. text_fts)

There is no CPU / eager fallback: the functions raise on non-GPU tensors (ops._p).
"""
import contextlib
import ctypes
import math
import weakref

import torch

from . import _lib, hostsync as hs, ops
from .ops import _p, _rows2d, _stream
from .pinned import upload_blob


# ------------------------------------------------------------------------------------------------
# packed-weight cache: bf16 hi/lo planes of W (forward) and W^T (dX), rebuilt when the parameter changes
# ------------------------------------------------------------------------------------------------
class _WeightCache:
    """bf16 hi/lo planes of W (forward GEMM) and W^T (dX GEMM) per nn.Parameter, keyed by the parameter OBJECT
    (weak) and its version counter, so optimizer steps / load_state_dict re-pack.  Temporaries (torch.cat of q|k|v
    weights) are never cached: the caching allocator recycles their addresses."""

    def __init__(self):
        self._ent = {}   # id(param) -> (weakref to param, entry); the weakref's callback drops the entry
        self._grp = {}   # ids of a group's members -> (weakrefs, entry): fused planes of weights that share their input
        self._member = {}   # id(param) -> (weakref, group key, first row in the fused planes)

    @staticmethod
    def _pack(w, transposed):
        src = w.detach().float()
        if transposed and src.dim() == 2 and src.shape[1] % 8 == 0 and src.is_contiguous():
            # W^T planes (K, roundup(N, 32)) in one pass over W (the transposing splitter of the activations) instead of
            # a transposed copy + a split: one launch less and no fp32 W^T per weight and step
            hi, lo, _, Np, _ = transpose_split(src)
            pw = ops.PackedLinear.__new__(ops.PackedLinear)
            pw.hi, pw.lo, pw.bias, pw.N, pw.K, pw.Kp = hi, lo, None, src.shape[1], src.shape[0], Np
            return pw
        return ops.PackedLinear(src.t().contiguous() if transposed else src.contiguous(), None)

    @staticmethod
    def _pack_both(w, into=None):
        """into = (PackedLinear of W, PackedLinear of W^T) of an earlier call: re-pack IN PLACE (the plane buffers of a trained
        weight keep their addresses for the life of the cache entry: captured training steps and the optimizer's plane
        records point at them)."""
        src = w.detach()
        if into is not None:
            transpose_split(src, want_rows=True, out=(into[1].hi, into[1].lo, into[0].hi, into[0].lo))
            return into
        hi, lo, _, Np, rows = transpose_split(src, want_rows=True)
        out = []
        for h, l, N, K, Kp in ((rows.hi, rows.lo, src.shape[0], src.shape[1], src.shape[1]),
                               (hi, lo, src.shape[1], src.shape[0], Np)):
            pw = ops.PackedLinear.__new__(ops.PackedLinear)
            pw.hi, pw.lo, pw.bias, pw.N, pw.K, pw.Kp = h, l, None, N, K, Kp
            out.append(pw)
        return out

    def _make_get(self, ent, w):
        made = ent.get("ver")

        def get(transposed, ent=ent, w=w):
            if ent.get("ver") != made:
                # a getter outlives its forward inside an autograd node (dX reads W^T in backward): the entry now belongs to a
                # NEWER version of the parameter (optimizer step / load_state_dict rewrote the persistent planes in place)
                raise RuntimeError("a weight was modified in place between a Linear's forward and its backward; its planes are "
                                   "persistent buffers shared across steps -- run backward before updating the weights")
            if transposed not in ent:
                if (w.requires_grad and w.dim() == 2 and w.dtype == torch.float32   # (grad mode is off inside Function.forward)
                        and w.shape[1] % 32 == 0 and w.shape[0] % 8 == 0 and w.is_contiguous()):
                    # a trained weight is needed in both orientations every step (forward: W, dX: W^T): one pass over
                    # it writes both pairs of planes instead of a split launch + a transposing launch
                    ent[False], ent[True] = self._pack_both(w, ent.get("keep"))
                    ent["keep"] = (ent[False], ent[True])         # persistent plane buffers of this parameter
                else:
                    ent[transposed] = self._pack(w, transposed)
            return ent[transposed]
        return get

    def _slot_of(self, w):
        slot = self._ent.get(id(w))
        return slot if (slot is not None and slot[0]() is w) else None

    def getter(self, w):
        """-> get(transposed) returning the PackedLinear of w or w^T."""
        if not isinstance(w, torch.nn.Parameter):
            return lambda transposed: self._pack(w, transposed)
        ver = (w._version, w.data_ptr(), tuple(w.shape))
        key = id(w)
        slot = self._slot_of(w)
        if slot is None:
            ent = {"ver": ver}
            self._ent[key] = (weakref.ref(w, lambda _r, key=key: self._ent.pop(key, None)), ent)
        elif slot[1]["ver"] != ver:
            # the parameter changed behind the cache (load_state_dict, an update that did not write the planes): the entry
            # keeps its persistent buffers ("keep") and re-packs into them on the next use
            ent = slot[1]
            keep = ent.get("keep")
            ent.clear()
            ent["ver"] = ver
            if keep is not None and tuple(keep[0].hi.shape) == tuple(w.shape):
                ent["keep"] = keep
        else:
            ent = slot[1]
        return self._make_get(ent, w)

    # ---- fused planes of several Parameters that share their input (q | k | v; k | v of one or of all local layers): the
    # members stay separate Parameters, their planes are row blocks of ONE pair of plane buffers
    @staticmethod
    def groupable(ws):
        if len(ws) < 2 or not all(isinstance(w, torch.nn.Parameter) for w in ws):
            return False
        K = ws[0].shape[-1]
        return all(w.dim() == 2 and w.dtype == torch.float32 and w.is_contiguous() and w.shape[1] == K and w.shape[0] % 8 == 0
                   for w in ws) and K % 32 == 0 and any(w.requires_grad for w in ws)

    def _group_slot(self, key):
        slot = self._grp.get(key)
        if slot is not None and any(r() is None for r in slot[0]):
            self._drop_group(key)
            return None
        return slot

    def _drop_group(self, key):
        self._grp.pop(key, None)
        for i in key:
            m = self._member.get(i)
            if m is not None and m[1] == key:
                self._member.pop(i, None)

    def group_getter(self, ws):
        key = tuple(id(w) for w in ws)
        ver = tuple((w._version, w.data_ptr(), tuple(w.shape)) for w in ws)
        slot = self._group_slot(key)
        if slot is None:
            ent = {"ver": ver}
            refs = [weakref.ref(w, lambda _r, key=key: self._drop_group(key)) for w in ws]
            self._grp[key] = (refs, ent)
            # A weight's planes are written by the optimizer in exactly ONE cache entry -- the first fused group that registered
            # it.  Any other entry holding the same weight (another grouping of it, its single entry) is never marked current
            # by an update: it goes stale with the version bump and is re-packed by launches (inside a captured step too), as
            # in round 4.  (k | v of one local layer for the text->vision direction vs k | v of all four layers: two groups.)
            row0 = 0
            for w, r in zip(ws, refs):
                old = self._member.get(id(w))
                single = self._slot_of(w)
                if (old is None or old[0]() is not w or self._group_slot(old[1]) is None) and \
                        (single is None or single[1].get("keep") is None):
                    self._member[id(w)] = (r, key, row0)
                row0 += int(w.shape[0])
        else:
            ent = slot[1]
            if ent["ver"] != ver:
                keep = ent.get("keep")
                ent.clear()
                ent["ver"] = ver
                if keep is not None:
                    ent["keep"] = keep

        def get(transposed, ent=ent, ws=ws):
            if transposed not in ent:
                cat = torch.cat([w.detach() for w in ws], 0)
                ent[False], ent[True] = self._pack_both(cat, ent.get("keep"))
                ent["keep"] = (ent[False], ent[True])
            return ent[transposed]
        return get

    def _member_of(self, w):
        m = self._member.get(id(w))
        if m is None or m[0]() is not w:
            return None
        slot = self._group_slot(m[1])
        return None if slot is None else (slot, m[1], m[2])

    # ---- optimizer-owned planes (csrc/optim.hip adamw_tiles): the fused AdamW launch writes the updated weight's planes in
    # both orientations itself, so no pack launch follows an optimizer step
    def optimizer_planes(self, w):
        """(hi, lo, thi, tlo, N, K, ldw, ldt) of a trained weight whose planes exist in both orientations, or None."""
        if not OPT_PLANES or w.dim() != 2 or w.shape[0] % 64 or w.shape[1] % 64 or w.dtype != torch.float32 or not w.is_contiguous():
            return None
        mem = self._member_of(w)
        if mem is not None:
            (refs, ent), key, row0 = mem
            keep = ent.get("keep")
            if keep is not None and row0 % 64 == 0:
                pf, pt = keep
                N, K = w.shape
                if pf.Kp == K and pf.hi.shape[1] == K and pt.hi.shape[0] == K:
                    return (pf.hi[row0:row0 + N], pf.lo[row0:row0 + N], pt.hi[:, row0:row0 + N], pt.lo[:, row0:row0 + N], N, K, K,
                            int(pt.hi.shape[1]))
        slot = self._slot_of(w)
        if slot is None:
            return None
        keep = slot[1].get("keep")
        if keep is None or w.dim() != 2 or w.shape[0] % 64 or w.shape[1] % 64 or w.dtype != torch.float32 or not w.is_contiguous():
            return None
        pf, pt = keep
        N, K = w.shape
        if pf.Kp != K or pt.Kp != N or tuple(pf.hi.shape) != (N, K) or tuple(pt.hi.shape) != (K, N):
            return None
        return pf.hi, pf.lo, pt.hi, pt.lo, N, K, K, N

    def mark_written(self, w):
        """The optimizer wrote the planes of `w` for its CURRENT version (after the version bump)."""
        mem = self._member_of(w)
        if mem is not None and mem[0][1].get("keep") is not None:
            (refs, ent), key, row0 = mem
            i = key.index(id(w))
            ver = list(ent["ver"])
            ver[i] = (w._version, w.data_ptr(), tuple(w.shape))
            ent["ver"] = tuple(ver)
            ent[False], ent[True] = ent["keep"]
            return
        slot = self._slot_of(w)
        if slot is None or slot[1].get("keep") is None:
            return
        ent = slot[1]
        ent["ver"] = (w._version, w.data_ptr(), tuple(w.shape))
        ent[False], ent[True] = ent["keep"]

    def ensure_current(self, params):
        """Before the replay of a captured training step (which holds no pack launches): re-pack, in place, every weight
        that changed behind the cache (load_state_dict, an eager update without plane records)."""
        for w in params:
            mem = self._member_of(w)
            if mem is not None and mem[0][1].get("keep") is not None and self.optimizer_planes(w) is not None:
                (refs, ent), key, row0 = mem
                ws = [r() for r in refs]
                if ent["ver"] != tuple((x._version, x.data_ptr(), tuple(x.shape)) for x in ws):
                    self.group_getter(tuple(ws))(False)
                continue
            slot = self._slot_of(w)
            if slot is None or slot[1].get("keep") is None or self.optimizer_planes(w) is None:
                continue                  # (weights without plane records are re-packed by launches INSIDE the captured step)
            if slot[1]["ver"] != (w._version, w.data_ptr(), tuple(w.shape)):
                self.getter(w)(False)

    def clear(self):
        self._ent.clear()
        self._grp.clear()
        self._member.clear()


WEIGHTS = _WeightCache()
SEED_DEV = None        # device int64[1]: per-replay seed word of a captured training step (set by train_graph.py)
OPT_PLANES = bool(int(__import__('os').environ.get('GRIDMM_OPT_PLANES', '1')))   # A/B switch: 0 = re-pack every updated weight (round 4)
SPLITK_OFF = bool(int(__import__('os').environ.get('GRIDMM_SPLITK_OFF', '0')))   # A/B switch for tools/bench_train.py


def _as2d(x):
    x = ops.uniform_rows(x)
    M, K, ld = _rows2d(x)
    return x, M, K, ld


def _gemm(a2d, pw, bias=None, residual=None, planes=False, plane_shift=None):
    """(M,K) fp32 @ packed (N,K)^T -> (M,N) fp32 through ops.linear (planes kernel when shapes allow).  planes: also the
    bf16 hi/lo planes of the result from the same epilogue -> (fp32, hi, lo); plane_shift: see ops.linear."""
    pw.bias = bias
    try:
        act = ops.linear(a2d, pw, residual=residual, allow_tiled=False, want_planes=planes, plane_shift=plane_shift)
        return (act.f32, act.hi, act.lo) if planes else act.f32
    finally:
        pw.bias = None


def transpose_split(x2d, want_colsum=False, want_rows=False, out=None):
    """fp32 (M,C) -> transposed bf16 hi/lo planes (C,Mp), Mp = roundup(M,32) [, column sums (C,)] [, the row-major
    planes as an ops.Act] -- one pass over x2d.  out = (hi, lo, row_hi, row_lo): write into these buffers (the persistent
    planes of a trained weight, _WeightCache)."""
    lib = _lib.load()
    x2d, M, C, ld = _as2d(x2d)
    Mp = (M + 31) // 32 * 32
    if out is not None:
        hi, lo = out[0], out[1]
        assert tuple(hi.shape) == (C, Mp) and hi.is_contiguous() and lo.is_contiguous()
    else:
        hi = torch.empty(C, Mp, dtype=torch.bfloat16, device=x2d.device)
        lo = torch.empty_like(hi)
    cs = torch.empty(C, dtype=torch.float32, device=x2d.device) if want_colsum else None
    cs_ws = torch.empty((Mp + 255) // 256, C, dtype=torch.float32, device=x2d.device) if want_colsum else None
    rows = None
    if want_rows and C % 8 == 0:
        if out is not None:
            rows = ops.Act(x2d, out[2], out[3])
            assert tuple(out[2].shape) == (M, C) and out[2].is_contiguous() and out[3].is_contiguous()
        else:
            rh = torch.empty(M, C, dtype=torch.bfloat16, device=x2d.device)
            rows = ops.Act(x2d, rh, torch.empty_like(rh))
    _lib.check(lib.gridmm_transpose_split(_p(x2d), ld, _p(hi), _p(lo), _p(cs), _p(cs_ws), _p(rows.hi if rows else None),
                                          _p(rows.lo if rows else None), C, M, C, Mp, _stream()),
               "gridmm_transpose_split")
    return hi, lo, cs, Mp, rows


def _param_versions(params):
    """What torch's saved-tensor check would remember of these parameters: the weight planes the layer C calls point at are
    persistent buffers that the optimizer (or a re-pack after load_state_dict) rewrites IN PLACE, so a backward that runs after
    such a rewrite would silently differentiate against the new weights."""
    return tuple((p._version if torch.is_tensor(p) else None) for p in params)


def _check_param_versions(ctx_versions, params, what):
    now = _param_versions(params)
    if now != ctx_versions:
        bad = [i for i, (a, b) in enumerate(zip(ctx_versions, now)) if a != b]
        raise RuntimeError("%s: parameter(s) %s were modified in place between this node's forward and its backward (an optimizer "
                           "step or load_state_dict under a retained graph); the weight planes the backward reads are shared, "
                           "persistent buffers -- run backward before updating the weights" % (what, bad))


# ------------------------------------------------------------------------------------------------
# deferred parameter gradients: ONE accumulation launch per backward instead of one per parameter and use
# ------------------------------------------------------------------------------------------------
class _DeferredGrads:
    """A fine-tuning iteration backpropagates through every navigation step of a rollout at once (agent_base.py:190-199):
    each step gives each parameter a gradient and autograd's AccumulateGrad adds them one launch at a time (~2 000
    launches of ~4 us per iteration at 7 steps).  Inside `deferred_param_grads()` the custom Functions of this module keep
    the gradients of leaf Parameters aside (hand()) and return None for them; `flush_param_grads()` -- call it right
    after backward() -- sums them into .grad with one multi-tensor launch (gridmm_multi_grad_accumulate), in the order they
    were produced: the same sums, bit for bit, as the sequential in-place adds.  Parameters that also receive gradients
    through plain torch ops keep those (the flush adds to an existing .grad).  Post-accumulate-grad hooks do not fire for
    deferred gradients; dist.GradientReducer.reduce() picks such gradients up by itself."""

    def __init__(self):
        self.active = False
        self.pending = {}            # id(param) -> (param, [gradients in production order])
        self.streams = set()         # streams the kept gradients were produced on (autograd runs a node on its forward's stream)

    def hand(self, param, g):
        if g is None or not self.active:
            return g
        if not (isinstance(param, torch.nn.Parameter) and param.is_leaf and g.is_cuda and g.dtype == torch.float32
                and param.dtype == torch.float32 and g.shape == param.shape and g.is_contiguous()):
            return g
        self.pending.setdefault(id(param), (param, []))[1].append(g)
        self.streams.add(torch.cuda.current_stream(g.device))
        return None


DEFERRED = _DeferredGrads()
_SUM_REC = None


@contextlib.contextmanager
def deferred_param_grads():
    prev, DEFERRED.active = DEFERRED.active, True
    try:
        yield
    except BaseException:
        DEFERRED.pending.clear()        # a backward that raised half-way: its gradients must not reach the next flush
        DEFERRED.streams.clear()
        raise
    finally:
        DEFERRED.active = prev


def flush_param_grads():
    """Sum the gradients kept aside since the last flush into their parameters' .grad (see _DeferredGrads)."""
    global _SUM_REC
    import numpy as np
    items = list(DEFERRED.pending.values())
    DEFERRED.pending.clear()
    # deferred gradients bypass AccumulateGrad, hence also autograd's end-of-backward sync of the caller's stream with the
    # streams its nodes ran on: order the summing launch behind every producing stream (a no-op for single-stream loops)
    if DEFERRED.streams:
        cur = torch.cuda.current_stream()
        for st in DEFERRED.streams:
            if st != cur:
                cur.wait_stream(st)
        DEFERRED.streams.clear()
    work = []
    for prm, gs in items:
        if prm.grad is None:
            prm.grad, gs = gs[0], gs[1:]            # the first gradient becomes .grad as it is (AccumulateGrad does the same)
        elif not (prm.grad.dtype == torch.float32 and prm.grad.is_contiguous()):
            for g in gs:
                prm.grad.add_(g)
            continue
        if gs:
            work.append((prm.grad, gs))
    if not work:
        return
    if _SUM_REC is None:
        _SUM_REC = np.dtype([("dst", "<u8"), ("src", "<u8", (7,)), ("n", "<i8"), ("n_src", "<i4"), ("pad", "<i4")])
        assert _SUM_REC.itemsize == 80
    lib = _lib.load()
    dev = work[0][0].device
    rnd = 0
    while True:                                      # more than 7 gradients of one parameter: further rounds, in order
        part = [(dst, gs[7 * rnd:7 * rnd + 7]) for dst, gs in work if len(gs) > 7 * rnd]
        if not part:
            break
        rec = np.zeros(len(part), _SUM_REC)
        first = np.zeros(len(part) + 1, np.int32)
        for i, (dst, gs) in enumerate(part):
            rec["dst"][i], rec["n"][i], rec["n_src"][i] = dst.data_ptr(), dst.numel(), len(gs)
            for k, g in enumerate(gs):
                rec["src"][i, k] = g.data_ptr()
            first[i + 1] = first[i] + (dst.numel() + 16383) // 16384
        blob = np.concatenate([rec.view(np.uint8).reshape(-1), first.view(np.uint8)])
        d = upload_blob(blob, dev)                   # (pinned ring: no pinned allocation per call)
        _lib.check(lib.gridmm_multi_grad_accumulate(_p(d), ctypes.c_void_p(d.data_ptr() + rec.nbytes), len(part), int(first[-1]),
                                                    _stream()), "gridmm_multi_grad_accumulate")
        rnd += 1


TN_GEMM = bool(int(__import__('os').environ.get('GRIDMM_TN_GEMM', '1')))   # A/B switch: 0 = transposed planes + NT GEMM (round 1-3)


# Producers (LayerNorm, GELU / ReLU, attention) also emit the bf16 planes of their output and hang them on the tensor; the
# Linear that consumes the tensor takes them as its A operand and keeps them for its weight gradient: no split pass over an
# activation that a kernel of ours just wrote.  The tag carries the tensor's version: an in-place edit drops it.
def _tag_planes(y, hi, lo):
    y._gridmm_planes = (hi, lo, y._version)
    return y


def _take_planes(x, M, K):
    """Planes a producer hung on x, as the A operand of a Linear -- never SHIFTED ones (a k | v or q | k | v projection tagged
    by linear(..., out_planes=<int c0>): its planes hold x - x[row 0 of the episode] for the columns >= c0, which only the
    attention kernels may read; a Linear over such a tensor splits x itself)."""
    p = getattr(x, "_gridmm_planes", None)
    if p is None or p[2] != x._version or not x.is_contiguous() or p[0].numel() != M * K or p[0].shape[-1] != K:
        return None
    if getattr(x, "_gridmm_shift", None) is not None:
        return None
    return p[0].view(M, K), p[1].view(M, K)


def _want_planes(width):
    return TN_GEMM and width % 8 == 0


def split_rows_pad(x2d, want_colsum=False):
    """fp32 (M,C) -> row-major bf16 hi/lo planes (Mp,C) with rows [M,Mp) zero, Mp = roundup(M,32) [, column sums (C,)]: one
    pass per activation / gradient for both of its GEMM roles (forward / dX: the first M rows; dW: gridmm_linear_planes_tn).
    (The bias gradient of a Linear with a weight gradient comes from the weight-gradient GEMM: _gemm_tn_rows(want_db=True).)"""
    lib = _lib.load()
    x2d, M, C, ld = _as2d(x2d)
    Mp = (M + 31) // 32 * 32
    hi = torch.empty(Mp, C, dtype=torch.bfloat16, device=x2d.device)
    lo = torch.empty_like(hi)
    cs = torch.empty(C, dtype=torch.float32, device=x2d.device) if want_colsum else None
    cs_ws = torch.empty((Mp + 255) // 256, C, dtype=torch.float32, device=x2d.device) if want_colsum else None
    _lib.check(lib.gridmm_split_rows_pad(_p(x2d), ld, _p(hi), _p(lo), C, _p(cs), _p(cs_ws), M, C, Mp, _stream()),
               "gridmm_split_rows_pad")
    return hi, lo, cs, Mp, ops.Act(x2d, hi[:M], lo[:M])


def _gemm_tn_rows(yp, xp, N, K, M, want_db=False):
    """dW (N,K) = dY^T X from the ROW planes yp = (hi, lo) (>= M rows, N) of dY and xp (>= M rows, K) of X.
    want_db: also db (N,) = the column sums of dY, computed by the GEMM itself from the planes (gridmm_linear_planes_tn_db)."""
    lib = _lib.load()
    splits = 1 if SPLITK_OFF else lib.gridmm_linear_planes_tn_splits(M, N, K)
    dev = yp[0].device
    dw = torch.empty(N, K, dtype=torch.float32, device=dev)
    ws = torch.empty(splits, N, K, dtype=torch.float32, device=dev) if splits > 1 else None
    if not want_db:
        _lib.check(lib.gridmm_linear_planes_tn(_p(yp[0]), _p(yp[1]), N, _p(xp[0]), _p(xp[1]), K, _p(dw), _p(ws), M, N, K, splits,
                                               _stream()), "gridmm_linear_planes_tn")
        return dw
    db = torch.empty(N, dtype=torch.float32, device=dev)
    db_ws = torch.empty(splits, N, dtype=torch.float32, device=dev)
    _lib.check(lib.gridmm_linear_planes_tn_db(_p(yp[0]), _p(yp[1]), N, _p(xp[0]), _p(xp[1]), K, _p(dw), _p(ws), M, N, K, splits,
                                              _p(db_ws), _p(db), _stream()), "gridmm_linear_planes_tn_db")
    return dw, db


def _gemm_tn(yt, xt, N, K, M, Mp, dy2d):
    """dW (N,K) = dY^T X from the transposed planes yt = (hi, lo) of dY and xt of X (contraction over the M rows)."""
    pw = ops.PackedLinear.__new__(ops.PackedLinear)
    pw.hi, pw.lo, pw.bias, pw.N, pw.K, pw.Kp = xt[0], xt[1], None, K, Mp, Mp
    if K % 4 == 0:
        # few output tiles, long contraction: split the M rows over enough workgroups to fill the chip
        tiles = -(-N // 128) * -(-K // 128)
        # measured (profiles/README.md, split-K): only the 768x768 outputs gain (70 -> 47 us at 6912 rows, 52 -> 20 us at 1824)
        splits = max(1, min(8, 288 // tiles, Mp // 256)) if tiles <= 36 else 1
        if SPLITK_OFF:
            splits = 1
        if splits > 1:
            lib = _lib.load()
            dw = torch.empty(N, K, dtype=torch.float32, device=dy2d.device)
            ws = torch.empty(splits, N, K, dtype=torch.float32, device=dy2d.device)
            _lib.check(lib.gridmm_linear_planes_splitk(_p(yt[0]), _p(yt[1]), Mp, _p(xt[0]), _p(xt[1]), Mp, _p(dw), _p(ws),
                                                       N, K, Mp, splits, _stream()), "gridmm_linear_planes_splitk")
            return dw
        return ops.linear(ops.Act(None, yt[0], yt[1]), pw).f32
    # K = 5 / 7 / 14 position-feature layers: fp32-A kernel (any N); A = dY^T zero-padded to Mp columns
    a = torch.zeros(N, Mp, dtype=torch.float32, device=dy2d.device)
    a[:, :M] = dy2d.t()
    return ops.linear(a, pw).f32


def _linear_fwd(ctx, x, wshape, w_req, bias, residual, packs, out_planes):
    """Forward of y = x W^T + b (+ residual) for a weight -- or a row-wise concatenation of weights -- of shape wshape whose
    planes come from packs(transposed).  Each activation / gradient is read once per role pair: the forward's input split
    emits the row planes that are also an operand of dW (saved instead of the fp32 input)."""
    N = wshape[0]
    K = x.shape[-1]
    ctx.packs, ctx.wshape = packs, tuple(wshape)
    x2 = x.float().contiguous().view(-1, K)
    r2 = None if residual is None else residual.float().contiguous().view(-1, N)
    need_w = w_req or (bias is not None and bias.requires_grad)
    xt = None
    a = x2
    ctx.tn = False
    if need_w and K % 8 == 0 and TN_GEMM and N % 8 == 0:
        pl = _take_planes(x, x2.shape[0], K) if x.dtype == torch.float32 else None
        if pl is not None:                              # the producer of x wrote its planes already
            xh, xl = pl
            Mp, a = x2.shape[0], ops.Act(x2, xh, xl)
        else:
            xh, xl, _, Mp, a = split_rows_pad(x2)       # row planes: A of this GEMM, operand of dW = dY^T X
        xt, ctx.tn = (xh, xl, Mp), True
    elif need_w and K % 8 == 0:
        xh, xl, _, Mp, rows = transpose_split(x2, want_rows=True)
        xt, a = (xh, xl, Mp), rows
    want_out = out_planes is not False and out_planes is not None and N % 8 == 0 and K % 32 == 0
    # out_planes = an int c0 (not True): the planes of the columns >= c0 are SHIFTED by row 0 of their episode (k | v
    # projections; csrc/attention_train.hip "SHIFTED K / V").  The shift table = the projection of row 0 of every episode:
    # one B-row GEMM over the A planes through the batched row map.
    shift = None
    bvec = None if bias is None else bias.detach().float()
    if want_out and out_planes is not True and x.dim() == 3 and isinstance(a, ops.Act) and a.hi is not None and r2 is None:
        B_, S_ = x.shape[0], x.shape[1]
        pwf = packs(False)
        pwf.bias = bvec
        try:
            shift = ops.linear(ops.Act(None, a.hi[:B_ * S_].view(B_, S_, K)[:, :1], a.lo[:B_ * S_].view(B_, S_, K)[:, :1]), pwf,
                               allow_tiled=False).f32.view(B_, N)
        finally:
            pwf.bias = None
    y = _gemm(a, packs(False), bvec, r2, planes=want_out,
              plane_shift=None if shift is None else (shift, x.shape[1], int(out_planes)))
    yh = yl = None
    if want_out:
        y, yh, yl = y
    if xt is not None:
        ctx.save_for_backward(xt[0], xt[1])
        ctx.Mp, ctx.saved_t = xt[2], True
    else:
        ctx.save_for_backward(x2)
        ctx.saved_t = False
    ctx.has_bias, ctx.has_res, ctx.M = bias is not None, residual is not None, x2.shape[0]
    y = y.view(*x.shape[:-1], N)
    if yh is not None:          # the consumer (the attention kernels on the bf16 matrix pipe) reads the planes, not y
        _tag_planes(y, yh.view(y.shape), yl.view(y.shape))
        y._gridmm_shift = shift         # (B, N) fp32 or None: row 0 of every episode (the planes >= c0 are relative to it)
    return y


def _linear_bwd(ctx, dy, need_x, need_w):
    """-> dx, dW (fp32, wshape), db: the backward's dY pass emits row planes (dX GEMM, dW GEMM operand) and db."""
    N, K = ctx.wshape
    dy2 = dy.contiguous().view(-1, N)
    M = ctx.M
    dx = dw = db = None
    yh = yl = rows = None
    if need_w and ctx.tn:
        yh, yl, db, Mp, rows = split_rows_pad(dy2)          # (db: by the weight-gradient GEMM, from these planes)
    elif need_w:
        yh, yl, db, Mp, rows = transpose_split(dy2, want_colsum=ctx.has_bias, want_rows=need_x)
    if need_x:
        dx = _gemm(rows if rows is not None else dy2, ctx.packs(True)).view(*dy.shape[:-1], K)
    if need_w and ctx.tn and ctx.has_bias:      # db: reduced from the split pass's partials by the dW summing pass
        dw, db = _gemm_tn_rows((yh, yl), (ctx.saved_tensors[0], ctx.saved_tensors[1]), N, K, M, want_db=True)
    elif need_w and ctx.tn:
        dw = _gemm_tn_rows((yh, yl), (ctx.saved_tensors[0], ctx.saved_tensors[1]), N, K, M)
    elif need_w:
        if ctx.saved_t:
            xt = (ctx.saved_tensors[0], ctx.saved_tensors[1])
        else:
            xh, xl, _, _, _ = transpose_split(ctx.saved_tensors[0])
            xt = (xh, xl)
        dw = _gemm_tn((yh, yl), xt, N, K, M, Mp, dy2)
    return dx, dw, db


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, packs, out_planes=False):
        ctx.set_materialize_grads(False)   # a branch the loss does not use passes None: its backward does no work
        ctx.wdtype, ctx.prm = weight.dtype, (weight, bias)
        return _linear_fwd(ctx, x, weight.shape, weight.requires_grad, bias, residual, packs, out_planes)

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 6
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dx, dw, db = _linear_bwd(ctx, dy, ctx.needs_input_grad[0], need_w)
        if dw is not None:
            dw = dw.to(ctx.wdtype)
        hand = DEFERRED.hand
        return dx, hand(ctx.prm[0], dw), hand(ctx.prm[1], db if ctx.has_bias else None), (dy if ctx.has_res else None), None, None


class _LinearGroup(torch.autograd.Function):
    """One GEMM for several Linear modules that share their input (fused q | k | v, k | v of one or of all local layers):
    the weights stay separate Parameters; their planes live row-block by row-block in ONE pair of fused plane buffers
    (_WeightCache.group_getter) that the optimizer keeps current, so neither a torch.cat of the weights nor a pack launch
    runs per step; the backward returns the row blocks of the fused dW / db as the members' gradients."""

    @staticmethod
    def forward(ctx, x, residual, packs, out_planes, nw, *wb):
        ctx.set_materialize_grads(False)
        ws, bs = wb[:nw], wb[nw:]
        ctx.prm = wb
        ctx.rows = [int(w.shape[0]) for w in ws]
        ctx.nb = len(bs)
        bias = torch.cat([b.detach() for b in bs], 0) if bs else None
        if bias is not None:
            bias.requires_grad_(any(b.requires_grad for b in bs))
        return _linear_fwd(ctx, x, (sum(ctx.rows), ws[0].shape[1]), any(w.requires_grad for w in ws), bias, residual, packs,
                           out_planes)

    @staticmethod
    def backward(ctx, dy):
        nw = len(ctx.rows)
        if dy is None:
            return (None,) * (5 + nw + ctx.nb)
        need_w = any(ctx.needs_input_grad[5:5 + nw]) or (ctx.has_bias and any(ctx.needs_input_grad[5 + nw:]))
        dx, dw, db = _linear_bwd(ctx, dy, ctx.needs_input_grad[0], need_w)
        dws = list(dw.split(ctx.rows, 0)) if dw is not None else [None] * nw
        dbs = list(db.split(ctx.rows, 0)) if (db is not None and ctx.nb) else [None] * ctx.nb
        gr = [DEFERRED.hand(prm, g) for prm, g in zip(ctx.prm, dws + dbs)]
        return (dx, (dy if ctx.has_res else None), None, None, None) + tuple(gr)


def linear_group(x, weights, biases, residual=None, out_planes=False):
    """x @ cat(weights)^T + cat(biases): see _LinearGroup."""
    ws = tuple(weights)
    if not WEIGHTS.groupable(ws):
        return linear(x, torch.cat(list(ws), 0), torch.cat(list(biases), 0) if biases else None, residual, out_planes=out_planes)
    return _LinearGroup.apply(x, residual, WEIGHTS.group_getter(ws), out_planes, len(ws), *ws, *tuple(biases or ()))


SKINNY_LINEAR = bool(int(__import__('os').environ.get('GRIDMM_SKINNY_LINEAR', '1')))   # A/B switch


class _LinearSkinny(torch.autograd.Function):
    """nn.Linear with K <= 16 input features (position / angle embeddings, vilmodel.py:454-470, 538-552, 640-655): fp32 FMA
    kernels (gridmm_linear_skinny / _bwd) instead of the tile GEMM's fp32-A fallback."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        N, K = weight.shape
        x2 = x.float().contiguous().view(-1, K)
        w = weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        M = x2.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        _lib.check(lib.gridmm_linear_skinny(_p(x2), K, _p(w), _p(b), _p(y), N, M, N, K, _stream()), "gridmm_linear_skinny")
        ctx.save_for_backward(x2, w)
        ctx.prm, ctx.has_bias, ctx.xshape = (weight, bias), bias is not None, tuple(x.shape)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        lib = _lib.load()
        x2, w = ctx.saved_tensors
        N, K = w.shape
        M = x2.shape[0]
        dy2 = dy.contiguous().view(M, N)
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dw = db = dx = None
        if need_w:
            dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
            db = torch.empty(N, dtype=torch.float32, device=dy.device) if ctx.has_bias else None
            ws = torch.empty(int(lib.gridmm_linear_skinny_bwd_workspace(M, N, K)), dtype=torch.uint8, device=dy.device)
            _lib.check(lib.gridmm_linear_skinny_bwd(_p(dy2), N, _p(x2), K, _p(dw), _p(db), _p(ws), M, N, K, _stream()),
                       "gridmm_linear_skinny_bwd")
            dw = dw.to(ctx.prm[0].dtype)
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, w).view(ctx.xshape)       # (features are inputs in every caller: normally not needed)
        return dx, DEFERRED.hand(ctx.prm[0], dw), DEFERRED.hand(ctx.prm[1], db)


class _RowDot(torch.autograd.Function):
    """nn.Linear(K, 1) (the heads' last layer, vilmodel.py:437-446): gridmm_rowdot / gridmm_rowdot_bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        K = weight.shape[1]
        x2 = x.float().contiguous().view(-1, K)
        w = weight.detach().float().contiguous().view(-1)
        b = None if bias is None else bias.detach().float().contiguous()
        M = x2.shape[0]
        y = torch.empty(M, dtype=torch.float32, device=x.device)
        _lib.check(lib.gridmm_rowdot(_p(x2), K, _p(w), _p(b), _p(y), M, K, _stream()), "gridmm_rowdot")
        ctx.save_for_backward(x2, w)
        ctx.prm, ctx.has_bias, ctx.xshape = (weight, bias), bias is not None, tuple(x.shape)
        return y.view(*x.shape[:-1], 1)

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        lib = _lib.load()
        x2, w = ctx.saved_tensors
        M, K = x2.shape
        dy2 = dy.contiguous().view(M)
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dx = torch.empty(M, K, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[0] else None
        dw = torch.empty(1, K, dtype=torch.float32, device=dy.device) if need_w else None
        db = torch.empty(1, dtype=torch.float32, device=dy.device) if (need_w and ctx.has_bias) else None
        ws = torch.empty(int(lib.gridmm_rowdot_bwd_workspace(M, K)), dtype=torch.uint8, device=dy.device)
        _lib.check(lib.gridmm_rowdot_bwd(_p(dy2), _p(x2), K, _p(w), _p(dx), K, _p(dw), _p(db), _p(ws), M, K, _stream()),
                   "gridmm_rowdot_bwd")
        if dw is not None:
            dw = dw.to(ctx.prm[0].dtype)
        return (None if dx is None else dx.view(ctx.xshape)), DEFERRED.hand(ctx.prm[0], dw), DEFERRED.hand(ctx.prm[1], db)


def linear(x, weight, bias=None, residual=None, out_planes=False):
    """x (..., K) @ weight (N, K)^T + bias (+ residual).  out_planes: the GEMM epilogue also writes the bf16 hi/lo planes of
    the result and hangs them on the returned tensor (q / k / v projections: the attention kernels take planes).  True: plain
    planes (a q projection); an int c0: the planes of the columns >= c0 are shifted by row 0 of their episode (H for a fused
    q | k | v projection, 0 for a k | v projection) and the tensor also carries that row (`_gridmm_shift`)."""
    if SKINNY_LINEAR and weight.shape[1] <= 16 and weight.shape[0] % 4 == 0 and residual is None and not out_planes and x.is_cuda \
            and weight.dtype == torch.float32:
        return _LinearSkinny.apply(x, weight, bias)
    if SKINNY_LINEAR and weight.shape[0] == 1 and weight.shape[1] % 4 == 0 and residual is None and not out_planes and x.is_cuda \
            and weight.dtype == torch.float32:
        return _RowDot.apply(x, weight, bias)
    return _Linear.apply(x, weight, bias, residual, WEIGHTS.getter(weight), out_planes)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps):
        ctx.set_materialize_grads(False)   # a branch the loss does not use passes None: its backward does no work
        x2 = ops.uniform_rows(x.float())
        r2 = None if residual is None else ops.uniform_rows(residual.float())
        act = ops.layernorm(x2, gamma.detach(), beta.detach(), eps, residual=r2, want_planes=_want_planes(x2.shape[-1]))
        y = act.f32
        ctx.save_for_backward(x2, r2, gamma)
        ctx.eps, ctx.has_res, ctx.prm = eps, residual is not None, (gamma, beta)
        return _tag_planes(y, act.hi, act.lo) if act.hi is not None else y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 5
        lib = _lib.load()
        x2, r2, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        M, H, ldx = _rows2d(x2)
        dx = torch.empty(x2.shape, dtype=torch.float32, device=dy.device)
        dg = torch.empty(H, dtype=torch.float32, device=dy.device)
        db = torch.empty_like(dg)
        ws = torch.empty((M + 3) // 4 * 2 * H, dtype=torch.float32, device=dy.device)
        _lib.check(lib.gridmm_layernorm_bwd(_p(x2), ldx, _p(r2), _rows2d(r2)[2] if r2 is not None else 0,
                                            _p(gamma.detach()), float(ctx.eps), _p(dy), H, _p(dx), H, _p(dg), _p(db),
                                            _p(ws), M, H, _stream()), "gridmm_layernorm_bwd")
        return dx, (dx if ctx.has_res else None), DEFERRED.hand(ctx.prm[0], dg), DEFERRED.hand(ctx.prm[1], db), None


class _LayerNormDropout(torch.autograd.Function):
    """LN(dropout(x) + residual) in one launch each way (gridmm_layernorm_dropout / _bwd): the hidden-state dropout of
    BertSelfOutput / BertOutput rides on the LayerNorm kernels; same mask as gridmm_dropout on the contiguous tensor."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps, p):
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        H = x.shape[-1]
        x2 = x.float().contiguous()
        r2 = None if residual is None else ops.uniform_rows(residual.float())
        M = x2.numel() // H
        seed = hs.host(lambda: int(torch.randint(0, 2 ** 62, (1,)).item()))
        seed_dev = SEED_DEV if hs.MODE is not None else None          # captured steps: the per-replay seed word
        y = torch.empty_like(x2)
        hi = lo = None
        if _want_planes(H):
            hi, lo = ops._planes_like(x2.shape, x2.device)
        _lib.check(lib.gridmm_layernorm_dropout_planes(_p(x2), _p(r2), _rows2d(r2)[2] if r2 is not None else 0,
                                                       _p(gamma.detach()), _p(beta.detach()), float(eps), _p(y), _p(hi), _p(lo),
                                                       float(p), seed, _p(seed_dev), M, H, _stream()),
                   "gridmm_layernorm_dropout")
        ctx.save_for_backward(x2, r2, gamma)
        ctx.eps, ctx.has_res, ctx.p, ctx.seed, ctx.seed_dev = eps, residual is not None, float(p), seed, seed_dev
        ctx.prm = (gamma, beta)
        return _tag_planes(y, hi, lo) if hi is not None else y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 6
        lib = _lib.load()
        x2, r2, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        H = x2.shape[-1]
        M = x2.numel() // H
        dx = torch.empty_like(x2)
        dr = torch.empty_like(x2) if ctx.has_res else None
        dg = torch.empty(H, dtype=torch.float32, device=dy.device)
        db = torch.empty_like(dg)
        ws = torch.empty((M + 3) // 4 * 2 * H, dtype=torch.float32, device=dy.device)
        _lib.check(lib.gridmm_layernorm_dropout_bwd(_p(x2), _p(r2), _rows2d(r2)[2] if r2 is not None else 0,
                                                    _p(gamma.detach()), float(ctx.eps), _p(dy), _p(dx), _p(dr), _p(dg), _p(db),
                                                    _p(ws), ctx.p, ctx.seed, _p(ctx.seed_dev), M, H, _stream()),
                   "gridmm_layernorm_dropout_bwd")
        return dx, dr, DEFERRED.hand(ctx.prm[0], dg), DEFERRED.hand(ctx.prm[1], db), None, None


def layer_norm(x, mod, residual=None, dropout_p=0.0):
    """mod: nn.LayerNorm-like (weight, bias, eps).  LN(x (+ residual)); dropout_p > 0: LN(dropout(x) (+ residual))."""
    if dropout_p > 0 and x.is_cuda and x.shape[-1] % 4 == 0:
        return _LayerNormDropout.apply(x, residual, mod.weight, mod.bias, mod.eps, float(dropout_p))
    if dropout_p > 0:
        x = dropout(x, dropout_p)          # widths the fused kernel does not take: same semantics in two launches
    return _LayerNorm.apply(x, residual, mod.weight, mod.bias, mod.eps)


class _Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        ctx.set_materialize_grads(False)   # a branch the loss does not use passes None: its backward does no work
        lib = _lib.load()
        x = x.contiguous()
        y = torch.empty_like(x)
        hi = lo = None
        if x.dim() >= 2 and _want_planes(x.shape[-1]):
            hi, lo = ops._planes_like(x.shape, x.device)
        _lib.check(lib.gridmm_activation_planes(_p(x), None, _p(y), _p(hi), _p(lo), x.numel(), mode, _stream()),
                   "gridmm_activation")
        ctx.save_for_backward(x)
        ctx.mode = mode
        return _tag_planes(y, hi, lo) if hi is not None else y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 2
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(lib.gridmm_activation(_p(x), _p(dy), _p(dx), x.numel(), ctx.mode + 1, _stream()),
                   "gridmm_activation")
        return dx, None


def gelu(x):
    return _Activation.apply(x, 0)


def relu(x):
    return _Activation.apply(x, 2)


BF16_ATTENTION = bool(int(__import__('os').environ.get('GRIDMM_TRAIN_ATTENTION_BF16', '1')))   # A/B switch: 0 = exact-fp32 MFMA kernels


def _planes_of(t):
    """(hi, lo) planes a producer hung on `t` (shape of t, contiguous), or None."""
    p = getattr(t, "_gridmm_planes", None)
    if p is None or p[2] != t._version or not t.is_contiguous() or tuple(p[0].shape) != tuple(t.shape):
        return None
    return p[0], p[1]


def _planes_strided(t):
    """(hi, lo) plane views with the shape AND strides of `t` (a column block of a tagged projection, see tag_plane_views),
    or None."""
    p = getattr(t, "_gridmm_planes", None)
    if p is None or p[2] != t._version or tuple(p[0].shape) != tuple(t.shape) or p[0].stride() != t.stride() or \
            p[1].stride() != t.stride():
        return None
    return p[0], p[1]


def split_with_planes(t, width):
    """t.split(width, dim=-1) whose pieces carry the matching column blocks of t's planes (the [k | v] blocks the layers of
    the local encoder read out of ONE shared context projection, map_nav_src/models/vilmodel.py:843-853)."""
    parts = t.split(width, dim=-1)
    p = _planes_of(t)
    if p is not None:
        sh = getattr(t, "_gridmm_shift", None)
        shs = sh.split(width, dim=-1) if sh is not None else [None] * len(parts)
        for part, hi, lo, s_ in zip(parts, p[0].split(width, dim=-1), p[1].split(width, dim=-1), shs):
            _tag_planes(part, hi, lo)
            part._gridmm_shift = s_
    return parts


class _Attention(torch.autograd.Function):
    """q_src (B,Sq,nq*H) with q at column q_col; kv_src (B,Sk,nk*H) with k at k_col, v at v_col (fused projection
    outputs are consumed in place through strides).  Returns (B,Sq,H).  When the projections carry their bf16 planes
    (linear(..., out_planes=True)) forward and backward run on the bf16 matrix pipe (gridmm_attention_rows_train / _bwd:
    3-term split, K / V -- Q / dO in the backward -- staged in LDS); otherwise on the exact-fp32 kernels."""

    @staticmethod
    def forward(ctx, q_src, kv_src, kmask, cols, heads, dropout_p=0.0):
        ctx.set_materialize_grads(False)   # a branch the loss does not use passes None: its backward does no work
        lib = _lib.load()
        H = heads * 64
        same = kv_src is None
        if same:
            kv_src = q_src
        def planes(t):     # planes with t's strides (a column block of a shared projection is read in place)
            p = _planes_strided(t) if BF16_ATTENTION else None
            ok = p is not None and t.dim() == 3 and t.stride(2) == 1 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0
            return p if ok else None
        qp = planes(q_src)
        kp = qp if same else planes(kv_src)
        fast = qp is not None and kp is not None and kv_src.shape[1] <= 2048
        def usable(t):     # the kernels take batch / row strides: a column block of a wider projection is read in place
            return t.dtype == torch.float32 and t.stride(-1) == 1 and t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0 \
                and t.data_ptr() % 16 == 0
        if not fast:
            q_src = q_src if usable(q_src) else q_src.contiguous()
            kv_src = q_src if same else (kv_src if usable(kv_src) else kv_src.contiguous())
        qc, kc, vc = cols
        B, Sq = q_src.shape[:2]
        Sk = kv_src.shape[1]
        if kmask is not None:
            kmask = kmask.contiguous()
            kmask = kmask.view(torch.uint8) if kmask.dtype == torch.bool else kmask.to(torch.uint8)
        Sqp = (Sq + 15) // 16 * 16
        out = torch.empty(B, Sq, H, dtype=torch.float32, device=q_src.device)
        lse = torch.empty(B, heads, Sqp, dtype=torch.float32, device=q_src.device)
        scale = 1.0 / math.sqrt(64.0)
        # one 63-bit seed per call from torch's CPU generator (reproducible under torch.manual_seed); the kernels
        # derive the keep-mask of element (b,h,q,k) from it, forward and backward alike
        seed = hs.host(lambda: int(torch.randint(0, 2 ** 62, (1,)).item())) if dropout_p > 0 else 0
        # captured steps (train_graph.py): the kernel arguments are frozen in the graph, so the part of the seed that
        # changes from replay to replay is a device word the kernels read (SEED_DEV, bumped before every replay)
        seed_dev = SEED_DEV if (dropout_p > 0 and hs.MODE is not None) else None
        hi = lo = None
        if TN_GEMM:
            hi, lo = ops._planes_like(out.shape, out.device)     # the output projection's A operand, straight from the kernel
        mbs = kmask.stride(0) if kmask is not None else 0
        if fast:
            qs, ks = (q_src.stride(0), q_src.stride(1)), (kv_src.stride(0), kv_src.stride(1))
            off = lambda t, c: ctypes.c_void_p(t.data_ptr() + 2 * c)      # noqa: E731  (bf16 planes: 2 bytes per element)
            sh = getattr(kv_src, "_gridmm_shift", None)                  # (B, width) row 0 of every episode: shifted k | v planes
            assert sh is None or (sh.dim() == 2 and sh.stride(1) == 1 and sh.dtype == torch.float32)
            vb = ctypes.c_void_p(sh.data_ptr() + 4 * vc) if sh is not None else ctypes.c_void_p(0)
            vbs = sh.stride(0) if sh is not None else 0
            _lib.check(lib.gridmm_attention_rows_train(
                off(qp[0], qc), off(qp[1], qc), qs[0], qs[1], off(kp[0], kc), off(kp[1], kc), ks[0], ks[1], off(kp[0], vc),
                off(kp[1], vc), ks[0], ks[1], _p(kmask), mbs, _p(out), Sq * H, H, _p(hi), _p(lo), Sq * H, H, _p(lse), Sqp, vb, vbs,
                B, heads, Sq, Sk, scale, float(dropout_p), seed, _p(seed_dev), _stream()), "gridmm_attention_rows_train")
            ctx.save_for_backward(qp[0], qp[1], kp[0], kp[1], kmask, out, lse, *([sh] if sh is not None else []))
            ctx.q_shape, ctx.kv_shape, ctx.strides = tuple(q_src.shape), tuple(kv_src.shape), (qs, ks)
        else:
            q, k, v = q_src[..., qc:qc + H], kv_src[..., kc:kc + H], kv_src[..., vc:vc + H]
            _lib.check(lib.gridmm_attention_train_planes(
                _p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1), _p(v), v.stride(0), v.stride(1),
                _p(kmask), mbs, _p(out), Sq * H, H, _p(hi), _p(lo), Sq * H, H, _p(lse), Sqp,
                B, heads, Sq, Sk, scale, float(dropout_p), seed, _p(seed_dev), _stream()), "gridmm_attention_train")
            ctx.save_for_backward(q_src, kv_src, kmask, out, lse)
        ctx.fast = fast
        ctx.cols, ctx.heads, ctx.same, ctx.scale = cols, heads, same, scale
        ctx.dropout_p, ctx.seed, ctx.seed_dev = float(dropout_p), seed, seed_dev
        return _tag_planes(out, hi, lo) if hi is not None else out

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * 6
        lib = _lib.load()
        heads, H = ctx.heads, ctx.heads * 64
        qc, kc, vc = ctx.cols
        dout = dout.contiguous()
        # the kernels write every row of the q / k / v column blocks; only columns outside them (the K|V blocks of the
        # other layers in a shared context projection) need the zero fill
        def covered(width, blocks):
            return sorted(blocks) == list(range(0, width, H)) and width % H == 0
        if ctx.fast:
            qh, ql, kh, kl, kmask, out, lse, *shl = ctx.saved_tensors
            sh = shl[0] if shl else None
            vb = ctypes.c_void_p(sh.data_ptr() + 4 * vc) if sh is not None else ctypes.c_void_p(0)
            vbs = sh.stride(0) if sh is not None else 0
            q_shape, kv_shape = ctx.q_shape, ctx.kv_shape
            B, Sq, Wq = q_shape
            Sk, Wk = kv_shape[1], kv_shape[2]
            Sqp = lse.shape[2]
            dev = dout.device
            mk = lambda shape, full: (torch.empty if full else torch.zeros)(shape, dtype=torch.float32, device=dev)   # noqa: E731
            if ctx.same:
                dq_src = mk(q_shape, covered(Wq, [qc, kc, vc]))
                dkv_src = dq_src
            else:
                dq_src = mk(q_shape, covered(Wq, [qc]))
                dkv_src = mk(kv_shape, covered(Wk, [kc, vc]))
            need = lib.gridmm_attention_rows_bwd_workspace(B, heads, Sq)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            off = lambda t, c: ctypes.c_void_p(t.data_ptr() + 2 * c)      # noqa: E731
            foff = lambda t, c: ctypes.c_void_p(t.data_ptr() + 4 * c)     # noqa: E731
            qs, ks = ctx.strides
            _lib.check(lib.gridmm_attention_rows_bwd(
                off(qh, qc), off(ql, qc), qs[0], qs[1], off(kh, kc), off(kl, kc), ks[0], ks[1], off(kh, vc), off(kl, vc), ks[0], ks[1],
                _p(kmask), kmask.stride(0) if kmask is not None else 0, _p(out), Sq * H, H, _p(dout), Sq * H, H, _p(lse), vb, vbs,
                _p(ws), need, foff(dq_src, qc), Sq * Wq, Wq, foff(dkv_src, kc), Sk * Wk, Wk, foff(dkv_src, vc), Sk * Wk, Wk, B, heads, Sq, Sk,
                Sqp, ctx.scale, ctx.dropout_p, ctx.seed, _p(ctx.seed_dev), _stream()), "gridmm_attention_rows_bwd")
            return dq_src, (None if ctx.same else dkv_src), None, None, None, None
        q_src, kv_src, kmask, out, lse = ctx.saved_tensors
        B, Sq = q_src.shape[:2]
        Sk = kv_src.shape[1]
        Sqp = lse.shape[2]
        cf = dict(memory_format=torch.contiguous_format)     # (the sources may be strided column blocks: dense gradients)
        if ctx.same:
            dq_src = (torch.empty_like if covered(q_src.shape[-1], [qc, kc, vc]) else torch.zeros_like)(q_src, **cf)
            dkv_src = dq_src
        else:
            dq_src = (torch.empty_like if covered(q_src.shape[-1], [qc]) else torch.zeros_like)(q_src, **cf)
            dkv_src = (torch.empty_like if covered(kv_src.shape[-1], [kc, vc]) else torch.zeros_like)(kv_src, **cf)
        delta = torch.empty_like(lse)
        q, k, v = q_src[..., qc:qc + H], kv_src[..., kc:kc + H], kv_src[..., vc:vc + H]
        dq, dk, dv = dq_src[..., qc:qc + H], dkv_src[..., kc:kc + H], dkv_src[..., vc:vc + H]
        _lib.check(lib.gridmm_attention_bwd(
            _p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1), _p(v), v.stride(0), v.stride(1),
            _p(kmask), kmask.stride(0) if kmask is not None else 0, _p(out), Sq * H, H, _p(dout), Sq * H, H, _p(lse),
            _p(delta), _p(dq), dq.stride(0), dq.stride(1), _p(dk), dk.stride(0), dk.stride(1), _p(dv), dv.stride(0),
            dv.stride(1), B, heads, Sq, Sk, Sqp, ctx.scale, ctx.dropout_p, ctx.seed, _p(ctx.seed_dev), _stream()),
                   "gridmm_attention_bwd")
        return dq_src, (None if ctx.same else dkv_src), None, None, None, None


def self_attention(qkv, kmask, heads, dropout_p=0.0):
    """qkv (B,S,3H) = fused [q | k | v] projection.  dropout_p: dropout on the attention probabilities."""
    H = heads * 64
    return _Attention.apply(qkv, None, kmask, (0, H, 2 * H), heads, dropout_p)


def cross_attention(q, kv, kmask, heads, kv_col=0, dropout_p=0.0):
    """q (B,Sq,H); kv (B,Sk,n*2H) with [k | v] of this layer at column kv_col."""
    H = heads * 64
    return _Attention.apply(q, kv, kmask, (0, kv_col, kv_col + H), heads, dropout_p)


def attention_dropout_mask(seed, B, heads, Sq, Sk, p):
    """The keep-mask the kernels derive from `seed` (host restatement of csrc/common.h dropout_keep; tests only)."""
    import numpy as np
    M = np.uint64(0xFFFFFFFF)

    def h32(x):
        x = x & M
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x85ebca6b)) & M
        x ^= x >> np.uint64(13); x = (x * np.uint64(0xc2b2ae35)) & M
        x ^= x >> np.uint64(16)
        return x & M
    idx = np.arange(B * heads * Sq * Sk, dtype=np.uint64) & M
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    x = h32(((idx * np.uint64(0x9E3779B1)) & M) ^ lo) ^ hi
    u = (h32(x) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).reshape(B, heads, Sq, Sk)


class _GridAggregate(torch.autograd.Function):
    """cells (B,196,D) = per-cell softmax(max_l <x_j, text_l>)-weighted sum of the fp16 slab rows."""

    @staticmethod
    def forward(ctx, text_fts, slab, perm, cell_start):
        ctx.set_materialize_grads(False)   # a branch the loss does not use passes None: its backward does no work
        text_fts = text_fts.contiguous()
        B, L, D = text_fts.shape
        frag = ops.text_fragments(text_fts)
        cells, occ, rel, amax = ops.grid_aggregate(slab, perm, cell_start, frag, L, want_relevance=True, want_amax=True)
        ctx.amax = amax                      # routing of the backward (None: generic kernel, the backward recomputes it)
        # The grid memory re-bins its whole history in place every step: keep THIS step's point order.  The slab
        # is append-only within a rollout (rows this step's perm refers to are never rewritten; a training rollout
        # gets a fresh slab, GridMemoryBatch.reset), so it is referenced, not copied -- and kept out of
        # save_for_backward, whose version check would trip on the later in-place appends.
        ctx.save_for_backward(text_fts, kernel_copy(perm), kernel_copy(cell_start), rel)
        ctx.slab = slab
        # ... which makes "the slab rows are still the ones this step saw" OUR invariant to check: GridMemoryBatch tags
        # its slab with an epoch that advances whenever the rows are recycled in place (reset() of a memory that is not
        # kept for backward); backward refuses to run on a recycled slab instead of returning wrong gradients.
        ctx.epoch_ref = getattr(slab, "_gridmm_epoch", None)
        ctx.epoch = None if ctx.epoch_ref is None else ctx.epoch_ref[0]
        if text_fts.requires_grad and ctx.epoch_ref is not None:
            slab._gridmm_in_graph = True     # reset() then allocates a fresh slab for the next rollout
        ctx.mark_non_differentiable(occ)
        return cells, occ

    @staticmethod
    def backward(ctx, dcells, _docc):
        if dcells is None:
            return (None,) * 4
        lib = _lib.load()
        text_fts, perm, cell_start, rel = ctx.saved_tensors
        slab = ctx.slab
        if ctx.epoch_ref is not None and ctx.epoch_ref[0] != ctx.epoch:
            raise RuntimeError("grid_aggregate backward: the grid memory's feature slab was recycled (reset()) after this "
                               "step's forward; set GridMemoryBatch.keep_for_backward = True for training rollouts")
        B, L, D = text_fts.shape
        cap = slab.shape[1]
        dcells = dcells.contiguous()
        dtext = torch.empty_like(text_fts)
        da = torch.empty(B, cap, dtype=torch.float32, device=slab.device)
        if ctx.amax is not None:
            dw = torch.empty(B, cap, dtype=torch.float32, device=slab.device)
            part = torch.empty(lib.gridmm_grid_aggregate_bwd_workspace(B, D, L) // 4, dtype=torch.float32, device=slab.device)
            _lib.check(lib.gridmm_grid_aggregate_bwd_routed(_p(slab), _p(perm), _p(cell_start), _p(rel), _p(ctx.amax),
                                                            _p(dcells), _p(dtext), _p(da), _p(dw), _p(part), B, cap, D, L,
                                                            _stream()), "gridmm_grid_aggregate_bwd_routed")
            return dtext, None, None, None
        am = torch.empty(B, cap, dtype=torch.int32, device=slab.device)
        _lib.check(lib.gridmm_grid_aggregate_bwd(_p(slab), _p(perm), _p(cell_start), _p(rel), _p(text_fts),
                                                 _p(dcells), _p(dtext), _p(da), _p(am), B, cap, D, L, _stream()),
                   "gridmm_grid_aggregate_bwd")
        return dtext, None, None, None


def grid_aggregate(text_fts, slab, perm, cell_start):
    return _GridAggregate.apply(text_fts, slab, perm, cell_start)


# ---- row-wise training stages on the library's kernels (csrc/train_rowops.hip) ---------------------------------------
class _Dropout(torch.autograd.Function):
    """Hidden-state dropout (vilmodel.py:86,166,205; transformer.py dropout / dropout1 / dropout2): counter-based keep-mask
    from one seed per call (torch's CPU generator: reproducible under torch.manual_seed); the backward re-applies it."""

    @staticmethod
    def forward(ctx, x, p):
        ctx.set_materialize_grads(False)   # a branch the loss does not use passes None: its backward does no work
        lib = _lib.load()
        x = x.contiguous()
        seed = hs.host(lambda: int(torch.randint(0, 2 ** 62, (1,)).item()))
        seed_dev = SEED_DEV if hs.MODE is not None else None          # captured steps: the per-replay seed word
        y = torch.empty_like(x)
        _lib.check(lib.gridmm_dropout(_p(x), _p(y), x.numel(), float(p), seed, _p(seed_dev), _stream()), "gridmm_dropout")
        ctx.p, ctx.seed, ctx.seed_dev = float(p), seed, seed_dev
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 2
        lib = _lib.load()
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.check(lib.gridmm_dropout(_p(dy), _p(dx), dy.numel(), ctx.p, ctx.seed, _p(ctx.seed_dev), _stream()),
                   "gridmm_dropout")
        return dx, None


# ------------------------------------------------------------------------------------------------
# kernel-only forms of the few torch operations that become MEMCPY / MEMSET nodes when the step is captured
# ------------------------------------------------------------------------------------------------
# A captured training step must consist of kernel nodes only: the runtime's pre-recorded graph packets (ROCm 7.2 default)
# mishandle copy / fill nodes in a large graph -- one queue slot of the replay keeps the packet of an EARLIER dispatch,
# whose kernel arguments have been recycled by then (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION as soon as eager work
# alternates with replays; DESIGN.md section 5, tools/dbg_train_graph_fault.py, tools/find_copy_nodes.py).  torch issues
# such nodes for contiguous device-to-device copies (clone, select_backward), for the semaphores of a large `sum`
# (the gradient of a broadcast add) and inside the sort-based embedding backward.
def kernel_copy(x):
    """x.clone() as an elementwise kernel (a contiguous same-dtype copy_ is a hipMemcpyAsync, i.e. a memcpy node)."""
    if x.dtype.is_floating_point:
        return x * 1
    if x.dtype == torch.bool:
        return x | False
    return x + 0


def _colsum(dy2d):
    """(M, C) -> (C,) as a GEMM with a row of ones (torch's reduction zeroes its semaphores with a memset node)."""
    return torch.mm(torch.ones(1, dy2d.shape[0], dtype=dy2d.dtype, device=dy2d.device), dy2d).view(-1)


class _AddRow(torch.autograd.Function):
    """x + table[row] broadcast over the leading dims of x (the token-type rows of BertEmbeddings / ImageEmbeddings:
    map_nav_src/models/vilmodel.py:72-77, 486-490); backward: dx = dy, dtable[row] = column sums of dy."""

    @staticmethod
    def forward(ctx, x, table, row):
        ctx.row, ctx.shape = int(row), table.shape
        return x + table[int(row)]

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dt = torch.zeros(ctx.shape, dtype=dy.dtype, device=dy.device)
        dt[ctx.row].add_(_colsum(dy.view(-1, dy.shape[-1])))
        return dy, dt, None


def add_row(x, table, row):
    return _AddRow.apply(x, table, row)


class _SmallEmbedding(torch.autograd.Function):
    """table[idx] for a table of a few rows (nav_type_embedding: 3 rows); backward as onehot^T @ dy (deterministic, a
    GEMM) instead of torch's sort-based embedding backward."""

    @staticmethod
    def forward(ctx, idx, table):
        ctx.save_for_backward(idx)
        ctx.n = table.shape[0]
        return table.index_select(0, idx.reshape(-1)).view(*idx.shape, table.shape[1])

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, dy.shape[-1])
        onehot = (idx.reshape(1, -1) == torch.arange(ctx.n, device=idx.device).view(-1, 1)).to(dy2.dtype)
        return None, torch.mm(onehot, dy2)


def small_embedding(idx, table):
    return _SmallEmbedding.apply(idx, table)


def dropout(x, p):
    """x fp32 with numel % 4 == 0 (hidden states: last dim 768)."""
    return _Dropout.apply(x, p) if p > 0 else x


def dropout_mask(seed, n, p):
    """The keep-mask gridmm_dropout derives from `seed` for n elements (host restatement of csrc/common.h; tests only)."""
    import numpy as np
    m32 = np.uint64(0xFFFFFFFF)

    def h32(x):
        x = x & m32
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x85ebca6b)) & m32
        x ^= x >> np.uint64(13); x = (x * np.uint64(0xc2b2ae35)) & m32
        x ^= x >> np.uint64(16)
        return x
    idx = np.arange(n, dtype=np.uint64)
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    x = h32(((idx * np.uint64(0x9E3779B1)) & m32) ^ lo) ^ hi
    return (h32(x) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0) >= np.float32(p)


class _CellsCompact(torch.autograd.Function):
    """x (B,196,H) = grid_proj(cells) + position embedding, occ (B,196) uint8 -> rows compacted to the front in cell order,
    zeros behind (vilmodel.py:813-823); backward: rows scattered back (gridmm_cells_compact_bwd)."""

    @staticmethod
    def forward(ctx, proj, pos, occ):
        ctx.set_materialize_grads(False)   # a branch the loss does not use passes None: its backward does no work
        B, C, H = proj.shape
        out = torch.empty(B, C, H, dtype=torch.float32, device=proj.device)
        mask = torch.empty(B, C, dtype=torch.uint8, device=proj.device)
        from . import ops
        ops.cells_compact(proj.contiguous(), pos.contiguous(), occ, out, mask)
        ctx.save_for_backward(occ)
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, dout, _dmask):
        if dout is None:
            return (None,) * 3
        lib = _lib.load()
        (occ,) = ctx.saved_tensors
        dout = dout.contiguous()
        B, C, H = dout.shape
        d = torch.empty(B, C, H, dtype=torch.float32, device=dout.device)
        _lib.check(lib.gridmm_cells_compact_bwd(_p(dout), C * H, _p(occ), _p(d), B, H, _stream()), "gridmm_cells_compact_bwd")
        return d, d, None


def cells_compact(proj, pos, occ):
    """-> (compacted (B,196,H), key mask (B,196) uint8 with the reference's view quirk)."""
    return _CellsCompact.apply(proj, pos, occ)


class _FuseLogits(torch.autograd.Function):
    """gridmm_fuse_logits / gridmm_fuse_logits_bwd (vilmodel.py:859-899) with the integer index maps."""

    @staticmethod
    def forward(ctx, g_raw, l_raw, grid_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, cand_of_node, cand_visited):
        from . import ops
        u8 = lambda m: (m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)).contiguous()   # noqa: E731
        gm, gv, vn = u8(gmap_masks), u8(gmap_visited), u8(vp_nav_masks)
        g_raw, l_raw, grid_raw = g_raw.contiguous(), l_raw.contiguous(), grid_raw.contiguous()
        fuse_raw = None if fuse_raw is None else fuse_raw.contiguous()
        con, cv = cand_of_node.to(torch.int32).contiguous(), u8(cand_visited)
        outs = ops.fuse_logits(g_raw, l_raw, grid_raw, fuse_raw, gm, gv, vn, con, cv)
        ctx.save_for_backward(g_raw, l_raw, fuse_raw, gm, gv, vn, con, cv)
        ctx.set_materialize_grads(False)      # an output the loss does not use arrives as None, not as zeros
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_global, d_local, d_grid, d_fused):
        lib = _lib.load()
        g_raw, l_raw, fuse_raw, gm, gv, vn, con, cv = ctx.saved_tensors
        B, G = g_raw.shape
        V = l_raw.shape[1]
        c = lambda t: None if t is None else t.contiguous()   # noqa: E731
        d_global, d_local, d_grid, d_fused = c(d_global), c(d_local), c(d_grid), c(d_fused)
        if d_global is None and d_local is None and d_grid is None and d_fused is None:
            return (None,) * 9
        dg, dl, dgr = torch.empty_like(g_raw), torch.empty_like(l_raw), torch.empty_like(g_raw)
        df = None if fuse_raw is None else torch.empty_like(fuse_raw)
        _lib.check(lib.gridmm_fuse_logits_bwd(_p(g_raw), _p(l_raw), _p(fuse_raw), _p(gm), _p(gv), _p(vn), _p(con), _p(cv),
                                              _p(d_global), _p(d_local), _p(d_grid), _p(d_fused), _p(dg), _p(dl), _p(dgr),
                                              _p(df), B, G, V, _stream()), "gridmm_fuse_logits_bwd")
        # a head whose logits the loss does not use gets NO gradient (its parameters keep grad None and the optimizer skips
        # them, weight decay included: grid_sap_head in the fine-tune loss, agent.py:340-347)
        used_gl = d_global is not None or d_fused is not None
        used_l = d_local is not None or d_fused is not None
        return (dg if used_gl else None, dl if used_l else None, dgr if d_grid is not None else None,
                df if (used_gl or used_l) else None, None, None, None, None, None)


def fuse_logits(g_raw, l_raw, grid_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, cand_of_node, cand_visited):
    """-> (global, local, grid, fused) logits, -inf where masked."""
    return _FuseLogits.apply(g_raw, l_raw, grid_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, cand_of_node,
                             cand_visited)


# ------------------------------------------------------------------------------------------------
# one cross-modal layer as ONE autograd node: gridmm_xattn_layer_train_fwd / gridmm_xattn_layer_bwd (csrc/layer_train.hip)
# ------------------------------------------------------------------------------------------------
class _CLinearTrain(ctypes.Structure):
    _fields_ = [("w_hi", ctypes.c_void_p), ("w_lo", ctypes.c_void_p), ("Kp", ctypes.c_int),
                ("wt_hi", ctypes.c_void_p), ("wt_lo", ctypes.c_void_p), ("Np", ctypes.c_int),
                ("bias", ctypes.c_void_p), ("N", ctypes.c_int), ("K", ctypes.c_int)]


class _CLnTrain(ctypes.Structure):
    _fields_ = [("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("eps", ctypes.c_float)]


class _CXLayerTrain(ctypes.Structure):
    _fields_ = [(n, _CLinearTrain) for n in ("xq", "xo", "sqkv", "so", "ffn_i", "ffn_o")] + \
               [(n, _CLnTrain) for n in ("x_ln", "s_ln", "f_ln")] + \
               [("p_hidden", ctypes.c_float), ("p_attn", ctypes.c_float), ("seed", ctypes.c_ulonglong * 5),
                ("seed_dev", ctypes.c_void_p), ("attention_fp32", ctypes.c_int)]


class _CXLayerGrads(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("xq_w", "xq_b", "xo_w", "xo_b", "sqkv_w", "sqkv_b", "so_w", "so_b", "ffn_i_w",
                                                "ffn_i_b", "ffn_o_w", "ffn_o_b", "x_ln_g", "x_ln_b", "s_ln_g", "s_ln_b",
                                                "f_ln_g", "f_ln_b")]


def _ptr(t):
    return None if t is None else t.data_ptr()


class _XLayer(torch.autograd.Function):
    """y = GraphLXRTXLayer(x | kv) (vilmodel.py:399-414): cross attention over the projected context, self attention, feed
    forward -- forward and backward each ONE C call.  params: xq.w, xq.b, xo.w, xo.b, q.w, q.b, k.w, k.b, v.w, v.b, so.w, so.b,
    ffn_i.w, ffn_i.b, ffn_o.w, ffn_o.b, x_ln.w, x_ln.b, s_ln.w, s_ln.b, f_ln.w, f_ln.b (22 tensors)."""

    @staticmethod
    def forward(ctx, x, kv, ctx_mask, self_mask, k_col, heads, p_hidden, p_attn, eps, *params):
        lib = _lib.load()
        H = heads * 64
        B, Sq = x.shape[:2]
        cross = kv is not None                      # kv = None: a BertLayer (self attention + feed forward; bert_layer_fused)
        Sk = kv.shape[1] if cross else 0
        x2 = x.float().contiguous()
        kvp = kvs = None
        if cross:
            kvp = _planes_strided(kv) if BF16_ATTENTION else None     # planes of the context projections (same strides as kv)
            kvs = getattr(kv, "_gridmm_shift", None) if kvp is not None else None   # (B, width): row 0 of every episode (shifted planes)
            if kv.stride(2) != 1 or kv.dtype != torch.float32:
                kv, kvp, kvs = kv.float().contiguous(), None, None
            assert kvs is None or (kvs.dim() == 2 and kvs.stride(1) == 1 and kvs.shape[1] == kv.shape[2])
        (xqw, xqb, xow, xob, qw, qb, kw, kb, vw, vb, sow, sob, fiw, fib, fow, fob, xg, xb, sg, sb, fg, fb) = params
        ctx.prm = params
        qkvb = torch.cat([qb, kb, vb], 0)
        I = fiw.shape[0]
        keep = []                                   # planes / biases / masks the C struct points into

        def lin(w, b):
            if w is None:
                return _CLinearTrain()
            get = WEIGHTS.getter(w)
            pf, pt = get(False), get(True)
            bb = b.detach().float().contiguous()
            keep.extend([pf, pt, bb])
            return _CLinearTrain(pf.hi.data_ptr(), pf.lo.data_ptr(), pf.Kp, pt.hi.data_ptr(), pt.lo.data_ptr(), pt.Kp,
                                 bb.data_ptr(), w.shape[0], w.shape[1])

        def lnp(g, b, e):
            if g is None:
                return _CLnTrain()
            gg, bb = g.detach().float().contiguous(), b.detach().float().contiguous()
            keep.extend([gg, bb])
            return _CLnTrain(gg.data_ptr(), bb.data_ptr(), float(e))
        draw = lambda on: hs.host(lambda: int(torch.randint(0, 2 ** 62, (1,)).item())) if on else 0   # noqa: E731
        seeds = [draw(cross and p_attn > 0), draw(cross and p_hidden > 0), draw(p_attn > 0), draw(p_hidden > 0), draw(p_hidden > 0)]
        seed_dev = SEED_DEV if ((p_attn > 0 or p_hidden > 0) and hs.MODE is not None) else None
        def lin_group(ws, b):           # q | k | v as ONE projection: fused planes of the three Parameters (no torch.cat of weights)
            if WEIGHTS.groupable(tuple(ws)):
                get = WEIGHTS.group_getter(tuple(ws))
            else:
                get = WEIGHTS.getter(torch.cat(list(ws), 0))
            pf, pt = get(False), get(True)
            bb = b.detach().float().contiguous()
            keep.extend([pf, pt, bb])
            return _CLinearTrain(pf.hi.data_ptr(), pf.lo.data_ptr(), pf.Kp, pt.hi.data_ptr(), pt.lo.data_ptr(), pt.Kp,
                                 bb.data_ptr(), sum(int(w.shape[0]) for w in ws), ws[0].shape[1])
        L = _CXLayerTrain(lin(xqw, xqb), lin(xow, xob), lin_group((qw, kw, vw), qkvb), lin(sow, sob), lin(fiw, fib), lin(fow, fob),
                          lnp(xg, xb, eps[0]), lnp(sg, sb, eps[1]), lnp(fg, fb, eps[2]), float(p_hidden), float(p_attn),
                          (ctypes.c_ulonglong * 5)(*seeds), _ptr(seed_dev), 0 if BF16_ATTENTION else 1)

        def u8(m):
            if m is None:
                return None
            m = m.contiguous()
            return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
        cm, sm = u8(ctx_mask), u8(self_mask)
        saved = torch.empty(int(lib.gridmm_xattn_layer_train_saved_bytes(B, Sq, H, I)), dtype=torch.uint8, device=x.device)
        ws = torch.empty(int(lib.gridmm_xattn_layer_train_workspace(B, Sq, H, I)), dtype=torch.uint8, device=x.device)
        y = torch.empty(B, Sq, H, dtype=torch.float32, device=x.device)
        _lib.check(lib.gridmm_xattn_layer_train_fwd(
            ctypes.byref(L), _p(x2), _p(kv), _p(kvp[0] if kvp else None), _p(kvp[1] if kvp else None), _p(kvs),
            kvs.stride(0) if kvs is not None else 0, kv.stride(0) if cross else 0, kv.stride(1) if cross else 0, int(k_col),
            int(k_col) + H, _p(cm),
            cm.stride(0) if cm is not None else 0, _p(sm), sm.stride(0) if sm is not None else 0, _p(y), _p(saved),
            saved.numel(), _p(ws), ws.numel(), B, Sq, Sk, heads, _stream()), "gridmm_xattn_layer_train_fwd")
        ctx.save_for_backward(x2, kv, cm, sm, saved, *(kvp if kvp else ()), *([kvs] if kvs is not None else []))
        ctx.L, ctx.keep, ctx.dims = L, keep, (B, Sq, Sk, H, I, heads, int(k_col))
        ctx.shapes = [tuple(p.shape) if p is not None else None for p in params]
        ctx.versions = _param_versions(params)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        _check_param_versions(ctx.versions, ctx.prm, "x-layer backward")
        x2, kv, cm, sm, saved, *rest = ctx.saved_tensors
        kvp, kvs = rest[:2], (rest[2] if len(rest) > 2 else None)
        B, Sq, Sk, H, I, heads, k_col = ctx.dims
        dev = dy.device
        dy = dy.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        cross = kv is not None
        g = {"sqkv_w": torch.empty(3 * H, H, **f32), "sqkv_b": torch.empty(3 * H, **f32),
             "so_w": torch.empty(H, H, **f32), "so_b": torch.empty(H, **f32), "ffn_i_w": torch.empty(I, H, **f32),
             "ffn_i_b": torch.empty(I, **f32), "ffn_o_w": torch.empty(H, I, **f32), "ffn_o_b": torch.empty(H, **f32)}
        if cross:
            g.update({"xq_w": torch.empty(H, H, **f32), "xq_b": torch.empty(H, **f32), "xo_w": torch.empty(H, H, **f32),
                      "xo_b": torch.empty(H, **f32), "x_ln_g": torch.empty(H, **f32), "x_ln_b": torch.empty(H, **f32)})
        for n in ("s_ln_g", "s_ln_b", "f_ln_g", "f_ln_b"):
            g[n] = torch.empty(H, **f32)
        G = _CXLayerGrads(*[_ptr(g.get(n)) for n, _ in _CXLayerGrads._fields_])
        dx = torch.empty_like(x2)
        dkv, C = None, 0
        if cross:
            C = kv.shape[-1]
            covered = (k_col == 0 and C == 2 * H)
            dkv = (torch.empty if covered else torch.zeros)(B, Sk, C, **f32)    # K / V blocks of other layers: zero gradient here
        ws = torch.empty(int(lib.gridmm_xattn_layer_train_workspace(B, Sq, H, I)), dtype=torch.uint8, device=dev)
        _lib.check(lib.gridmm_xattn_layer_bwd(
            ctypes.byref(ctx.L), _p(x2), _p(kv), _p(kvp[0] if kvp else None), _p(kvp[1] if kvp else None), _p(kvs),
            kvs.stride(0) if kvs is not None else 0, kv.stride(0) if cross else 0, kv.stride(1) if cross else 0, k_col,
            k_col + H, _p(cm),
            cm.stride(0) if cm is not None else 0, _p(sm), sm.stride(0) if sm is not None else 0, _p(saved), saved.numel(),
            _p(dy), _p(dx), _p(dkv), Sk * C, C, ctypes.byref(G), _p(ws), ws.numel(), B, Sq, Sk, heads, _stream()),
            "gridmm_xattn_layer_bwd")
        qw, kw, vw = g["sqkv_w"].split(H, 0)
        qb, kb, vb = g["sqkv_b"].split(H, 0)
        grads = [g.get("xq_w"), g.get("xq_b"), g.get("xo_w"), g.get("xo_b"), qw, qb, kw, kb, vw, vb, g["so_w"], g["so_b"],
                 g["ffn_i_w"], g["ffn_i_b"], g["ffn_o_w"], g["ffn_o_b"], g.get("x_ln_g"), g.get("x_ln_b"), g["s_ln_g"],
                 g["s_ln_b"], g["f_ln_g"], g["f_ln_b"]]
        grads = [DEFERRED.hand(ctx.prm[i], gr) if (gr is not None and ctx.needs_input_grad[9 + i]) else None
                 for i, gr in enumerate(grads)]
        return (dx, dkv, None, None, None, None, None, None, None) + tuple(grads)


class _CPreLNLayer(ctypes.Structure):
    _fields_ = [(n, _CLinearTrain) for n in ("qkv", "out", "ffn1", "ffn2")] + [(n, _CLnTrain) for n in ("ln1", "ln2")] + \
               [("p", ctypes.c_float), ("seed", ctypes.c_ulonglong * 4), ("seed_dev", ctypes.c_void_p)]


class _CPreLNGrads(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("qkv_w", "qkv_b", "out_w", "out_b", "ffn1_w", "ffn1_b", "ffn2_w", "ffn2_b",
                                                "ln1_g", "ln1_b", "ln2_g", "ln2_b")]


class _PreLNLayer(torch.autograd.Function):
    """One pre-LayerNorm transformer layer (transformer.py:170-182) as ONE autograd node: gridmm_preln_layer_train_fwd / _bwd.
    params: in_proj.w, in_proj.b, out_proj.w, out_proj.b, linear1.w, linear1.b, linear2.w, linear2.b, norm1.w, norm1.b,
    norm2.w, norm2.b (12 tensors)."""

    @staticmethod
    def forward(ctx, x, mask, heads, p, eps, *params):
        lib = _lib.load()
        H = heads * 64
        B, S = x.shape[:2]
        x2 = x.float().contiguous()
        (qw, qb, ow, ob, w1, b1, w2, b2, g1, be1, g2, be2) = params
        ctx.prm = params
        I = w1.shape[0]
        keep = []

        def lin(w, b):
            get = WEIGHTS.getter(w)
            pf, pt = get(False), get(True)
            bb = b.detach().float().contiguous()
            keep.extend([pf, pt, bb])
            return _CLinearTrain(pf.hi.data_ptr(), pf.lo.data_ptr(), pf.Kp, pt.hi.data_ptr(), pt.lo.data_ptr(), pt.Kp,
                                 bb.data_ptr(), w.shape[0], w.shape[1])

        def lnp(g, b, e):
            gg, bb = g.detach().float().contiguous(), b.detach().float().contiguous()
            keep.extend([gg, bb])
            return _CLnTrain(gg.data_ptr(), bb.data_ptr(), float(e))
        draw = lambda on: hs.host(lambda: int(torch.randint(0, 2 ** 62, (1,)).item())) if on else 0   # noqa: E731
        seeds = [draw(p > 0) for _ in range(4)]      # attention, after out_proj, after the activation, after linear2
        seed_dev = SEED_DEV if (p > 0 and hs.MODE is not None) else None
        L = _CPreLNLayer(lin(qw, qb), lin(ow, ob), lin(w1, b1), lin(w2, b2), lnp(g1, be1, eps[0]), lnp(g2, be2, eps[1]),
                         float(p), (ctypes.c_ulonglong * 4)(*seeds), _ptr(seed_dev))
        m = None
        if mask is not None:
            m = mask.contiguous()
            m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
        saved = torch.empty(int(lib.gridmm_preln_layer_saved_bytes(B, S, H, I)), dtype=torch.uint8, device=x.device)
        ws = torch.empty(int(lib.gridmm_preln_layer_workspace(B, S, H, I)), dtype=torch.uint8, device=x.device)
        y = torch.empty(B, S, H, dtype=torch.float32, device=x.device)
        _lib.check(lib.gridmm_preln_layer_train_fwd(ctypes.byref(L), _p(x2), _p(m), m.stride(0) if m is not None else 0, _p(y),
                                                    _p(saved), saved.numel(), _p(ws), ws.numel(), B, S, heads, _stream()),
                   "gridmm_preln_layer_train_fwd")
        ctx.save_for_backward(x2, m, saved)
        ctx.L, ctx.keep, ctx.dims = L, keep, (B, S, H, I, heads)
        ctx.versions = _param_versions(ctx.prm)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 17
        lib = _lib.load()
        _check_param_versions(ctx.versions, ctx.prm, "pre-LN layer backward")
        x2, m, saved = ctx.saved_tensors
        B, S, H, I, heads = ctx.dims
        dev = dy.device
        dy = dy.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        g = [torch.empty(3 * H, H, **f32), torch.empty(3 * H, **f32), torch.empty(H, H, **f32), torch.empty(H, **f32),
             torch.empty(I, H, **f32), torch.empty(I, **f32), torch.empty(H, I, **f32), torch.empty(H, **f32),
             torch.empty(H, **f32), torch.empty(H, **f32), torch.empty(H, **f32), torch.empty(H, **f32)]
        G = _CPreLNGrads(*[t.data_ptr() for t in g])
        dx = torch.empty_like(x2)
        ws = torch.empty(int(lib.gridmm_preln_layer_workspace(B, S, H, I)), dtype=torch.uint8, device=dev)
        _lib.check(lib.gridmm_preln_layer_bwd(ctypes.byref(ctx.L), _p(x2), _p(m), m.stride(0) if m is not None else 0, _p(saved),
                                              saved.numel(), _p(dy), _p(dx), ctypes.byref(G), _p(ws), ws.numel(), B, S, heads,
                                              _stream()), "gridmm_preln_layer_bwd")
        grads = [DEFERRED.hand(ctx.prm[i], gr) if ctx.needs_input_grad[5 + i] else None for i, gr in enumerate(g)]
        return (dx, None, None, None, None) + tuple(grads)


def pre_ln_layer_fused(x, mask, heads, p, layer):
    """layer: TransformerEncoderLayer-like (.norm1, .self_attn.in_proj_weight / in_proj_bias / out_proj, .norm2, .linear1,
    .linear2); p: its dropout probability in train() mode (0 otherwise)."""
    a = layer.self_attn
    params = (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, layer.linear1.weight, layer.linear1.bias,
              layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight,
              layer.norm2.bias)
    return _PreLNLayer.apply(x, mask, heads, float(p), (float(layer.norm1.eps), float(layer.norm2.eps)), *params)


def bert_layer_fused(x, self_mask, heads, p_hidden, p_attn, selfatt, inter, output):
    """BertLayer (vilmodel.py:214-231: BertAttention + BertIntermediate + BertOutput) as ONE autograd node: the layer C calls
    with no context (KV = NULL).  Same kernels in the same order as vilmodel_train.bert_layer's op-by-op form."""
    s = selfatt.self
    params = (None, None, None, None, s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias,
              selfatt.output.dense.weight, selfatt.output.dense.bias, inter.dense.weight, inter.dense.bias,
              output.dense.weight, output.dense.bias, None, None,
              selfatt.output.LayerNorm.weight, selfatt.output.LayerNorm.bias, output.LayerNorm.weight, output.LayerNorm.bias)
    eps = (0.0, float(selfatt.output.LayerNorm.eps), float(output.LayerNorm.eps))
    return _XLayer.apply(x, None, None, self_mask, 0, heads, float(p_hidden), float(p_attn), eps, *params)


def x_layer_fused(x, kv, ctx_mask, self_mask, k_col, heads, p_hidden, p_attn, xatt, selfatt, inter, output):
    """xatt: BertXAttention-like (.att.query, .output.dense, .output.LayerNorm); selfatt: BertAttention-like (.self.query /
    key / value, .output.dense / LayerNorm); inter / output: BertIntermediate / BertOutput."""
    s = selfatt.self
    params = (xatt.att.query.weight, xatt.att.query.bias, xatt.output.dense.weight, xatt.output.dense.bias,
              s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias,
              selfatt.output.dense.weight, selfatt.output.dense.bias, inter.dense.weight, inter.dense.bias,
              output.dense.weight, output.dense.bias, xatt.output.LayerNorm.weight, xatt.output.LayerNorm.bias,
              selfatt.output.LayerNorm.weight, selfatt.output.LayerNorm.bias, output.LayerNorm.weight, output.LayerNorm.bias)
    eps = (float(xatt.output.LayerNorm.eps), float(selfatt.output.LayerNorm.eps), float(output.LayerNorm.eps))
    return _XLayer.apply(x, kv, ctx_mask, self_mask, k_col, heads, float(p_hidden), float(p_attn), eps, *params)
