"""Host decisions of a training step, recorded once and replayed under hipGraph capture.

The pre-training forward takes a handful of decisions on the host that depend on the batch's METADATA (lengths, vpid
lists, which tokens are masked): shapes from `int(lens.max())`, index tensors built in python loops and uploaded,
`x[mask]` gathers (pretrain_src/model/pretrain_cmt.py:131-290 and vilmodel.py:569-604 do the same on the host).
Each of them is a device synchronisation or a pageable H2D copy, which a stream capture cannot contain.  They all go
through this module:

    MODE is None   plain eager behaviour: the callable runs where it stands (default; nothing is recorded)
    record()       the callables run (synchronising) and their results -- python scalars and device tensors -- go on a tape
    replay(tape)   the callables do NOT run: the taped values come back in the same order (the code path is the same
                   because the batch metadata is); device tensors on the tape are persistent, so a captured graph may
                   read them

A tape is valid for batches with the same metadata as the recorded one (train_graph.GraphedTrainStep keeps one graph
per tape).
"""
import contextlib

MODE = None
_tape, _pos = None, 0


@contextlib.contextmanager
def record():
    global MODE, _tape, _pos
    assert MODE is None
    MODE, _tape, _pos = "record", [], 0
    try:
        yield _tape
    finally:
        MODE = None


@contextlib.contextmanager
def replay(tape):
    global MODE, _tape, _pos
    assert MODE is None
    MODE, _tape, _pos = "replay", tape, 0
    try:
        yield
        assert _pos == len(tape), "host decisions consumed: %d of %d" % (_pos, len(tape))
    finally:
        MODE = None


def host(fn):
    """fn() -> python value(s) and / or device tensors; may synchronise or upload."""
    global _pos
    if MODE is None:
        return fn()
    if MODE == "record":
        v = fn()
        _tape.append(v)
        return v
    v = _tape[_pos]
    _pos += 1
    return v


def select(x, mask):
    """x[mask] for a boolean mask over the leading dims of x.  Eager: exactly that.  Recorded / replayed: the row ids
    are a host decision, the gather is an index_select (same values, same gradient)."""
    if MODE is None:
        return x[mask]
    idx = host(lambda: mask.reshape(-1).nonzero().squeeze(1))
    return x.reshape((-1,) + tuple(x.shape[mask.dim():])).index_select(0, idx)


# ---- backward segments (train_graph.GraphedTrainStep with several ranks) -------------------------------------------
# A captured backward that should overlap its gradient exchange has to be cut into several graphs: bucket k's exchange can
# only start once a graph that finishes its gradients has been launched.  The model marks natural separators of its
# autograd graph with boundary(level, t); while CUTS is a list the marked tensor is replaced by a detached leaf, so that
# the part of the graph above it and the part below it are separate autograd graphs; segmented_backward() then runs them
# level by level, handing each leaf's accumulated gradient to the tensor it was cut from.  Level j tensors may only be
# consumed by code whose own boundaries have a smaller level (or none): when level j starts, its leaves hold their full
# gradient.  CUTS is None (the default): boundary() returns its argument, the backward is the usual single pass.
CUTS = None


def boundary(level, t):
    if CUTS is None or not t.requires_grad:
        return t
    leaf = t.detach().requires_grad_()
    CUTS.append((level, t, leaf))
    return leaf


def segmented_backward(loss, cuts, segment=None):
    """loss.backward() in len(levels) + 1 passes.  segment: optional context-manager factory called with the segment
    number around each pass (a hipGraph capture per segment).  Same gradients as the single pass up to the summation
    order of tensors with several consumers."""
    import torch
    seg = segment or (lambda k: contextlib.nullcontext())
    with seg(0):
        loss.backward()
    for k, lv in enumerate(sorted({c[0] for c in cuts}), 1):
        roots = [(o, leaf.grad) for (L, o, leaf) in cuts if L == lv and leaf.grad is not None]
        with seg(k):
            if roots:
                torch.autograd.backward([o for o, _ in roots], [g for _, g in roots])
    return len({c[0] for c in cuts}) + 1
