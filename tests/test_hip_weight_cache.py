"""GPU: optimizer-owned weight planes (round 5; csrc/optim.hip adamw_tiles, autograd._WeightCache).  The fused AdamW launch
writes the bf16 hi / lo planes of every updated 2-D weight in both orientations itself (pretrain_src/optim/adamw.py:56-112 is
the update; the planes are what the forward / dX GEMMs of autograd._Linear read), so no pack launch follows an optimizer step.
Pinned here: the planes the optimizer leaves are BIT-identical to a fresh pack of the updated weight, for single weights and
for the row blocks of a fused q | k | v group; the update itself is bit-identical to the flat (plane-less) kernel; a weight
that sits in two groupings is written in the first one only and the other goes stale (and is re-packed on use)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fresh(w):
    from gridmm_amd import autograd as ag
    return ag._WeightCache._pack_both(w.detach().clone())


def _planes_equal(a, b):
    return torch.equal(a.hi, b.hi) and torch.equal(a.lo, b.lo)


def test_optimizer_writes_planes_bit_identical_to_a_fresh_pack():
    from gridmm_amd import autograd as ag
    from gridmm_amd.optim import AdamW
    dev = torch.device("cuda")
    torch.manual_seed(0)
    ag.WEIGHTS.clear()
    w1 = torch.nn.Parameter(torch.randn(192, 128, device=dev) * 0.05)         # single entry
    q, k, v = (torch.nn.Parameter(torch.randn(128, 128, device=dev) * 0.05) for _ in range(3))   # fused group
    odd = torch.nn.Parameter(torch.randn(40, 128, device=dev) * 0.05)          # N % 64 != 0: no plane record, re-packed
    x = torch.randn(4, 7, 128, device=dev)
    opt = AdamW([w1, q, k, v, odd], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, decay_first=True)
    ref = [p.detach().clone().requires_grad_() for p in (w1, q, k, v, odd)]
    ropt = AdamW(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, decay_first=True)
    for step in range(3):
        y = ag.linear(x, w1) .sum() + ag.linear_group(x, [q, k, v], []).pow(2).sum() + ag.linear(x, odd).sum()
        y.backward()
        for r, p in zip(ref, (w1, q, k, v, odd)):
            r.grad = p.grad.detach().clone()
        opt.step(max_grad_norm=1.0)
        owned_before = ag.OPT_PLANES
        ag.OPT_PLANES = False                      # the reference update: same kernel launch without plane records
        try:
            ropt.step(max_grad_norm=1.0)
        finally:
            ag.OPT_PLANES = owned_before
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        for r, p in zip(ref, (w1, q, k, v, odd)):
            assert torch.equal(r.detach(), p.detach()), step        # tile-wise update == flat update, bit for bit
        # single weight: both orientations
        pf, pt = ag.WEIGHTS.getter(w1)(False), ag.WEIGHTS.getter(w1)(True)
        ff, ft = _fresh(w1)
        assert _planes_equal(pf, ff) and _planes_equal(pt, ft), step
        assert ag.WEIGHTS.optimizer_planes(w1) is not None and ag.WEIGHTS.optimizer_planes(odd) is None
        # fused group: row blocks / column blocks of the shared planes
        get = ag.WEIGHTS.group_getter((q, k, v))
        gf, gt = get(False), get(True)
        cf, ct = _fresh(torch.cat([q.detach(), k.detach(), v.detach()], 0))
        assert _planes_equal(gf, cf) and _planes_equal(gt, ct), step


def test_plane_buffers_keep_their_addresses_and_follow_outside_changes():
    """Captured steps point at the plane buffers: a re-pack (load_state_dict, copy_) must write IN PLACE."""
    from gridmm_amd import autograd as ag
    dev = torch.device("cuda")
    ag.WEIGHTS.clear()
    w = torch.nn.Parameter(torch.randn(128, 64, device=dev))
    pf = ag.WEIGHTS.getter(w)(False)
    ptr = (pf.hi.data_ptr(), pf.lo.data_ptr(), ag.WEIGHTS.getter(w)(True).hi.data_ptr())
    with torch.no_grad():
        w.copy_(torch.randn(128, 64, device=dev))          # version bump behind the cache
    ag.WEIGHTS.ensure_current([w])
    pf2, pt2 = ag.WEIGHTS.getter(w)(False), ag.WEIGHTS.getter(w)(True)
    assert (pf2.hi.data_ptr(), pf2.lo.data_ptr(), pt2.hi.data_ptr()) == ptr
    ff, ft = _fresh(w)
    assert _planes_equal(pf2, ff) and _planes_equal(pt2, ft)


def test_a_weight_in_two_groupings_is_owned_by_the_first():
    from gridmm_amd import autograd as ag
    from gridmm_amd.optim import AdamW
    dev = torch.device("cuda")
    torch.manual_seed(1)
    ag.WEIGHTS.clear()
    k0, v0, k1, v1 = (torch.nn.Parameter(torch.randn(64, 64, device=dev) * 0.1) for _ in range(4))
    x = torch.randn(2, 5, 64, device=dev)
    opt = AdamW([k0, v0, k1, v1], lr=1e-2)
    small = ag.WEIGHTS.group_getter((k0, v0))
    small(False)
    big = ag.WEIGHTS.group_getter((k0, v0, k1, v1))
    big(False)
    (ag.linear_group(x, [k0, v0], []).sum() + ag.linear_group(x, [k0, v0, k1, v1], []).pow(2).sum()).backward()
    opt.step()
    torch.cuda.synchronize()
    small_ent = ag.WEIGHTS._grp[tuple(id(w) for w in (k0, v0))][1]
    big_ent = ag.WEIGHTS._grp[tuple(id(w) for w in (k0, v0, k1, v1))][1]
    cur = lambda ws: tuple((w._version, w.data_ptr(), tuple(w.shape)) for w in ws)       # noqa: E731
    assert small_ent["ver"] == cur((k0, v0))                      # owner of k0, v0: written by the update
    assert big_ent["ver"] != cur((k0, v0, k1, v1))                # the second grouping went stale ...
    gf = ag.WEIGHTS.group_getter((k0, v0, k1, v1))(False)          # ... and is re-packed on use
    cf, _ = _fresh(torch.cat([w.detach() for w in (k0, v0, k1, v1)], 0))
    assert _planes_equal(gf, cf)
    sf = ag.WEIGHTS.group_getter((k0, v0))(False)
    cs, _ = _fresh(torch.cat([k0.detach(), v0.detach()], 0))
    assert _planes_equal(sf, cs)


def test_backward_after_a_weight_update_fails_loudly():
    """the planes a Linear's backward reads (W^T for dX) are persistent buffers that the optimizer rewrites in place,
    and the weight itself is not a saved tensor of the custom Function -- torch's version check cannot see an update that lands
    between forward and backward (retain_graph, delayed backward).  The weight cache's getter and the fused layer nodes remember
    the parameter versions of their forward and refuse a backward against newer weights."""
    from gridmm_amd import autograd as ag
    from gridmm_amd.optim import AdamW
    dev = torch.device("cuda")
    torch.manual_seed(0)
    ag.WEIGHTS.clear()
    w = torch.nn.Parameter(torch.randn(128, 128, device=dev) * 0.05)
    x = torch.randn(2, 9, 128, device=dev, requires_grad=True)
    opt = AdamW([w], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decay_first=True)
    y = ag.linear(x, w).sum()
    y.backward(retain_graph=True)                # fine: same weights
    opt.step(max_grad_norm=1.0)                  # rewrites w and its planes in place
    with pytest.raises(RuntimeError, match="modified in place"):
        y.backward()
    # a fresh forward after the update works
    x.grad = None
    ag.linear(x, w).sum().backward()
    assert torch.isfinite(x.grad).all()
