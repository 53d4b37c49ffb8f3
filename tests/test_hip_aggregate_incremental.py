"""gridmm_grid_aggregate_incremental: the two-pass aggregation of a device-resident memory with the relevance pass restricted
to the points that have no value yet (csrc/aggregate_inc.hip).  The relevance of a point depends only on its slab row and the
instruction (map_nav_src/models/vilmodel.py:797-798), so keeping it across the steps of an episode must not change a single
bit: every check here is torch.equal against gridmm_grid_aggregate on the same state, step after step, with the cells of ALL
points re-drawn at every step (the memory is re-binned in the current egocentric frame, env.py:337-369), ragged histories,
inactive episodes, a rewound memory whose last rows are rewritten, a new instruction, reset()."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state(B, cap, dev="cuda"):
    from gridmm_amd import _lib
    n = int(_lib.load().gridmm_grid_aggregate_incremental_scratch(B, cap))
    return {"hist": torch.zeros(B, cap, dtype=torch.float32, device=dev), "valid": torch.zeros(B, dtype=torch.int32, device=dev),
            "rel": torch.zeros(B, cap, dtype=torch.float32, device=dev), "scratch": torch.empty(n, dtype=torch.uint8, device=dev)}


def _bin(ids, n_pts):
    from gridmm_amd import ops
    B, cap = ids.shape
    perm = torch.empty(B, cap, dtype=torch.int32, device=ids.device)
    cs = torch.empty(B, 198, dtype=torch.int32, device=ids.device)
    ops.grid_sort_ids(ids, n_pts, perm, cs)
    return perm, cs


def _check(slab, ids, n_pts, frag, L, st, active, n_new, n_chunks=None, full=False):
    from gridmm_amd import ops
    perm, cs = _bin(ids, n_pts)
    want = ops.grid_aggregate(slab, perm, cs, frag, L, n_chunks=n_chunks, want_relevance=True)
    got = ops.grid_aggregate_incremental(slab, perm, cs, frag, L, n_pts, active, n_new, st, n_chunks=n_chunks, full=full)
    assert got is not None and ops.LAST_AGGREGATE_RC == 2
    torch.cuda.synchronize()
    assert torch.equal(got[1], want[1])
    assert torch.equal(got[0], want[0])
    for b in range(slab.shape[0]):
        nv = int(cs[b, 196])
        assert torch.equal(st["rel"][b, :nv], want[2][b, :nv])                  # relevance by sorted position
        p = perm[b, :nv].long()
        assert torch.equal(st["hist"][b][p], want[2][b, :nv])                   # ... and kept by history index
    assert torch.equal(st["valid"], n_pts)


@pytest.mark.parametrize("D,L,n_new,steps,n_chunks", [
    (768, 80, 588, 5, None),      # the reference's native observation (12 x 49 tokens of 768 dims): relevance_wide
    (768, 40, 300, 3, 4),
    (768, 96, 588, 3, None),      # L > 80 at D = 768: relevance GEMM
    (512, 120, 700, 3, None),     # long instructions at D = 512: relevance GEMM
    (512, 20, 333, 3, 8),         # short instructions
    (768, 80, 2100, 4, 24),
])
def test_incremental_equals_full_recompute_step_by_step(D, L, n_new, steps, n_chunks):
    from gridmm_amd import ops
    B, cap = 4, n_new * steps
    g = torch.Generator().manual_seed(7 + D + L)
    rng = np.random.default_rng(D + L + n_new)
    slab = torch.zeros(B, cap, D, dtype=torch.float16, device="cuda")
    frag = ops.text_fragments((torch.randn(B, L, D, generator=g) * 0.3).cuda())
    st = _state(B, cap)
    n_pts = torch.zeros(B, dtype=torch.int32, device="cuda")
    n_host = np.zeros(B, np.int64)
    # a point without depth never gets a cell (env.py:283-285, 359-369): validity is a property of the point, the cell is not
    invalid = rng.random((B, cap)) < 0.1

    def draw_ids():
        ids = rng.integers(0, 196, size=(B, cap))
        ids[invalid] = -1
        return torch.from_numpy(ids.astype(np.int16)).cuda()
    for t in range(steps):
        act = np.ones(B, bool)
        if t >= 2:
            act[1] = False                                       # episode 1 ended after two steps
        if t == 1:
            act[3] = False                                       # episode 3 skips a step (ragged histories)
        for b in np.nonzero(act)[0]:
            slab[b, n_host[b]:n_host[b] + n_new] = (torch.randn(n_new, D, generator=g) * 0.5).half().cuda()
            n_host[b] += n_new
        n_pts.copy_(torch.from_numpy(n_host.astype(np.int32)))
        ids = draw_ids()                                         # re-binned: all cells change
        active = torch.from_numpy(act.astype(np.uint8)).cuda()
        # (first step: the `full` form -- the plain passes + one launch that files their values -- as the model front uses it)
        _check(slab, ids, n_pts, frag, L, st, active if t else None, n_new, n_chunks, full=(t == 0 and n_new != 300))
        _check(slab, ids, n_pts, frag, L, st, active if t else None, n_new, n_chunks)           # a repeated call is idempotent
    # a rewound memory: the last observation of episode 0 is replaced (graph replays at a fixed depth restore n_pts and
    # append again): those rows are recomputed although `valid` covers them
    slab[0, n_host[0] - n_new:n_host[0]] = (torch.randn(n_new, D, generator=g) * 0.5).half().cuda()
    act = np.array([True, False, False, False])
    ids = draw_ids()
    _check(slab, ids, n_pts, frag, L, st, torch.from_numpy(act.astype(np.uint8)).cuda(), n_new, n_chunks)
    # a new instruction: the caller clears `valid` (GridMemoryBatch.relevance_cache does on a key change)
    frag2 = ops.text_fragments((torch.randn(B, L, D, generator=g) * 0.3).cuda())
    st["valid"].zero_()
    _check(slab, ids, n_pts, frag2, L, st, None, n_new, n_chunks)
    # ... or asks for the full form, which does not read `valid` at all
    _check(slab, ids, n_pts, frag, L, st, None, n_new, n_chunks, full=True)
    ids = draw_ids()
    _check(slab, ids, n_pts, frag, L, st, torch.zeros(B, dtype=torch.uint8, device="cuda"), n_new, n_chunks)   # all kept


def test_one_pass_shapes_are_refused():
    from gridmm_amd import ops
    B, cap, D, L = 2, 640, 512, 80
    slab = torch.zeros(B, cap, D, dtype=torch.float16, device="cuda")
    n_pts = torch.full((B,), cap, dtype=torch.int32, device="cuda")
    perm, cs = _bin(torch.zeros(B, cap, dtype=torch.int16, device="cuda"), n_pts)
    frag = ops.text_fragments(torch.zeros(B, L, D, device="cuda"))
    assert not ops.two_pass_aggregation(D, L)
    assert ops.grid_aggregate_incremental(slab, perm, cs, frag, L, n_pts, None, 320, _state(B, cap)) is None


def _native_model(fx_name="nav_reduced.npz"):
    from test_hip_navigation import _model, _to_dev
    from conftest import golden_nav_batch, load_golden
    fx = load_golden(fx_name)
    model, _ = _model(fx)
    return model, _to_dev(golden_nav_batch(fx))


def test_navigation_with_kept_relevance_equals_recompute_over_an_episode():
    """forward('navigation') on a GridMemoryBatch of the native geometry (D = 768): the logits of every step are bit-identical
    with and without the kept relevance; the cache is cleared once per instruction / reset, not per step."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.grid_memory import GridMemoryBatch
    model, batch = _native_model()
    B, T = 3, 4
    rs = np.random.RandomState(11)
    mem_a, mem_b = GridMemoryBatch(B, S.NATIVE, max_steps=T), GridMemoryBatch(B, S.NATIVE, max_steps=T)
    mem_b.relevance_cache_enabled = False
    txt2 = batch["txt_embeds"].flip(0).contiguous()
    for episode, txt in enumerate((batch["txt_embeds"], txt2)):
        if episode:
            mem_a.reset()
            mem_b.reset()
        for t in range(T):
            eps = [S.make_observations(rs, S.NATIVE, 1, feat_scale=0.35)[0] for _ in range(B)]
            active = None if t < 2 else [True, False, True]
            for mem in (mem_a, mem_b):
                mem.step(np.stack([e["depth"].reshape(-1) for e in eps]), np.stack([e["feats"] for e in eps]),
                         [(e["x"], e["y"]) for e in eps], [e["heading"] for e in eps], active=active)
            outs = [model("navigation", dict(batch, txt_embeds=txt, grid_fts=None, grid_map=None, gridmap_pos_fts=None,
                                             grid_memory=mem)) for mem in (mem_a, mem_b)]
            for k in ("fused_logits", "grid_logits", "global_logits", "local_logits", "gmap_embeds"):
                a, b = outs[0][k], outs[1][k]
                f = torch.isfinite(a)
                assert torch.equal(f, torch.isfinite(b)) and torch.equal(a[f], b[f]), (episode, t, k)
        assert mem_a._rel["clears"] == episode + 1 and mem_b._rel is None
        assert torch.equal(mem_a._rel["valid"], mem_a.n_pts)


def test_a_caller_with_a_new_instruction_tensor_per_call_falls_back_to_the_plain_passes():
    """Keeping values pays only when the instruction tensor persists over the steps of an episode: four calls in a row that each
    bring a new tensor switch the memory back to gridmm_grid_aggregate (same results either way)."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.grid_memory import GridMemoryBatch
    model, batch = _native_model()
    B = 3
    rs = np.random.RandomState(3)
    mem = GridMemoryBatch(B, S.NATIVE, max_steps=6)
    ref = GridMemoryBatch(B, S.NATIVE, max_steps=6)
    ref.relevance_cache_enabled = False
    for t in range(6):
        eps = [S.make_observations(rs, S.NATIVE, 1, feat_scale=0.35)[0] for _ in range(B)]
        for m in (mem, ref):
            m.step(np.stack([e["depth"].reshape(-1) for e in eps]), np.stack([e["feats"] for e in eps]),
                   [(e["x"], e["y"]) for e in eps], [e["heading"] for e in eps])
        txt = batch["txt_embeds"].clone()                       # same values, new tensor: the key cannot match
        a, b = (model("navigation", dict(batch, txt_embeds=txt, grid_fts=None, grid_map=None, gridmap_pos_fts=None,
                                         grid_memory=m)) for m in (mem, ref))
        f = torch.isfinite(a["fused_logits"])
        assert torch.equal(a["fused_logits"][f], b["fused_logits"][f])
    assert mem.relevance_cache_enabled is False and mem._rel["clears"] == 4
