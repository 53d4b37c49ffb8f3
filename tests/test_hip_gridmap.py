"""GPU: fill_gridmap on HIP vs the reference's golden vectors (bit-exact cell ids) and vs the oracle."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import gridmap_oracle as G

pytestmark = pytest.mark.gpu


def _mem(B, geom, steps):
    from gridmm_amd.grid_memory import GridMemoryBatch
    return GridMemoryBatch(B, geom, max_steps=steps, device="cuda")


def test_cell_ids_bit_exact_vs_reference_golden():
    from gridmm_amd import synthetic as S
    fx = load_golden("fill_gridmap_native.npz")
    E = int(fx["n_episodes"])
    T = max(int(fx["e%d_steps" % e]) for e in range(E))
    mem = _mem(E, S.NATIVE, T)
    rs = np.random.RandomState(0)
    for t in range(T):
        act = np.array([t < int(fx["e%d_steps" % e]) for e in range(E)])
        depth = np.zeros((E, 588), np.uint16)
        poses, heads = [(0.0, 0.0)] * E, [0.0] * E
        for e in range(E):
            if act[e]:
                depth[e] = fx["e%d_t%d_depth" % (e, t)].reshape(-1)
                x, y, h = [float(v) for v in fx["e%d_t%d_pose" % (e, t)]]
                poses[e], heads[e] = (x, y), h
        feats = rs.standard_normal((E, 588, 768)).astype(np.float16)
        mem.step(depth, feats, poses, heads, active=None if act.all() else act)
        torch.cuda.synchronize()
        for e in range(E):
            if not act[e]:
                continue
            n = 588 * (t + 1)
            ids = mem.cell_id[e, :n].cpu().numpy()
            assert np.array_equal(ids, fx["e%d_t%d_grid_map" % (e, t)]), (e, t)
            assert np.allclose(mem.pos_fts[e].cpu().numpy(), fx["e%d_t%d_pos_fts" % (e, t)], atol=2e-6), (e, t)
            assert torch.equal(mem.grid_fts(e)[-588:].cpu(), torch.from_numpy(feats[e]))
            # sorted lists are a stable counting sort of the ids
            perm = mem.perm[e, :n].cpu().numpy()
            cs = mem.cell_start[e].cpu().numpy()
            key = np.where(ids < 0, 196, ids)
            assert np.array_equal(perm, np.argsort(key, kind="stable"))
            assert np.array_equal(cs[:197], np.searchsorted(np.sort(key), np.arange(197)))
            assert cs[197] == n


@pytest.mark.parametrize("geom_name,B,T", [("NATIVE", 5, 6), ("BASELINE", 4, 3)])
def test_matches_oracle_on_random_walks(geom_name, B, T):
    from gridmm_amd import synthetic as S
    geom, og = getattr(S, geom_name), getattr(G, geom_name)
    rs = np.random.RandomState(11)
    eps = [S.make_observations(rs, geom, T) for _ in range(B)]
    for ob in eps[1]:
        ob["heading"] = float(rs.uniform(-9, 9))          # arbitrary headings, not only k*30 deg
    oracles = [G.GridMemory(og) for _ in range(B)]
    mem = _mem(B, geom, T)
    for t in range(T):
        depth = np.stack([eps[b][t]["depth"].reshape(-1) for b in range(B)])
        feats = np.stack([eps[b][t]["feats"] for b in range(B)])
        mem.step(depth, feats, [(eps[b][t]["x"], eps[b][t]["y"]) for b in range(B)],
                 [eps[b][t]["heading"] for b in range(B)])
        for b in range(B):
            o = eps[b][t]
            f, gm, pf, hl = oracles[b].step(o["depth"], o["feats"], o["x"], o["y"], o["heading"])
            n = gm.shape[0]
            assert np.array_equal(mem.cell_id[b, :n].cpu().numpy(), gm.astype(np.int16)), (b, t)
            assert np.float32(mem.half_len[b].item()) == hl
            assert np.array_equal(mem.hist_x[b, :n].cpu().numpy(), np.concatenate(oracles[b].hist_x))
            assert np.allclose(mem.pos_fts[b].cpu().numpy(), pf, atol=2e-6)
            assert torch.equal(mem.grid_fts(b).cpu(), torch.from_numpy(f))
    fts, gmaps, pos = mem.as_reference_obs()
    assert gmaps[0].dtype == torch.float64 and fts[0].dtype == torch.float16 and pos.shape == (B, 196, 5)


def test_rebinning_is_idempotent_and_permutation_complete():
    """Size-independent properties at the BASELINE size: perm is a permutation; re-running the
    binning with the same pose reproduces the same ids."""
    from gridmm_amd import synthetic as S, ops
    rs = np.random.RandomState(3)
    B, T = 8, 2
    mem = _mem(B, S.BASELINE, T)
    for t in range(T):
        d = rs.randint(0, 20000, size=(B, 7056)).astype(np.uint16)
        f = rs.standard_normal((B, 7056, 512)).astype(np.float16)
        poses = [(float(rs.uniform(-5, 5)), float(rs.uniform(-5, 5))) for _ in range(B)]
        heads = [float(rs.randint(0, 12)) * math.pi / 6 for _ in range(B)]
        mem.step(d, f, poses, heads)
    n = int(mem.n_pts_host[0])
    ids0, perm0 = mem.cell_id.clone(), mem.perm.clone()
    pose = torch.tensor([[np.float32(p[0]), np.float32(p[1])] for p in poses], device="cuda")
    hcs = torch.tensor([[np.float32(math.cos(-h)), np.float32(math.sin(-h))] for h in heads], device="cuda")
    ops.grid_bin(mem.hist_x, mem.hist_y, mem.hist_valid, mem.n_pts, pose, hcs, mem.half_len, mem.cell_id, mem.perm,
                 mem.cell_start)
    assert torch.equal(ids0, mem.cell_id) and torch.equal(perm0, mem.perm)
    for b in range(B):
        assert torch.equal(torch.sort(mem.perm[b, :n].long())[0], torch.arange(n, device="cuda"))


def test_vlnce_twin_cell_ids_bit_exact_vs_reference_golden():
    from gridmm_amd import synthetic as S
    fx = load_golden("fill_gridmap_vlnce.npz")
    for name, geom in (("r2r", S.VLNCE_R2R), ("rxr", S.VLNCE_RXR)):
        T = int(fx[name + "_steps"])
        mem = _mem(1, geom, T)
        for t in range(T):
            p = "%s_t%d_" % (name, t)
            x, y, h = [float(v) for v in fx[p + "pose"]]
            mem.step(fx[p + "depth"].reshape(1, -1), np.zeros((1, 588, 768), np.float16), [(x, y)], [h])
            n = 588 * (t + 1)
            assert np.array_equal(mem.cell_id[0, :n].cpu().numpy(), fx[p + "grid_map"]), (name, t)
            assert np.allclose(mem.pos_fts[0].cpu().numpy(), fx[p + "pos_fts"], atol=2e-6), (name, t)


@pytest.mark.gpu
@pytest.mark.parametrize("slices", [2, 8, 16])
def test_sliced_rebin_is_identical_to_the_single_workgroup_sort(slices):
    """gridmm_grid_bin_sliced (histogram | scan | scatter over `slices` workgroups per episode, used for memories of more
    than a few thousand points) vs gridmm_grid_bin: same cell ids, same cell_start, same stable order -- on ragged
    histories (different lengths per episode, one empty, one not a multiple of anything)."""
    from gridmm_amd import ops
    B, cap = 5, 40000
    g = torch.Generator().manual_seed(slices)
    n = torch.tensor([40000, 12345, 0, 63, 7056], dtype=torch.int32)
    hx = (torch.rand(B, cap, generator=g) * 30 - 15).cuda()
    hy = (torch.rand(B, cap, generator=g) * 30 - 15).cuda()
    hv = (torch.rand(B, cap, generator=g) > 0.1).to(torch.uint8).cuda()
    pose = (torch.rand(B, 2, generator=g) * 4 - 2).cuda()
    ang = torch.rand(B, generator=g) * 6.28
    head = torch.stack([torch.cos(ang), torch.sin(ang)], 1).cuda()
    half = (torch.rand(B, generator=g) * 10 + 8).cuda()
    outs = []
    for S in (1, slices):
        cid = torch.full((B, cap), -7, dtype=torch.int16, device="cuda")
        perm = torch.full((B, cap), -1, dtype=torch.int32, device="cuda")
        cs = torch.zeros(B, 198, dtype=torch.int32, device="cuda")
        ws = torch.empty(B, S * 17, 197, dtype=torch.int32, device="cuda")
        ops.grid_bin(hx, hy, hv, n.cuda(), pose, head, half, cid, perm, cs, 0, workspace=ws, slices=S)
        torch.cuda.synchronize()
        outs.append((cid.cpu(), perm.cpu(), cs.cpu()))
    (c0, p0, s0), (c1, p1, s1) = outs
    assert torch.equal(s0, s1)
    for b in range(B):
        k = int(n[b])
        assert int(s0[b, 197]) == k
        assert torch.equal(c0[b, :k], c1[b, :k]) and torch.equal(p0[b, :k], p1[b, :k])
        assert sorted(p1[b, :k].tolist()) == list(range(k))                      # a permutation
        assert (p1[b, k:] == -1).all() and (c1[b, k:] == -7).all()               # nothing written past the history


def test_tracked_cell_count_is_dropped_when_the_memory_is_rebinned_by_another_route():
    """GridMemoryBatch.cmax_hint() feeds the varlen bucket choice of NavigationGraphs / the eager varlen path: a count
    recorded by step() must not survive a re-binning that did not go through step() (set_pose + project_and_bin, a graph
    replay after set_pose, reset) -- a stale count that is too small would silently drop occupied cells."""
    from gridmm_amd import synthetic as S
    rs = np.random.RandomState(3)
    mem = _mem(2, S.NATIVE, 3)
    mem.track_cmax = True
    obs = [S.make_observations(rs, S.NATIVE, 2, with_feats=True) for _ in range(2)]
    d0 = np.stack([o[0]["depth"].reshape(-1) for o in obs])
    f0 = np.stack([o[0]["feats"] for o in obs])
    mem.step(d0, f0, [(o[0]["x"], o[0]["y"]) for o in obs], [o[0]["heading"] for o in obs])
    cs = mem.cell_start[:, :197]
    assert mem.cmax_hint() == int((cs[:, 1:] > cs[:, :-1]).sum(1).max())
    mem.set_pose([(o[1]["x"], o[1]["y"]) for o in obs], [o[1]["heading"] for o in obs])
    assert mem.cmax_hint() is None
    mem.step(d0, f0, [(o[1]["x"], o[1]["y"]) for o in obs], [o[1]["heading"] for o in obs])
    assert mem.cmax_hint() is not None
    mem.project_and_bin(torch.from_numpy(d0.astype(np.int32)).to(torch.uint16).cuda().reshape(2, -1))
    assert mem.cmax_hint() is None
    mem.reset()
    assert mem.cmax_hint() is None


def test_stage_extra_hands_out_disjoint_regions():
    from gridmm_amd import synthetic as S
    mem = _mem(2, S.NATIVE, 1)
    h0, d0 = mem.stage_extra(100)
    h1, d1 = mem.stage_extra(40)
    assert h0.data_ptr() + 100 <= h1.data_ptr() and d0.data_ptr() + 100 <= d1.data_ptr()
    assert h1.data_ptr() % 16 == 0
    with pytest.raises(ValueError):
        mem.stage_extra(mem.STAGE_EXTRA)


def test_cell_count_max_equals_the_host_expression():
    """gridmm_grid_cell_count_max: the batch's largest occupied-cell count (the reference's max_cell_num,
    map_nav_src/models/vilmodel.py:809-823) from the binning's cell table in one launch."""
    import torch
    from gridmm_amd import ops
    g = torch.Generator().manual_seed(4)
    for B in (1, 5, 32):
        sizes = torch.randint(0, 4, (B, 197), generator=g) * (torch.rand(B, 197, generator=g) < 0.6)
        cs = torch.zeros(B, 198, dtype=torch.int32)
        cs[:, 1:] = torch.cumsum(sizes, 1).to(torch.int32)
        want = int((cs[:, 1:197] > cs[:, :196]).sum(1).max())
        out = torch.full((1,), -7, dtype=torch.int32, device="cuda")
        ops.grid_cell_count_max(cs.cuda().contiguous(), out)
        assert int(out[0]) == want
