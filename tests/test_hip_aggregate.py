"""gridmm_grid_aggregate across the regimes of its two kernels (aggregate_pipe.hip: D <= 512, 33 <= L <= 96 or so;
aggregate.hip: everything else), against an fp64 restatement of vilmodel.py:793-807 (max over instruction tokens of
x . t, per-cell softmax-weighted sum) on the same fp16 slab.  The point layouts are chosen to hit the kernel's corner
cases: ~1 point per cell (32 runs in a 32-point tile: two slot passes, every row flushed), one crowded cell spanning
dozens of tiles (the open row carried and rescaled from tile to tile), ragged episodes, empty episodes, short last
tiles, chunk counts from 1 to 24."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(slab, ids, text, L):
    """slab (N,D) fp16 cpu, ids (N,) int, text (L,D) fp32 -> cells (196,D) f64, occ (196,), w (N,) f64"""
    x = slab.double()
    w = (x @ text.double().t()).max(-1).values
    cells = torch.zeros(196, x.shape[1], dtype=torch.float64)
    occ = torch.zeros(196, dtype=torch.uint8)
    for c in ids.unique().tolist():
        if c < 0:
            continue
        sel = ids == c
        cells[c] = (torch.softmax(w[sel], 0)[:, None] * x[sel]).sum(0)
        occ[c] = 1
    return cells, occ, w


def _episode_ids(kind, n, rng):
    if kind == "sparse":            # about one point per cell, unsorted
        return rng.integers(0, 196, size=n)
    if kind == "crowded":           # one cell holds 70 % of the points, the rest spread thin
        ids = rng.integers(0, 196, size=n)
        ids[rng.random(n) < 0.7] = 77
        return ids
    if kind == "blocks":            # runs of 5..40 points, some cells empty
        out = []
        while len(out) < n:
            out += [int(rng.integers(0, 196))] * int(rng.integers(5, 41))
        return np.array(out[:n])
    raise ValueError(kind)


CASES = [
    # D, L, kind, points per episode, n_chunks
    (512, 80, "sparse", [150, 97, 0, 260], None),
    (512, 80, "crowded", [3000, 1111, 33, 1], None),
    (512, 80, "blocks", [2048, 2047, 2049, 31], 8),
    (512, 80, "crowded", [5000, 64], 1),
    (512, 80, "blocks", [9000, 4000], 24),
    (512, 96, "blocks", [1500, 700], None),      # 2 B-waves, 16 blocks each
    (512, 48, "sparse", [400, 300], None),       # 3 R-waves, 11 rows each
    (512, 33, "crowded", [1200], None),
    (512, 20, "blocks", [900, 100], None),       # L <= 32: relevance GEMM pass + accumulation pass
    (512, 120, "blocks", [900, 100], None),      # L > 96: relevance GEMM pass + accumulation pass (aggregate_relg.hip)
    (512, 200, "blocks", [2048, 2047, 300, 31], None),   # max_instr_len of the reference's scripts
    (512, 200, "crowded", [9000, 1111, 33, 1], 8),
    (512, 250, "sparse", [400, 300, 0], None),   # 16 token tiles
    (256, 200, "blocks", [3000, 500], None),
    (512, 300, "blocks", [2048, 300, 31], None),   # more than 16 token tiles: relevance GEMM in two launches over token groups
    (768, 300, "crowded", [3000, 64, 700], 8),     # (rxr_pretrain.json: max_txt_len 300)
    (768, 512, "sparse", [400, 300, 0], None),     # BERT's position table
    (256, 270, "crowded", [1500, 40], 4),          # second group of one token tile (17 tiles)
    (256, 200, "crowded", [1500, 40], 4),          # D = 256 accumulation pass, many tiles per workgroup (a round-3 race:
    (256, 120, "crowded", [1500, 40], 8),          #  run-dependent sums, NaN -- the single-group accumulator form)
    (768, 120, "blocks", [2000, 300], None),
    (768, 200, "crowded", [5000, 64, 700], 24),
    (768, 200, "sparse", [150, 97, 0, 260], None),
    (256, 80, "blocks", [3000, 500], None),
    (256, 64, "sparse", [190, 10], 4),
    (768, 80, "blocks", [2000, 300], None),      # D = 768, L <= 80: relevance pass + accumulation pass
    (768, 80, "sparse", [150, 97, 0, 260], None),
    (768, 80, "crowded", [9000, 1111, 33, 1], 8),
    (768, 40, "blocks", [2047, 31], 4),          # 3 token tiles: waves 5..7 hold no text
    (768, 16, "crowded", [700], None),
    (768, 96, "blocks", [900, 100], None),       # D = 768, L > 80: relevance GEMM pass
    (512, 80, "crowded", [210000], 8),           # run-head bitmask of the episode exceeds LDS: generic kernel
    (512, 80, "crowded", [150000], 8),           # largest memories the pipelined kernel takes (4700 tiles per workgroup)
    # point-balanced chunks: the crowded cell is split over many chunks (head pieces, tail pieces, whole-chunk pieces)
    (512, 80, "crowded", [5000, 64, 700], 24),
    (512, 80, "crowded", [40, 33, 32, 31, 1, 2], 8),     # fewer tiles than chunks: empty chunks, one-tile chunks
    (256, 80, "crowded", [3000, 100], 16),
    (768, 80, "crowded", [4000, 10], 24),
]


@pytest.mark.parametrize("D,L,kind,npts,n_chunks", CASES)
def test_grid_aggregate_regimes(D, L, kind, npts, n_chunks):
    from gridmm_amd import ops
    from gridmm_amd.grid_memory import pack_reference_lists
    rng = np.random.default_rng(1234 + D + L + len(npts))
    g = torch.Generator().manual_seed(99 + D + L)
    B = len(npts)
    fts, maps = [], []
    for n in npts:
        fts.append((torch.randn(n, D, generator=g) * 0.5).half())
        maps.append(torch.from_numpy(_episode_ids(kind, n, rng).astype(np.int64)) if n else torch.zeros(0, dtype=torch.int64))
    text = torch.randn(B, L, D, generator=g) * 0.3          # relevance spread of a few units: a real softmax
    slab, perm, cs = pack_reference_lists([f.cuda() for f in fts], [m.cuda().double() for m in maps])
    cells, occ, rel, amax = ops.grid_aggregate(slab, perm, cs, ops.text_fragments(text.cuda()), L, n_chunks=n_chunks,
                                               want_relevance=True, want_amax=True)
    torch.cuda.synchronize()
    again = ops.grid_aggregate(slab, perm, cs, ops.text_fragments(text.cuda()), L, n_chunks=n_chunks, want_relevance=True,
                               want_amax=True)
    assert torch.equal(again[0], cells) and torch.equal(again[2], rel)          # no run-to-run variation on any path
    if max(npts) <= 45000:
        # every shape up to L = 512 runs on a pipelined path (past 256 tokens: the relevance GEMM over two token groups) (one pass: D <= 512, 33 <= L <= 96; else relevance +
        # accumulation passes), which also delivers the backward's routing; the generic kernel (rc 1) does not
        assert ops.LAST_AGGREGATE_RC == 0 and amax is not None
    for b in range(B):
        ref_cells, ref_occ, w = _ref(fts[b], maps[b], text[b], L)
        assert torch.equal(occ[b].cpu(), ref_occ)
        n_valid = int(cs[b, 196])
        assert n_valid == npts[b]
        if n_valid:
            got = torch.zeros(n_valid, dtype=torch.float64)
            got[perm[b, :n_valid].long().cpu()] = rel[b, :n_valid].double().cpu()   # relevance is by sorted position
            assert (got - w).abs().max() < 2e-5 * max(1.0, float(w.abs().max()))
            if amax is not None:                     # arg-max token: attains the maximum (ties / near-ties may differ)
                tok = amax[b, :n_valid].long().cpu()
                assert int(tok.min()) >= 0 and int(tok.max()) < L
                pts = perm[b, :n_valid].long().cpu()
                s_at = (fts[b][pts].double() * text[b][tok].double()).sum(-1)
                assert (s_at - w[pts]).abs().max() < 2e-5 * max(1.0, float(w.abs().max()))
                exact = (fts[b].double() @ text[b].double().t()).argmax(-1)[pts]
                assert (tok == exact).float().mean() > 0.999
        err = (cells[b].double().cpu() - ref_cells).abs().max()
        assert err < 3e-5, (b, float(err))
        assert (cells[b][ref_occ.cuda() == 0] == 0).all()


def test_grid_aggregate_is_deterministic_and_chunk_invariant():
    """Same inputs, same chunking: bit-identical (no atomics: a cell is reduced in point order inside a workgroup, the
    pieces of a cell that a chunk boundary splits are merged in chunk order).  Different chunkings only move the 32-point
    tile / piece boundaries inside a cell: fp32 summation-order noise."""
    from gridmm_amd import ops
    from gridmm_amd.grid_memory import pack_reference_lists
    rng = np.random.default_rng(5)
    g = torch.Generator().manual_seed(5)
    n, D, L = 4000, 512, 80
    fts = [(torch.randn(n, D, generator=g) * 0.5).half().cuda()]
    maps = [torch.from_numpy(_episode_ids("blocks", n, rng)).double().cuda()]
    frag = ops.text_fragments((torch.randn(1, L, D, generator=g) * 0.3).cuda())
    slab, perm, cs = pack_reference_lists(fts, maps)
    outs = [ops.grid_aggregate(slab, perm, cs, frag, L, n_chunks=k)[0].clone() for k in (8, 8, 1, 32)]
    assert torch.equal(outs[1], outs[0])
    for o in outs[2:]:
        assert (o - outs[0]).abs().max() < 2e-6


def test_grid_aggregate_chunk_invariance_at_bench_depth():
    """The benchmark's deepest memory (t = 15: 105 840 points per episode, D = 512, L = 80) with crowded cells -- thousands
    of points in a few cells, each split over several workgroups: the result does not depend on how many chunks an episode
    is cut into (1 = no split at all), every launch is bit-reproducible, and each cell is a convex combination of its
    points (the softmax weights are positive and sum to one)."""
    from gridmm_amd import ops
    from gridmm_amd.grid_memory import pack_reference_lists
    rng = np.random.default_rng(11)
    g = torch.Generator().manual_seed(11)
    B, n, D, L = 4, 105840, 512, 80
    fts = [(torch.randn(n, D, generator=g) * 0.5).half().cuda() for _ in range(B)]
    maps = []
    for b in range(B):
        ids = rng.integers(0, 196, size=n)
        hot = rng.choice(196, size=5, replace=False)
        sel = rng.random(n) < 0.55                       # 55 % of the points in five cells: ~11 600 points each
        ids[sel] = hot[rng.integers(0, 5, size=int(sel.sum()))]
        ids[rng.random(n) < 0.02] = -1                   # out-of-window points
        maps.append(torch.from_numpy(ids).double().cuda())
    frag = ops.text_fragments((torch.randn(B, L, D, generator=g) * 0.3).cuda())
    slab, perm, cs = pack_reference_lists(fts, maps)
    ref_cells, ref_occ = ops.grid_aggregate(slab, perm, cs, frag, L, n_chunks=1)
    ref_cells, ref_occ = ref_cells.clone(), ref_occ.clone()
    for k in (64, 64, 8, 23, 196):
        cells, occ = ops.grid_aggregate(slab, perm, cs, frag, L, n_chunks=k)
        assert torch.equal(occ, ref_occ)
        assert float((cells - ref_cells).abs().max()) < 5e-6, k
        if k == 64:
            first = cells.clone() if "first" not in locals() else first
            assert torch.equal(cells, first)
    # convexity on a sample of cells: min over the cell's points <= cell vector <= max
    ids0 = maps[0].long()
    for c in [int(x) for x in torch.unique(ids0)[1:6]]:
        pts = fts[0][ids0 == c].float()
        v = ref_cells[0, c]
        assert bool((v <= pts.max(0).values + 1e-4).all() and (v >= pts.min(0).values - 1e-4).all())
