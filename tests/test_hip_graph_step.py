"""The path bench.py times: GraphedNavStep (hipGraph replay of fill_gridmap + forward('navigation')) at the bench
configuration -- B = 32 episodes, 36 x 196 x 512 slab, full-size model -- against (i) the same step launched eagerly
and (ii) the CPU oracle for all 32 episodes (cell ids exact, logits within 1e-3; north star tolerance)."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
KEYS = ("global_logits", "local_logits", "grid_logits", "fused_logits")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda")


def _args(**kw):
    d = dict(batch=32, shape="baseline", mem_steps=1, eager=False)
    d.update(kw)
    return argparse.Namespace(**d)


def _finite_equal(a, w, atol):
    f = torch.isfinite(w)
    assert torch.equal(f, torch.isfinite(a))
    return float((a[f] - w[f]).abs().max()) if f.any() else 0.0, atol


@pytest.mark.parametrize("t", [1, 5])
def test_graph_replay_equals_eager_at_bench_config(dev, t):
    import bench
    model, batch, mem, eps, step, eager_step, geom = bench.build_workload(_args(mem_steps=t), dev, device_feats=(t > 1))
    for _ in range(2):                       # replays are repeatable
        got = {k: v.clone() for k, v in step().items() if k in KEYS}
        cells = mem.cell_id.clone()
        torch.cuda.synchronize()
    want = eager_step()
    torch.cuda.synchronize()
    assert torch.equal(cells, mem.cell_id)
    for k in KEYS:
        err, _ = _finite_equal(got[k], want[k], 0.0)
        assert err <= 1e-6, (k, err)
    chk = bench.check_replay(step, eager_step)
    assert chk["replay_vs_eager_max_abs"] <= 1e-6


@pytest.mark.parametrize("t", [1, 4])
def test_native_shape_replay_with_kept_relevance_equals_the_recompute(dev, t):
    """bench.py --shape native (the reference's 12 x 49 x 768 observations, two-pass aggregation): the captured step with the
    relevance of earlier observations kept beside the slab (gridmm_grid_aggregate_incremental inside the graph, every replay
    restoring the same history prefix) gives the logits of the step that recomputes every point -- bit for bit -- and of its own
    eager launches."""
    import bench
    outs = {}
    for keep in (True, False):
        a = _args(shape="native", mem_steps=t, batch=8, no_relevance_cache=not keep)
        model, batch, mem, eps, step, eager_step, geom = bench.build_workload(a, dev)
        assert mem.relevance_cache_enabled is keep and mem.relevance_cache_in_graphs is keep
        for _ in range(3):
            got = {k: v.clone() for k, v in step().items() if k in KEYS}
        torch.cuda.synchronize()
        assert (mem._rel is not None) is keep
        chk = bench.check_replay(step, eager_step)
        assert chk["bit_identical"]
        outs[keep] = got
        del model, batch, mem, step, eager_step
        torch.cuda.empty_cache()
    for k in KEYS:
        f = torch.isfinite(outs[True][k])
        assert torch.equal(f, torch.isfinite(outs[False][k])) and torch.equal(outs[True][k][f], outs[False][k][f]), k


def test_graph_replay_matches_oracle_all_32_episodes(dev):
    """t = 1 (the headline depth), host-generated features: every episode's cell ids exact vs the NumPy oracle, all four
    logit sets within 1e-3 of oracle.forward_navigation on the same weights."""
    import bench
    from oracle import gridmap_oracle as G, navcmt_oracle as O
    model, batch, mem, eps, step, eager_step, geom = bench.build_workload(_args(), dev)
    got = {k: v.clone().cpu() for k, v in step().items() if k in KEYS}
    torch.cuda.synchronize()
    refs = []
    for b in range(32):
        om = G.GridMemory(G.BASELINE)
        for o in eps[b]:
            r = om.step(o["depth"], o["feats"], o["x"], o["y"], o["heading"])
        refs.append(r)
        n = r[1].shape[0]
        assert np.array_equal(mem.cell_id[b, :n].cpu().numpy(), r[1].astype(np.int16)), "cell ids differ, episode %d" % b
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    keys = ("txt_embeds", "txt_masks", "gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_masks",
            "gmap_visited_masks", "vp_img_embeds", "vp_pos_fts", "vp_masks", "vp_nav_masks")
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    worst = 0.0
    for b0 in range(0, 32, 8):               # chunks of 8 episodes keep the oracle's (N, 768) temporaries small
        sl = slice(b0, b0 + 8)
        cb = {k: batch[k][sl].cpu() for k in keys}
        cb.update(gmap_vpids=batch["gmap_vpids"][sl], vp_cand_vpids=batch["vp_cand_vpids"][sl], vp_obj_masks=None,
                  gmap_pair_dists=None, grid_fts=[torch.from_numpy(r[0]) for r in refs[sl]],
                  grid_map=[torch.from_numpy(r[1]) for r in refs[sl]],
                  gridmap_pos_fts=torch.from_numpy(np.stack([r[2] for r in refs[sl]])))
        with torch.no_grad():
            want = O.forward_navigation(sd, cb)
        for k in KEYS:
            err, _ = _finite_equal(got[k][sl], want[k], 1e-3)
            worst = max(worst, err)
    assert worst < 1e-3, worst


def test_bucketed_back_graphs_equal_the_padded_sequence(dev):
    """Varlen map sequences (vilmodel.py:809-823 max_cell_num): on 'ring' depth (~90-120 occupied cells) the two-graph
    step with per-bucket back halves reproduces the 196-row padded step (same masks, logits to 1e-5), predicts the bucket
    from the previous step, and redoes a step whose count exceeds its bucket."""
    import bench
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT
    buckets = GlocalTextPathNavCMT.DEFAULT_BUCKETS
    model, batch, mem, eps, step, eager_step, geom = bench.build_workload(_args(), dev, device_feats=True, depth_mode="ring",
                                                                          buckets=buckets)
    g = step.graph
    got = {k: v.clone() for k, v in step().items() if k in KEYS}
    torch.cuda.synchronize()
    cmax = int(g.cmax[g.last_bucket].item())
    assert 40 < cmax <= g.last_bucket < 196, (cmax, g.last_bucket)          # a real truncation, correctly bucketed
    want = eager_step()                                                      # 196-row eager path, same inputs
    torch.cuda.synchronize()
    for k in KEYS:
        err, _ = _finite_equal(got[k], want[k], 0.0)
        assert err <= 1e-5, (k, err)
    # a too-small prediction is detected from the count the back graph writes, and redone on the right bucket
    g.bucket = buckets[0]
    redo = {k: v.clone() for k, v in g(*[[(e[0]["x"], e[0]["y"]) for e in eps], [e[0]["heading"] for e in eps]],
                                       check=True).items() if k in KEYS}
    torch.cuda.synchronize()
    assert g.redone == 1 and g.last_bucket >= cmax and g.bucket == g.last_bucket
    for k in KEYS:
        err, _ = _finite_equal(redo[k], want[k], 0.0)
        assert err <= 1e-5, (k, err)
    # eager calls with varlen on read the count on the host and run on the same bucket
    model.varlen_buckets = buckets
    try:
        ev = eager_step()
        torch.cuda.synchronize()
        assert ev["gmap_embeds"].shape == want["gmap_embeds"].shape
        for k in KEYS:
            err, _ = _finite_equal(ev[k], want[k], 0.0)
            assert err <= 1e-5, (k, err)
    finally:
        model.varlen_buckets = None


def _oracle_check(model, batch, mem, eps, got, B, geom_oracle, keys_out, tol=1e-3):
    from oracle import gridmap_oracle as G, navcmt_oracle as O
    refs = []
    for b in range(B):
        om = G.GridMemory(geom_oracle)
        for o in eps[b]:
            r = om.step(o["depth"], o["feats"], o["x"], o["y"], o["heading"])
        refs.append(r)
        n = r[1].shape[0]
        assert np.array_equal(mem.cell_id[b, :n].cpu().numpy(), r[1].astype(np.int16)), "cell ids differ, episode %d" % b
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    keys = ("txt_embeds", "txt_masks", "gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_masks",
            "gmap_visited_masks", "vp_img_embeds", "vp_pos_fts", "vp_masks", "vp_nav_masks")
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    cb = {k: batch[k].cpu() for k in keys}
    cb.update(gmap_vpids=batch["gmap_vpids"], vp_cand_vpids=batch["vp_cand_vpids"],
              vp_obj_masks=None if batch.get("vp_obj_masks") is None else batch["vp_obj_masks"].cpu(),
              gmap_pair_dists=None, grid_fts=[torch.from_numpy(r[0]) for r in refs],
              grid_map=[torch.from_numpy(r[1]) for r in refs], gridmap_pos_fts=torch.from_numpy(np.stack([r[2] for r in refs])))
    with torch.no_grad():
        want = O.forward_navigation(sd, cb)
    worst = 0.0
    for k in keys_out:
        err, _ = _finite_equal(got[k], want[k], tol)
        worst = max(worst, err)
    assert worst < tol, worst
    return worst


def test_config1_batch1_full_size_matches_oracle(dev):
    """BASELINE.json configs[0] at its stated size: B = 1, full-size model, the graph replay bench.py's b1_latency key times."""
    import bench
    from oracle import gridmap_oracle as G
    model, batch, mem, eps, step, eager_step, geom = bench.build_workload(_args(), dev, batch_size=1)
    got = {k: v.clone().cpu() for k, v in step().items() if k in KEYS}
    torch.cuda.synchronize()
    assert got["fused_logits"].shape == (1, 20)
    _oracle_check(model, batch, mem, eps, got, 1, G.BASELINE, KEYS)


def test_config4_reverie_b16_full_size_matches_oracle(dev):
    """BASELINE.json configs[3] at its stated size: B = 16, 36 views + 21 object tokens (V1 = 58), obj_logits from og_head
    (map_nav_src/reverie/env.py:263-372, vilmodel.py:903-907), full-size model, graph replay vs the oracle."""
    import bench
    from oracle import gridmap_oracle as G
    model, batch, mem, eps, step, eager_step, geom = bench.build_workload(_args(), dev, batch_size=16, with_obj=True, n_obj=21)
    out = step()
    got = {k: v.clone().cpu() for k, v in out.items() if k in KEYS + ("obj_logits",)}
    torch.cuda.synchronize()
    assert got["obj_logits"].shape == (16, 58) and got["local_logits"].shape == (16, 58)
    assert int(torch.isfinite(got["obj_logits"]).sum(1).max()) == 21            # the largest object set of the batch
    _oracle_check(model, batch, mem, eps, got, 16, G.BASELINE, KEYS + ("obj_logits",))
