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
