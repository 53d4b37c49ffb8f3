"""Oracle (torch restatement of vilmodel.py:782-918) vs golden vectors produced by the reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden, golden_state_dict, golden_nav_batch
from oracle import navcmt_oracle as O

TOL = 2e-5


def _cmp(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    inf = ~np.isfinite(b)
    assert np.array_equal(~np.isfinite(a), inf)
    assert np.array_equal(a[inf], b[inf])                       # -inf placement identical
    err = np.abs(a[~inf] - b[~inf]).max() if (~inf).any() else 0.0
    assert err <= tol, err
    return err


@pytest.mark.parametrize("name", ["nav_reduced.npz", "nav_reduced_obj.npz"])
def test_navigation_oracle_matches_reference_golden(name):
    fx = load_golden(name)
    sd = golden_state_dict(fx)
    batch = golden_nav_batch(fx)
    with torch.no_grad():
        outs = O.forward_navigation(sd, batch)
        emb, msk, _, _ = O.grid_aggregate(sd, batch["txt_embeds"], batch["grid_fts"], batch["grid_map"],
                                          batch["gridmap_pos_fts"])
    assert np.array_equal(msk.numpy(), fx["cap_grid_masks"])     # incl. the compaction-mask quirk
    _cmp(emb.numpy(), fx["cap_grid_map_embeds"], 1e-5)
    for k in ("gmap_embeds", "vp_embeds", "global_logits", "local_logits", "fused_logits", "grid_logits"):
        _cmp(outs[k].numpy(), fx["out_" + k])
    if "out_obj_logits" in fx.files:
        _cmp(outs["obj_logits"].numpy(), fx["out_obj_logits"])
    else:
        assert outs["obj_logits"] is None


def test_compaction_mask_quirk_is_present_in_golden():
    """vilmodel.py:817-821 mutates a view: rows with fewer cells than Cmax keep stale 1s."""
    fx = load_golden("nav_reduced.npz")
    m = fx["cap_grid_masks"]
    n_occ = [len(set(int(c) for c in fx["in_grid_map_%d" % b] if c >= 0)) for b in range(m.shape[0])]
    assert m.shape[1] == max(n_occ)
    assert any(m[b].sum() != n_occ[b] for b in range(m.shape[0])), "fixture should exercise the quirk"


@pytest.mark.parametrize("fixture", ["text_pano_reduced.npz", "text_pano_full_b2.npz"])
def test_text_and_panorama_oracle_match_reference_golden(fixture):
    """(text_pano_full_b2.npz: the released model size)"""
    fx = load_golden(fixture)
    sd = golden_state_dict(fx)
    with torch.no_grad():
        txt = O.forward_text(sd, torch.from_numpy(fx["in_txt_ids"]), torch.from_numpy(fx["in_txt_masks"]))
        pano, pm = O.forward_panorama(sd, torch.from_numpy(fx["in_view_img_fts"]), torch.from_numpy(fx["in_loc_fts"]),
                                      torch.from_numpy(fx["in_nav_types"]), torch.from_numpy(fx["in_view_lens"]))
    _cmp(txt.numpy(), fx["out_txt_embeds"])
    _cmp(pano.numpy(), fx["out_pano_embeds"])
    assert np.array_equal(pm.numpy(), fx["out_pano_masks"])


def test_navigation_oracle_matches_reference_golden_at_full_size():
    """nav_full_b2.npz: the 161 M-parameter configuration, B = 2, 1764 / 1176 grid points; inputs regenerated from the
    generator's seeds (the fixture stores outputs only)."""
    from oracle import gen_golden
    fx = load_golden("nav_full_b2.npz")
    sd = golden_state_dict(fx)
    batch = gen_golden.full_b2_inputs()
    torch.set_num_threads(4)
    try:
        with torch.no_grad():
            outs = O.forward_navigation(sd, batch)
    finally:
        torch.set_num_threads(1)
    for k in ("gmap_embeds", "vp_embeds", "global_logits", "local_logits", "fused_logits", "grid_logits"):
        _cmp(outs[k].numpy(), fx["out_" + k], 5e-5)
