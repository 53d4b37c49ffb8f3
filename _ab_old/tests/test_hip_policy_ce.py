"""GPU: the VLN-CE policy shell (gridmm_amd/policy_ce.GridMap.forward(mode=...)) against the imported reference's
GridMap.forward(mode='navigation') (Policy_ViewSelection_GridMap.py:500-625) over a scripted three-step episode pair:
trajectory bookkeeping (visited positions, mean panorama embeddings kept per visit, relative-pose features), the tuple
handed to the model, and the [stop]-to-the-back rotation of the logits.  tests/golden/policy_ce_nav.npz."""
import inspect
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import gen_golden as GG
from oracle.ref_harness import det_tensor

pytestmark = pytest.mark.gpu


def _dev(x):
    if torch.is_tensor(x):
        return x.cuda()
    if isinstance(x, list) and x and torch.is_tensor(x[0]):
        return [t.cuda() for t in x]
    return x


def test_policy_shell_navigation_matches_reference_over_three_steps():
    from gridmm_amd.policy_ce import GridMap
    from gridmm_amd.vilmodel_ce import GlocalTextPathNavCMT, default_config
    fx = load_golden("policy_ce_nav.npz")
    model = GlocalTextPathNavCMT(default_config(**json.loads(str(fx["cfg"])))).eval()
    sd = {k: det_tensor(k, v.shape, int(fx["weight_seed"])) if v.dtype.is_floating_point else v for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.cuda()
    inp = GG.policy_ce_inputs()
    pol = GridMap(model, batch_size=GG.POLICY_CE["B"], dataset="R2R")
    pol.start_positions = inp["start"]
    lang_feats = pol(mode="language", lang_idx_tokens=torch.randint(1, 50, (2, 6)).cuda(), lang_masks=torch.ones(2, 6, dtype=torch.bool).cuda())
    assert lang_feats.shape == (2, 6, 768)
    for t, st in enumerate(inp["steps"]):
        pol.positions, pol.headings, pol.action_step = st["positions"], st["headings"], t + 1
        with torch.no_grad():
            logits = pol(mode="navigation", lang_feats=inp["lang_feats"].cuda(), lang_masks=inp["lang_masks"].cuda(),
                         positions=st["positions"], candidate_lengths=st["cand_lens"], batch_angles=st["angles"],
                         batch_distances=st["distances"], batch_view_img_fts=st["view_img_fts"].cuda(),
                         batch_loc_fts=st["loc_fts"].cuda(), batch_nav_types=st["nav_types"].cuda(),
                         batch_view_lens=st["view_lens"].cuda(), batch_grid_fts=_dev(st["grid_fts"]),
                         batch_map_index=_dev(st["grid_map"]), batch_gridmap_pos_fts=st["gridmap_pos_fts"].cuda())
        want = torch.from_numpy(fx["logits_%d" % t])
        got = logits.cpu()
        assert got.shape == want.shape
        f = torch.isfinite(want)
        assert torch.equal(f, torch.isfinite(got)), t
        assert float((got[f] - want[f]).abs().max()) < 2e-4, (t, float((got[f] - want[f]).abs().max()))
    assert [len(m) for m in pol.traj_map] == [3, 3] and pol.traj_embeds[0][0].is_cuda      # visits stay on the device


def test_policy_shell_keeps_the_reference_keyword_signature_and_rejects_waypoint():
    from gridmm_amd.policy_ce import GridMap
    names = list(inspect.signature(GridMap.forward).parameters)[1:]
    assert names[:27] == ["mode", "waypoint_predictor", "observations", "lang_idx_tokens", "lang_masks", "lang_feats",
                          "lang_token_type_ids", "headings", "positions", "cand_rgb", "cand_depth", "cand_direction", "cand_mask",
                          "candidate_lengths", "batch_angles", "batch_distances", "masks", "batch_view_img_fts", "batch_loc_fts",
                          "batch_nav_types", "batch_view_lens", "batch_grid_fts", "batch_map_index", "batch_gridmap_pos_fts",
                          "in_train", "grid_memory"][:27]
    with pytest.raises(NotImplementedError):
        GridMap(None)(mode="waypoint")
    with pytest.raises(NotImplementedError):
        GridMap(None)(mode="nonsense")
