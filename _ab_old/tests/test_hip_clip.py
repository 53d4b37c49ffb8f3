"""GPU: the CLIP ViT-B/32 patch encoder on the HIP kernels (gridmm_amd/clip_encoder.py) against tokens of the imported
reference module (VLN_CE/vlnce_baselines/models/gridmap/clip.py; tests/golden/clip_tokens.npz), and the device-side
hand-off into the grid memory: patch tokens of every view land, fp16, in GridMemoryBatch.next_slot() in the order the
reference appends them (Policy_ViewSelection_GridMap.py:343-357)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import gen_golden as GG

pytestmark = pytest.mark.gpu


def _model(case):
    from gridmm_amd.clip_encoder import CLIP
    c = GG.CLIP_CASES[case]
    m = CLIP(224, 32, c["width"], c["layers"], c["heads"]).eval()
    fx = load_golden("clip_tokens.npz")
    assert list(m.state_dict().keys()) == json.loads(str(fx[case + "_keys"]))      # the reference's state_dict keys
    m.load_state_dict(GG.clip_det_state(m, c["seed"]), strict=True)
    return m.cuda(), torch.from_numpy(fx[case + "_tokens"])


@pytest.mark.parametrize("case", ["reduced", "full"])
def test_clip_tokens_match_reference(case):
    model, want = _model(case)
    got = model(GG.clip_images(case).cuda()).cpu()
    assert got.shape == want.shape and got.dtype == torch.float32
    err = float((got - want).abs().max())
    assert err < 2e-4, err          # ln_post output, |x| up to ~4: 3-term bf16 GEMMs + fp32 LayerNorm / softmax


def test_encode_into_writes_patch_tokens_into_the_grid_memory_slot():
    from gridmm_amd import synthetic as S
    from gridmm_amd.grid_memory import GridMemoryBatch
    c = dict(GG.CLIP_CASES["reduced"])
    model, want = _model("reduced")
    geom = S.GridGeometry(n_views=3, patches=7, feat_dim=c["width"], depth_div=1.0, vlnce=True)
    mem = GridMemoryBatch(2, geom, max_steps=2, device="cuda")           # 2 episodes x 3 views = the 6 images
    slot = mem.next_slot()
    assert slot.shape == (2, 3 * 49, c["width"]) and slot.dtype == torch.float16
    tok = model.encode_into(GG.clip_images("reduced").cuda(), slot, n_views=3)
    torch.cuda.synchronize()
    assert float((tok.cpu() - want).abs().max()) < 2e-4
    ref = want[:, 1:].reshape(2, 3 * 49, c["width"]).to(torch.float16)    # class token dropped, view-major
    assert torch.equal(mem.slab[:, :3 * 49].cpu(), tok[:, 1:].reshape(2, 3 * 49, c["width"]).to(torch.float16).cpu())
    assert float((mem.slab[:, :3 * 49].float().cpu() - ref.float()).abs().max()) < 4e-3      # one fp16 rounding of |x| <= 4
    assert (mem.slab[:, 3 * 49:] == 0).all()                             # nothing else touched


def test_vlnce_model_owns_the_clip_tower_and_feeds_the_memory():
    """with_clip_tower: the VLN-CE twin carries `clip.*` under the reference's keys (visual_encoder.* still ignored) and
    encode_observation() + GridMemoryBatch.step(feats=None) is the whole device-side producer path."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.vilmodel_ce import GlocalTextPathNavCMT, default_config
    cfg = default_config(num_l_layers=1, num_pano_layers=1, num_x_layers=1, intermediate_size=64, vocab_size=50,
                         with_clip_tower=True)
    model = GlocalTextPathNavCMT(cfg).eval()
    full, want = _model("full")
    sd = dict(model.state_dict())
    sd.update({"clip." + k: v for k, v in full.state_dict().items()})
    sd["visual_encoder.cls_token"] = torch.zeros(1, 1, 768)               # a key of the other tower: dropped on load
    model.load_state_dict(sd, strict=True)
    model.cuda()
    B, geom = 1, S.VLNCE_R2R
    mem = GridMemoryBatch(B, geom, max_steps=2, device="cuda")
    imgs = torch.cat([GG.clip_images("full").cuda()] * 6)                 # 12 views: the two fixture images, repeated
    tok = model.encode_observation(imgs, mem)
    assert float((tok[:2].cpu() - want).abs().max()) < 2e-4
    rs = np.random.RandomState(0)
    ob = S.make_observations(rs, geom, 1, with_feats=False)[0]
    mem.step(ob["depth"].reshape(1, -1), None, [(ob["x"], ob["y"])], [ob["heading"]])
    torch.cuda.synchronize()
    assert int(mem.n_pts_host[0]) == 12 * 49
    assert torch.equal(mem.slab[0, :49].cpu(), tok[0, 1:].to(torch.float16).cpu())
    assert int((mem.cell_id[0, :588] >= 0).sum()) > 0                     # the appended observation was binned
