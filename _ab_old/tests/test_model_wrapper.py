"""model.VLNBert: the reference's wrapper (map_nav_src/models/model.py:12-39) -- environment feature dropout on the
'panorama' image features in train() only; other modes pass through untouched."""
from types import SimpleNamespace

import torch
from torch import nn

from gridmm_amd.model import VLNBert


class _Probe(nn.Module):
    def __init__(self):
        super().__init__()
        self.seen = []

    def forward(self, mode, batch):
        self.seen.append((mode, batch))
        return mode


def test_feature_dropout_only_in_train_mode_and_only_on_panorama():
    torch.manual_seed(0)
    m = VLNBert(SimpleNamespace(feat_dropout=0.4), vln_bert=_Probe())
    x = torch.ones(4, 36, 768)
    m.eval()
    m("panorama", {"view_img_fts": x, "obj_img_fts": x.clone()})
    assert torch.equal(m.vln_bert.seen[-1][1]["view_img_fts"], x)
    m.train()
    m("panorama", {"view_img_fts": x, "obj_img_fts": x.clone()})
    v, o = m.vln_bert.seen[-1][1]["view_img_fts"], m.vln_bert.seen[-1][1]["obj_img_fts"]
    for t in (v, o):
        kept = t != 0
        assert abs(float(kept.float().mean()) - 0.6) < 0.01                  # p = 0.4 dropped
        assert torch.allclose(t[kept], torch.full_like(t[kept], 1 / 0.6))    # survivors scaled by 1 / (1 - p)
    assert not torch.equal(v, o)                                             # independent masks
    m("panorama", {"view_img_fts": x})                                       # no object features: key stays None
    assert m.vln_bert.seen[-1][1]["obj_img_fts"] is None
    m("navigation", {"txt_embeds": x})
    assert torch.equal(m.vln_bert.seen[-1][1]["txt_embeds"], x)
    assert m.vln_bert.seen[-1][1]["grid_fts"] is None                        # defaultdict(None), as the reference's
    assert m("language", {"txt_ids": x}) == "language"


def test_agent_wraps_a_bare_model():
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    cfg = default_config(num_l_layers=1, num_pano_layers=1, num_x_layers=1, intermediate_size=64, vocab_size=50)
    core = GlocalTextPathNavCMT(cfg)
    agent = GMapNavAgent(default_args(), env=None, vln_bert=core, device="cpu")
    assert isinstance(agent.vln_bert, VLNBert) and agent.vln_bert.vln_bert is core
    assert agent.vln_bert.training == core.training
    assert not GMapNavAgent(default_args(), env=None, vln_bert=GlocalTextPathNavCMT(cfg).eval(), device="cpu").vln_bert.training
    assert all(k.startswith("vln_bert.") for k in agent.vln_bert.state_dict())   # reference checkpoint key prefix
    probe = _Probe()
    assert GMapNavAgent(default_args(), env=None, vln_bert=probe, device="cpu").vln_bert is probe


def test_agent_checkpoint_round_trip_in_the_reference_format(tmp_path):
    """save() writes {'vln_bert': {epoch, state_dict, optimizer}, 'critic': {...}} (r2r/agent_base.py:213-228); load()
    restores it, strips a DDP 'module.' prefix, skips keys the model does not have, returns the epoch."""
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    cfg = default_config(num_l_layers=1, num_pano_layers=1, num_x_layers=1, intermediate_size=64, vocab_size=50)
    agent = GMapNavAgent(default_args(optim="adam"), env=None, vln_bert=GlocalTextPathNavCMT(cfg), device="cpu")
    path = str(tmp_path / "ckpt" / "latest_dict")
    agent.save(6, path)
    st = torch.load(path, map_location="cpu")
    assert set(st) == {"vln_bert", "critic"} and set(st["vln_bert"]) == {"epoch", "state_dict", "optimizer"}
    assert st["vln_bert"]["epoch"] == 7 and all(k.startswith("vln_bert.") for k in st["vln_bert"]["state_dict"])
    assert set(st["critic"]["state_dict"]) == {"state2value.0.weight", "state2value.0.bias", "state2value.3.weight", "state2value.3.bias"}
    want = {k: v.clone() for k, v in agent.vln_bert.state_dict().items()}
    with torch.no_grad():
        for p in agent.vln_bert.parameters():
            p.add_(1.0)
    assert agent.load(path) == 6
    assert all(torch.equal(v, want[k]) for k, v in agent.vln_bert.state_dict().items())
    # a checkpoint written from a DistributedDataParallel wrapper, with a key this model does not have
    st["vln_bert"]["state_dict"] = {"module." + k: v for k, v in st["vln_bert"]["state_dict"].items()}
    st["vln_bert"]["state_dict"]["module.vln_bert.not_here.weight"] = torch.zeros(3)
    torch.save(st, path)
    with torch.no_grad():
        for p in agent.vln_bert.parameters():
            p.mul_(0.0)
    assert agent.load(path) == 6
    assert all(torch.equal(v, want[k]) for k, v in agent.vln_bert.state_dict().items())
