"""The reference's model wrapper around GlocalTextPathNavCMT (map_nav_src/models/model.py:12-39).

`VLNBert.forward(mode, batch)` is what the agent loop calls (r2r/agent.py:276-343): 'language' and 'navigation' pass
through; 'panorama' first applies the ENVIRONMENT feature dropout -- nn.Dropout(args.feat_dropout), 0.4 in
scripts/run_r2r.sh -- to the view (and object) image features, active in train() only.  The arithmetic of the three
modes runs in libgridmm_hip.so (vilmodel.py); this class only holds the wrapped model and the dropout.
"""
import collections

import torch.nn.functional as F
from torch import nn

from .vilmodel import GlocalTextPathNavCMT, default_config


class VLNBert(nn.Module):
    def __init__(self, args, config=None, vln_bert=None):
        """args: namespace with feat_dropout (r2r/parser.py).  vln_bert: an existing GlocalTextPathNavCMT (e.g. with a
        checkpoint loaded); else one is built from `config` (default_config() when None) -- the reference builds it from
        the BERT / LXMERT initialisation of vlnbert_init.py:5-58, which is checkpoint plumbing outside the hot path."""
        super().__init__()
        self.args = args
        self.vln_bert = vln_bert if vln_bert is not None else GlocalTextPathNavCMT(config or default_config())
        self.feat_dropout = float(getattr(args, "feat_dropout", 0.0))
        self.train(self.vln_bert.training)          # wrapping an eval() model must not switch the feature dropout on

    def drop_env(self, x):
        return F.dropout(x, self.feat_dropout, self.training) if (self.training and self.feat_dropout > 0) else x

    def forward(self, mode, batch):
        batch = collections.defaultdict(lambda: None, batch)
        if mode == "language":
            return self.vln_bert(mode, batch)
        if mode == "panorama":
            batch["view_img_fts"] = self.drop_env(batch["view_img_fts"])
            if batch.get("obj_img_fts") is not None:
                batch["obj_img_fts"] = self.drop_env(batch["obj_img_fts"])
            return self.vln_bert(mode, batch)
        if mode == "navigation":
            return self.vln_bert(mode, batch)
        raise NotImplementedError("wrong mode: %s" % mode)


class Critic(nn.Module):
    """The value head of the reference's (vestigial) A2C branch (map_nav_src/models/model.py:43-55): kept so that agent
    checkpoints -- which store a 'critic' entry beside 'vln_bert' (r2r/agent_base.py:213-228) -- load and save unchanged.
    No released script trains it (rollout(train_rl=...) is never called with True); plain torch modules, off the hot path."""

    def __init__(self, args):
        super().__init__()
        self.state2value = nn.Sequential(nn.Linear(768, 512), nn.ReLU(), nn.Dropout(float(getattr(args, "dropout", 0.5))),
                                         nn.Linear(512, 1))

    def forward(self, state):
        return self.state2value(state).squeeze()
