"""TEST INFRASTRUCTURE: CPU stand-ins with the product interfaces, built on the oracle restatements.

  OracleGridMemory  -- the reset()/step()/as_reference_obs() surface of gridmm_amd.grid_memory.GridMemoryBatch,
                       backed by oracle.gridmap_oracle.GridMemory (NumPy) per episode
  OracleVLNBert     -- a (mode, batch) callable with the reference's forward() contract, backed by
                       oracle.navcmt_oracle (torch, functional over a state_dict)
Used by tests/ to drive gridmm_amd.agent.GMapNavAgent without a GPU, and by gen_golden.py.
"""
import numpy as np
import torch

from . import gridmap_oracle as G
from . import navcmt_oracle as O


class OracleGridMemory:
    slab = None   # no device-resident form: the agent falls back to the reference's list form

    def __init__(self, batch_size, geom=G.NATIVE):
        self.B, self.geom = batch_size, geom
        self.reset()

    def reset(self):
        self.mems = [G.GridMemory(self.geom) for _ in range(self.B)]
        self.last = [None] * self.B

    def step(self, depth, feats, poses, headings, active=None):
        for b in range(self.B):
            if active is not None and not active[b]:
                continue
            d = np.asarray(depth[b]).reshape(self.geom.n_views, -1)
            self.last[b] = self.mems[b].step(d, np.asarray(feats[b]), poses[b][0], poses[b][1], headings[b])

    def as_reference_obs(self):
        return ([torch.from_numpy(r[0]) for r in self.last], [torch.from_numpy(r[1]) for r in self.last],
                torch.from_numpy(np.stack([r[2] for r in self.last])))


class OracleVLNBert:
    def __init__(self, state_dict):
        self.sd = state_dict

    @torch.no_grad()
    def __call__(self, mode, batch):
        if mode == "language":
            return O.forward_text(self.sd, batch["txt_ids"], batch["txt_masks"])
        if mode == "panorama":
            return O.forward_panorama(self.sd, batch["view_img_fts"], batch["loc_fts"], batch["nav_types"],
                                      batch["view_lens"])
        if mode == "navigation":
            return O.forward_navigation(self.sd, batch)
        raise NotImplementedError("wrong mode: %s" % mode)
