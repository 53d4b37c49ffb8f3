"""Seeded synthetic episodes for benches, smoke and parity tests.

There is no simulator, dataset or checkpoint on the build/GPU boxes, so every
measurement uses synthetic observations with the shapes the reference produces:

  * observation for the grid memory: sampled depth (n_views, P*P) uint16 ~ U[0,20000)
    with 10 % zeros (invalid), CLIP patch tokens (n_views*P*P, D_in) fp16 ~ N(0, s),
    pose random walk (step ~ U[1,3] m), heading in {k*30 deg}
    (reference inputs: map_nav_src/r2r/env.py:278-296)
  * navigation inputs with the keys/shapes of map_nav_src/r2r/agent.py:163-169,199-205,330-333
"""
import math

import numpy as np
import torch


class GridGeometry:
    """Shape of one observation slab (native 12x7x7x768; BASELINE 36x14x14x512)."""

    def __init__(self, n_views=12, patches=7, feat_dim=768, depth_div=4000.0,
                 tan_half_fov=math.tan(math.pi / 6), vlnce=False, max_dist=30.0):
        self.n_views, self.patches, self.feat_dim = n_views, patches, feat_dim
        self.depth_div, self.tan_half_fov = depth_div, tan_half_fov
        # VLN-CE twin (Policy_ViewSelection_GridMap.py:632-641, 689-825): float32 depth in metres, view angles
        # relative to the heading, mirrored y, rotation by pi, MAX_DIST 25 / 40
        self.vlnce, self.max_dist = vlnce, max_dist

    @property
    def pts_per_obs(self):
        return self.n_views * self.patches * self.patches


NATIVE = GridGeometry()
BASELINE = GridGeometry(36, 14, 512)
VLNCE_R2R = GridGeometry(depth_div=1.0, tan_half_fov=math.tan(math.pi / 4.), vlnce=True, max_dist=25.0)
VLNCE_RXR = GridGeometry(depth_div=1.0, tan_half_fov=math.tan(math.pi * 79. / 360.), vlnce=True, max_dist=40.0)


def make_observations(rs, geom, steps, feat_scale=1.0, zero_frac=0.1):
    """One episode's observation sequence: list of dicts(depth, feats, x, y, heading)."""
    obs = []
    x, y = float(rs.uniform(-5, 5)), float(rs.uniform(-5, 5))
    for _ in range(steps):
        d = rs.randint(0, 20000, size=(geom.n_views, geom.patches ** 2)).astype(np.uint16)
        d[rs.rand(*d.shape) < zero_frac] = 0
        f = (rs.standard_normal((geom.pts_per_obs, geom.feat_dim)) * feat_scale).astype(np.float16)
        obs.append(dict(depth=d, feats=f, x=x, y=y, heading=float(rs.randint(0, 12)) * math.pi / 6))
        r, a = rs.uniform(1, 3), rs.uniform(0, 2 * math.pi)
        x, y = float(x + r * math.cos(a)), float(y + r * math.sin(a))
    return obs


def make_nav_batch(rs, B, L=80, G=20, n_visited=6, V1=37, n_cand=4, H=768,
                   min_len=30, ragged_gmap=True, with_obj=False):
    """Navigation-mode inputs EXCEPT the grid memory (grid_fts / grid_map / gridmap_pos_fts).

    Returns a dict of CPU tensors + python vpid lists with the reference's key names.
    """
    txt_lens = rs.randint(min_len, L + 1, size=B)
    txt_lens[rs.randint(B)] = L
    txt_masks = torch.from_numpy(np.arange(L)[None, :] < txt_lens[:, None])
    txt_embeds = torch.from_numpy(rs.standard_normal((B, L, H)).astype(np.float32))

    gmap_lens = np.full(B, G)
    if ragged_gmap and B > 1:
        gmap_lens = rs.randint(max(n_visited + 3, G - 6), G + 1, size=B)
        gmap_lens[rs.randint(B)] = G
    gmap_masks = torch.from_numpy(np.arange(G)[None, :] < gmap_lens[:, None])
    gmap_img = rs.standard_normal((B, G, H)).astype(np.float32)
    gmap_img[:, 0] = 0  # [stop] token row is zeros (agent.py:133-135)
    gmap_img *= gmap_masks.numpy()[:, :, None]
    gmap_step_ids = np.zeros((B, G), np.int64)
    gmap_visited = np.zeros((B, G), bool)
    gmap_vpids, vp_cand_vpids = [], []
    nav_types = np.zeros((B, V1 - 1), np.int64)
    for b in range(B):
        n = int(gmap_lens[b])
        ids = ["vp%03d_%d" % (b, k) for k in range(1, n)]
        gmap_vpids.append([None] + ids)
        gmap_visited[b, 1:1 + n_visited] = True
        gmap_step_ids[b, 1:1 + n_visited] = np.arange(1, n_visited + 1)
        # candidates: the previous (visited) node + unvisited frontier nodes
        cands = [ids[n_visited - 2]] + list(rs.choice(ids[n_visited:], size=n_cand - 1, replace=False))
        vp_cand_vpids.append([None] + cands)
        nav_types[b, :n_cand] = 1
    gmap_pos_fts = rs.uniform(-1, 1, size=(B, G, 7)).astype(np.float32) * gmap_masks.numpy()[:, :, None]
    vp_pos_fts = rs.uniform(-1, 1, size=(B, V1, 14)).astype(np.float32)
    vp_pos_fts[:, n_cand + 1:, 7:] = 0
    vp_pos_fts[:, 0, 7:] = 0
    vp_img = rs.standard_normal((B, V1, H)).astype(np.float32)
    vp_img[:, 0] = 0
    vp_nav_masks = np.concatenate([np.ones((B, 1), bool), nav_types == 1], 1)
    batch = {
        "txt_embeds": txt_embeds, "txt_masks": txt_masks,
        "gmap_img_embeds": torch.from_numpy(gmap_img),
        "gmap_step_ids": torch.from_numpy(gmap_step_ids),
        "gmap_pos_fts": torch.from_numpy(gmap_pos_fts),
        "gmap_masks": gmap_masks,
        "gmap_pair_dists": torch.zeros(B, G, G),
        "gmap_visited_masks": torch.from_numpy(gmap_visited),
        "gmap_vpids": gmap_vpids,
        "vp_img_embeds": torch.from_numpy(vp_img),
        "vp_pos_fts": torch.from_numpy(vp_pos_fts),
        "vp_masks": torch.ones(B, V1, dtype=torch.bool),
        "vp_nav_masks": torch.from_numpy(vp_nav_masks),
        "vp_obj_masks": None,
        "vp_cand_vpids": vp_cand_vpids,
    }
    if with_obj:
        m = np.zeros((B, V1), bool)
        m[:, V1 - 5:] = True
        batch["vp_obj_masks"] = torch.from_numpy(m)
    return batch


def batch_to(batch, device):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v.to(device)
        elif isinstance(v, list) and len(v) and torch.is_tensor(v[0]):
            out[k] = [t.to(device) for t in v]
        else:
            out[k] = v
    return out
