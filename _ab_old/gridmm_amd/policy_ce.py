"""VLN-CE policy shell: the `GridMap.forward(mode=..., ...)` surface of the reference's habitat policy network.

Reference: /root/reference/VLN_CE/vlnce_baselines/models/Policy_ViewSelection_GridMap.py
  GridMap.forward(mode, ...)                :249-625   keyword signature kept verbatim
    mode 'language'                         :262-267   -> vln_bert('language', (ids, masks))
    mode 'waypoint'                         :269-498   waypoint predictor + habitat observations + ResNet / CLIP / ViT
                                                       towers + getGlobalMap: simulator / perception glue, NOT here --
                                                       its device-side parts are GridMemoryBatch.step (getGlobalMap)
                                                       and vilmodel_ce.encode_observation (the CLIP tower)
    mode 'navigation'                       :500-625   panorama encoding, trajectory bookkeeping (visited positions,
                                                       their mean panorama embeddings, relative-pose features),
                                                       vln_bert('navigation', tuple), candidate rotation of the logits
  episode state is plain attributes the trainer sets / pops from outside (ss_trainer_GridMap.py:236-254, 430-450):
  positions, headings, start_positions, action_step, traj_map, traj_embeds.

What changes under the same surface: `vln_bert` is gridmm_amd.vilmodel_ce.GlocalTextPathNavCMT (HIP kernels); the grid
memory may be handed over as `grid_memory=` (a GridMemoryBatch: device-resident slab + per-cell lists) instead of
batch_grid_fts / batch_map_index / batch_gridmap_pos_fts; visited-node embeddings stay on the device (the reference
parks them on the CPU and re-uploads them every step, :518, 560).  Position tuples are (x, z, y) as habitat gives them
(vlnce_baselines/models/utils.py:125-152).  Pinned by tests/golden/policy_ce_nav.npz (the imported reference driven
over three steps).
"""
import math

import numpy as np
import torch

from .graph_utils import angle_features

LIMITS = {"R2R": (25.0, 20.0), "RxR": (40.0, 30.0)}          # MAX_DIST, MAX_STEP per dataset (:273-289)


def rel_pose(a, b, base_heading=0.0, base_elevation=0.0):
    """heading / elevation / distance of b seen from a, both (x, z, y); (0, 0, 0) for coincident points
    (vlnce_baselines/models/utils.py:125-144)."""
    dx, dz, dy = b[0] - a[0], b[1] - a[1], b[2] - a[2]
    if dx == 0 and dz == 0 and dy == 0:
        return 0.0, 0.0, 0.0
    flat = max(math.sqrt(dx * dx + dy * dy), 1e-8)
    full = max(math.sqrt(dx * dx + dy * dy + dz * dz), 1e-8)
    heading = math.asin(dx / flat)
    if b[2] < a[2]:
        heading = math.pi - heading
    return heading - base_heading, math.asin(dz / full) - base_elevation, full


def _pose_rows(rows, max_dist, max_step):
    """[(heading, elevation, line, path, hops)] -> (n, 7) float32: sin/cos heading, sin/cos elevation, scaled distances."""
    r = np.asarray(rows, dtype=np.float64).reshape(-1, 5)
    ang = r[:, :2].astype(np.float32)
    rel = (r[:, 2:] / np.array([max_dist, max_dist, max_step])).astype(np.float32)
    return np.concatenate([angle_features(ang[:, 0], ang[:, 1]), rel], 1)


class GridMap:
    def __init__(self, vln_bert, batch_size=1, dataset="R2R", device="cuda"):
        self.vln_bert, self.dataset, self.device = vln_bert, dataset, torch.device(device)
        self.headings = [0.0] * batch_size
        self.positions = None
        self.start_positions = None
        self.start_headings = None
        self.traj_embeds = [[] for _ in range(batch_size)]     # per episode: (1, H) mean panorama embedding per visit
        self.traj_map = [[] for _ in range(batch_size)]        # per episode: (position, distance from the previous one)
        self.action_step = 0

    def __call__(self, *a, **kw):
        return self.forward(*a, **kw)

    def forward(self, mode=None, waypoint_predictor=None, observations=None, lang_idx_tokens=None, lang_masks=None,
                lang_feats=None, lang_token_type_ids=None, headings=None, positions=None, cand_rgb=None, cand_depth=None,
                cand_direction=None, cand_mask=None, candidate_lengths=None, batch_angles=None, batch_distances=None,
                masks=None, batch_view_img_fts=None, batch_loc_fts=None, batch_nav_types=None, batch_view_lens=None,
                batch_grid_fts=None, batch_map_index=None, batch_gridmap_pos_fts=None, in_train=True, grid_memory=None):
        if mode == "language":
            return self.vln_bert("language", (lang_idx_tokens, lang_masks))
        if mode == "navigation":
            return self._navigation(lang_feats, lang_masks, positions, candidate_lengths, batch_angles, batch_distances,
                                    batch_view_img_fts, batch_loc_fts, batch_nav_types, batch_view_lens, batch_grid_fts,
                                    batch_map_index, batch_gridmap_pos_fts, grid_memory)
        if mode == "waypoint":
            raise NotImplementedError("mode 'waypoint' is simulator / perception glue (waypoint predictor, habitat "
                                      "observations, ResNet / ViT towers); its device-side pieces are "
                                      "GridMemoryBatch.step and GlocalTextPathNavCMT.encode_observation")
        raise NotImplementedError("wrong mode: %s" % mode)

    # ---- mode 'navigation' (:500-625)
    def _navigation(self, lang_feats, lang_masks, positions, cand_lens, angles, distances, view_img_fts, loc_fts,
                    nav_types, view_lens, grid_fts, map_index, gridmap_pos_fts, grid_memory):
        max_dist, max_step = LIMITS[self.dataset]
        dev = view_img_fts.device
        B = view_img_fts.shape[0]
        pano, pano_masks = self.vln_bert("panorama", (view_img_fts, loc_fts, nav_types, view_lens))
        m = pano_masks.unsqueeze(2).to(pano.dtype)
        mean_pano = (pano * m).sum(1) / m.sum(1)
        vp_img = torch.cat([torch.zeros_like(pano[:, :1]), pano], 1)
        for i in range(B):                                       # this visit joins the episode's trajectory
            prev = self.traj_map[i][-1][0] if self.traj_map[i] else None
            step_len = 0
            if prev is not None:                                 # (x, z, y): summed in the reference's x, y, z order
                dx, dz, dy = positions[i][0] - prev[0], positions[i][1] - prev[1], positions[i][2] - prev[2]
                step_len = math.sqrt(dx * dx + dy * dy + dz * dz)
            self.traj_embeds[i].append(mean_pano[i:i + 1])
            self.traj_map[i].append((positions[i], step_len))

        H = pano.shape[-1]
        node_embeds, node_steps, node_pos, vp_pos, node_lens = [], [], [], [], []
        for i in range(B):
            nc, visits = int(cand_lens[i]) - 1, self.traj_map[i]
            here, facing = self.positions[i], self.headings[i]
            # map nodes: [stop] | this step's candidates (ghost nodes at their polar offsets) | visited, newest first
            cand_rows = [(float(angles[i][j]), 0.0, float(distances[i][j]), float(distances[i][j]), 1.0) for j in range(nc)]
            rows = [(0.0, 0.0, 0.0, 0.0, 0.0)] + cand_rows
            steps = [0] + [len(visits) + 1] * nc
            embeds = [pano.new_zeros(1, H), pano[i, :nc]]
            walked = 0.0
            for j in range(len(visits) - 1, -1, -1):
                hd, el, line = rel_pose(here, visits[j][0], base_heading=facing)
                rows.append((hd, el, line, walked, float(self.action_step - j - 1)))
                walked += visits[j][1]
                steps.append(j + 1)
                embeds.append(self.traj_embeds[i][j].to(dev))
            node_pos.append(torch.from_numpy(_pose_rows(rows, max_dist, max_step)))
            node_steps.append(torch.tensor(steps, dtype=torch.long))
            node_embeds.append(torch.cat(embeds, 0))
            node_lens.append(node_embeds[-1].shape[0])
            # local branch: every token carries the pose of the start position (:590-601), candidates add their own
            hd, el, line = rel_pose(here, self.start_positions[i], base_heading=facing)
            p = np.zeros((vp_img.shape[1], 14), dtype=np.float32)
            p[:, :7] = _pose_rows([(hd, el, line, walked, float(self.action_step))], max_dist, max_step)
            if nc:
                p[1:nc + 1, 7:] = _pose_rows(cand_rows, max_dist, max_step)
            vp_pos.append(torch.from_numpy(p))

        G = max(node_lens)
        pad = lambda ts, fill=0: torch.stack([torch.cat([t, t.new_full((G - t.shape[0],) + tuple(t.shape[1:]), fill)]) for t in ts])
        node_lens_t = torch.tensor(node_lens)
        node_masks = (torch.arange(G)[None] < node_lens_t[:, None]).to(dev)
        vp_nav_masks = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=dev), nav_types == 1], 1)
        vp_masks = torch.arange(int(view_lens.max()) + 1, device=dev)[None] < (view_lens + 1)[:, None]
        batch = (lang_feats, lang_masks, pad(node_embeds), pad(node_steps).to(dev), pad(node_pos).to(dev), node_masks,
                 vp_img, torch.stack(vp_pos).to(dev), vp_masks, vp_nav_masks, grid_fts, map_index, gridmap_pos_fts, cand_lens)
        logits = self.vln_bert("navigation", batch, grid_memory=grid_memory) if grid_memory is not None \
            else self.vln_bert("navigation", batch)
        for b in range(B):                                       # [stop] moves behind the candidates (:618-619)
            n = int(cand_lens[b])
            logits[b, :n] = torch.cat((logits[b, 1:n], logits[b, 0:1]), 0)
        return logits
