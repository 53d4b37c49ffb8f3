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


def make_observations(rs, geom, steps, feat_scale=1.0, zero_frac=0.1, with_feats=True, depth_mode="uniform",
                      inner_frac=0.01):
    """One episode's observation sequence: list of dicts(depth, feats, x, y, heading).  with_feats=False leaves
    feats None (callers that fill the slab on the device) and draws nothing for them.
    depth_mode "uniform": depth ~ U[0, 5 m) -- every one of the 196 cells ends up occupied (SURVEY 8d's generator);
    "ring": walls at one distance (N(3 m, 5 %)) plus a fraction `inner_frac` of nearer points: the map scale rule
    (half_len = 2/3 of the extent, env.py:322-331) clamps the wall points into the border cells, so an episode occupies
    roughly 60-120 cells -- the regime in which the reference's max_cell_num truncation shortens the sequence."""
    obs = []
    x, y = float(rs.uniform(-5, 5)), float(rs.uniform(-5, 5))
    for _ in range(steps):
        if depth_mode == "ring":
            shape = (geom.n_views, geom.patches ** 2)
            r = rs.normal(12000.0, 600.0, size=shape)
            inner = rs.rand(*shape) < inner_frac
            r[inner] = rs.uniform(2400.0, 12000.0, size=int(inner.sum()))
            d = np.clip(r, 1, 65535).astype(np.uint16)
        else:
            d = rs.randint(0, 20000, size=(geom.n_views, geom.patches ** 2)).astype(np.uint16)
        d[rs.rand(*d.shape) < zero_frac] = 0
        f = (rs.standard_normal((geom.pts_per_obs, geom.feat_dim)) * feat_scale).astype(np.float16) if with_feats else None
        obs.append(dict(depth=d, feats=f, x=x, y=y, heading=float(rs.randint(0, 12)) * math.pi / 6))
        r, a = rs.uniform(1, 3), rs.uniform(0, 2 * math.pi)
        x, y = float(x + r * math.cos(a)), float(y + r * math.sin(a))
    return obs


def make_nav_batch(rs, B, L=80, G=20, n_visited=6, V1=37, n_cand=4, H=768,
                   min_len=30, ragged_gmap=True, with_obj=False, n_obj=5):
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
    if with_obj:      # the last n_obj view slots are object tokens (map_nav_src/reverie/env.py:263-372: up to 20 + padding)
        m = np.zeros((B, V1), bool)
        if n_obj == 5:
            m[:, V1 - 5:] = True
        else:         # ragged object counts, the batch maximum = n_obj
            cnt = rs.randint(max(1, n_obj // 2), n_obj + 1, size=B)
            cnt[rs.randint(B)] = n_obj
            for b in range(B):
                m[b, V1 - n_obj:V1 - n_obj + cnt[b]] = True
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


# ------------------------------------------------------------------------------------------------
# pre-training batches (pretrain_src/data/tasks.py collates)
# ------------------------------------------------------------------------------------------------
def make_pretrain_batch(rs, B, task, max_steps=3, L=24, vocab=2000, H=768, image_prob_size=50, n_pts=(200, 588),
                        feat_scale=0.35, views=(36, 33), with_obj=False, obj_feat_size=768, obj_prob_size=50):
    """One collated batch with the keys / dtypes of mlm_collate / mrc_collate / sap_collate (tasks.py:104-141,
    229-275, 334-377) for R2R (no object tokens).  Episodes are random but self-consistent: every global-map node is
    either a visited viewpoint or a candidate seen from one, the last step's candidates define the local branch.

    CPU tensors + python lists; `grid_fts` is a list of (N_b, 768) fp16, `grid_map` a list of (N_b,) int64.
    """
    txt_ids, txt_labels, txt_lens = [], [], []
    view, loc, types, step_lens, view_lens = [], [], [], [], []
    traj_vpids, traj_cand_vpids, gmap_vpids = [], [], []
    gmap_step_ids, gmap_visited, gmap_pos, gmap_dists = [], [], [], []
    vp_pos, grid_fts, grid_map, gpos = [], [], [], []
    g_labels, l_labels, mrc_masks, mrc_probs = [], [], [], []
    objs, obj_lens, obj_labels, obj_mrc_masks, obj_mrc_probs = [], [], [], [], []
    for b in range(B):
        n_tok = int(rs.randint(8, L - 1))
        ids = [101] + rs.randint(1000, vocab, size=n_tok).tolist() + [102]
        lab = [-1] * len(ids)
        if task == "mlm":                                        # random_word (data/common.py): 15 % masked
            for k in range(1, len(ids) - 1):
                if rs.rand() < 0.15:
                    lab[k], ids[k] = ids[k], 103
            if all(x == -1 for x in lab):
                lab[1], ids[1] = ids[1], 103
        txt_ids.append(torch.tensor(ids, dtype=torch.int32))
        txt_labels.append(torch.tensor(lab, dtype=torch.int32))
        txt_lens.append(len(ids))

        T = int(rs.randint(1, max_steps + 1))
        path = ["b%d_v%d" % (b, t) for t in range(T)]
        cands_per_step, seen = [], []
        for t in range(T):
            n_c = int(rs.randint(2, 5))
            c = ["b%d_v%d_c%d" % (b, t, j) for j in range(n_c)]
            if t + 1 < T:
                c[int(rs.randint(n_c))] = path[t + 1]            # the next viewpoint is one of the candidates
            if t > 0:
                c[0 if c[0] != (path[t + 1] if t + 1 < T else None) else 1] = path[t - 1]   # and the way back
            cands_per_step.append(c)
            V = int(views[rs.randint(len(views))])
            n_o = 0
            if with_obj:        # object tokens follow the views (nav_type 2); the last step always has some
                n_o = int(rs.randint(1 if t == T - 1 else 0, 7))
                objs.append(torch.from_numpy(rs.standard_normal((n_o, obj_feat_size)).astype(np.float32)))
                obj_lens.append(n_o)
            view.append(torch.from_numpy(rs.standard_normal((V, H)).astype(np.float32)))
            loc.append(torch.from_numpy(rs.uniform(-1, 1, size=(V + n_o, 7)).astype(np.float32)))
            types.append(torch.tensor([1] * n_c + [0] * (V - n_c) + [2] * n_o, dtype=torch.int32))
            view_lens.append(V)
            for x in c:
                if x not in seen:
                    seen.append(x)
        step_lens.append(np.int32(T))
        traj_vpids.append(path)
        traj_cand_vpids.append(cands_per_step)
        nodes = path + [x for x in seen if x not in path]
        gmap_vpids.append([None] + nodes)
        G = len(nodes) + 1
        visited = [False] + [x in path for x in nodes]
        gmap_visited.append(torch.tensor(visited))
        gmap_step_ids.append(torch.tensor([0] + [path.index(x) + 1 if x in path else 0 for x in nodes], dtype=torch.int32))
        gmap_pos.append(torch.from_numpy(rs.uniform(-1, 1, size=(G, 7)).astype(np.float32)))
        gmap_dists.append(torch.from_numpy(rs.uniform(0, 1, size=(G, G)).astype(np.float32)))
        Vl = view_lens[-1] + 1 + (obj_lens[-1] if with_obj else 0)
        p = rs.uniform(-1, 1, size=(Vl, 14)).astype(np.float32)
        p[len(cands_per_step[-1]) + 1:, 7:] = 0
        vp_pos.append(torch.from_numpy(p))
        N = int(rs.randint(n_pts[0], n_pts[1] + 1))
        grid_fts.append(torch.from_numpy((rs.standard_normal((N, H)) * feat_scale).astype(np.float16)))
        m = rs.randint(-1, 196, size=N).astype(np.int64)
        m[:40] = rs.randint(0, 3, size=40)
        grid_map.append(torch.from_numpy(m))
        gpos.append(torch.from_numpy(rs.uniform(-1, 1, size=(196, 5)).astype(np.float32)))
        # action labels (dataset.py get_act_labels): 0 = stop, else index of the next node / candidate
        unvisited = [j for j in range(1, G) if not visited[j]]
        stop = rs.rand() < 0.3
        gl = 0 if stop else int(unvisited[rs.randint(len(unvisited))])
        node = gmap_vpids[-1][gl]
        ll = 0 if stop else (cands_per_step[-1].index(node) + 1 if node in cands_per_step[-1] else -100)
        if ll == -100:                                           # label must be a current candidate for the local CE
            ll = int(rs.randint(1, len(cands_per_step[-1]) + 1))
            gl = gmap_vpids[-1].index(cands_per_step[-1][ll - 1])
            if visited[gl]:
                gl, ll = 0, 0
        g_labels.append(gl)
        l_labels.append(ll)
        mm = rs.rand(view_lens[-1]) < 0.15
        if not mm.any():
            mm[rs.randint(view_lens[-1])] = True
        mrc_masks.append(torch.from_numpy(mm))
        pr = rs.standard_normal((view_lens[-1], image_prob_size)).astype(np.float32)
        mrc_probs.append(torch.softmax(torch.from_numpy(pr), -1))
        if task == "mrc":                                        # _mask_img_feat: masked views are zeroed
            view[-1] = view[-1].masked_fill(mrc_masks[-1].unsqueeze(-1), 0)
        if with_obj:
            n_o = obj_lens[-1]
            obj_labels.append(int(rs.randint(n_o)))
            om = rs.rand(n_o) < 0.3
            if not om.any():
                om[rs.randint(n_o)] = True
            obj_mrc_masks.append(torch.from_numpy(om))
            obj_mrc_probs.append(torch.softmax(torch.from_numpy(
                rs.standard_normal((n_o, obj_prob_size)).astype(np.float32)), -1))
            if task == "mrc":
                objs[-1] = objs[-1].masked_fill(obj_mrc_masks[-1].unsqueeze(-1), 0)

    pad = torch.nn.utils.rnn.pad_sequence

    def pad_t(ts):
        n = max(t.shape[0] for t in ts)
        out = torch.zeros((len(ts), n) + tuple(ts[0].shape[1:]), dtype=ts[0].dtype)
        for i, t in enumerate(ts):
            out[i, :t.shape[0]] = t
        return out
    Gm = max(len(x) for x in gmap_vpids)
    dists = torch.zeros(B, Gm, Gm)
    for i, d in enumerate(gmap_dists):
        dists[i, :d.shape[0], :d.shape[1]] = d
    batch = {
        "txt_ids": pad(txt_ids, batch_first=True, padding_value=0).to(torch.int32),
        "txt_lens": torch.tensor(txt_lens, dtype=torch.int32),
        "traj_step_lens": step_lens,
        "traj_vp_view_lens": torch.tensor(view_lens, dtype=torch.int32),
        "traj_view_img_fts": pad_t(view), "traj_loc_fts": pad_t(loc),
        "traj_nav_types": pad(types, batch_first=True, padding_value=0).to(torch.int32),
        "traj_vpids": traj_vpids, "traj_cand_vpids": traj_cand_vpids,
        "gmap_vpids": gmap_vpids,
        "gmap_lens": torch.tensor([len(x) for x in gmap_vpids], dtype=torch.int32),
        "gmap_step_ids": pad(gmap_step_ids, batch_first=True, padding_value=0).to(torch.int32),
        "gmap_visited_masks": pad(gmap_visited, batch_first=True, padding_value=False),
        "gmap_pos_fts": pad_t(gmap_pos), "gmap_pair_dists": dists,
        "vp_lens": torch.tensor([x.shape[0] for x in vp_pos], dtype=torch.int32),
        "vp_pos_fts": pad_t(vp_pos),
        "grid_fts": grid_fts, "grid_map": grid_map, "gridmap_pos_fts": torch.stack(gpos, 0),
        "target_patch_id": torch.zeros(B, dtype=torch.int32),
    }
    if with_obj:
        batch["traj_obj_img_fts"] = pad_t(objs)
        batch["traj_vp_obj_lens"] = torch.tensor(obj_lens, dtype=torch.int32)
        if task == "og":
            batch["obj_labels"] = torch.tensor(obj_labels, dtype=torch.int32)
        if task == "mrc":
            batch["vp_obj_mrc_masks"] = pad(obj_mrc_masks, batch_first=True, padding_value=False)
            batch["vp_obj_probs"] = pad_t(obj_mrc_probs)
    if task == "mlm":
        batch["txt_labels"] = pad(txt_labels, batch_first=True, padding_value=-1).to(torch.int32)
    if task == "sap":
        batch["global_act_labels"] = torch.tensor(g_labels, dtype=torch.int32)
        batch["local_act_labels"] = torch.tensor(l_labels, dtype=torch.int32)
    if task == "mrc":
        batch["vp_view_mrc_masks"] = pad(mrc_masks, batch_first=True, padding_value=False)
        batch["vp_view_probs"] = pad_t(mrc_probs)
    return batch
