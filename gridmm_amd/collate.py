"""Batched host-side collation for GMapNavAgent.rollout: the same `pano_inputs` / `nav_inputs` dictionaries as
agent.py's per-episode restatement of the reference (map_nav_src/r2r/agent.py:51-205), built with one vectorised pass
over the batch instead of per-node tensor operations.

What the reference does per step and episode: stacks 36 view features one by one, keeps one tensor per graph node for
the running mean of its embeddings (models/graph_utils.py:125-137) and stacks them again every step, and calls
get_pos_fts three times (graph nodes, candidates, start node).  At B = 32 that is ~2 000 tiny tensor operations per
step -- 34 ms of host time around a 4 ms model call (bench.py `rollout`, round-3 first measurement).  Here:

  * node embeddings live in ONE device tensor per rollout (B, node slots, H) of running sums + host-side counts;
    a step applies all overwrites with one index_put and all accumulations with another, and reads
    `gmap_img_embeds` back with one gather (count 1 -> multiplied by 1.0: the stored row, exactly);
  * position features of the graph nodes, the candidates and the start node of all episodes go through one
    `graph_utils.batched_pos_features` call (same arithmetic and precisions as TopoMap.pos_features);
  * panorama inputs are assembled into padded numpy arrays with one fancy-index per episode and uploaded once.

agent.GMapNavAgent uses it by default (`fast_collate`); tests/test_agent_loop.py checks it against the per-episode
restatement on the same observations, key by key.
"""
import numpy as np
import torch

from .graph_utils import UNREACHABLE, batched_pos_features


class NavCollator:
    def __init__(self, args, device, node_slots=48, node_buckets=None, view_buckets=None):
        """node_buckets / view_buckets: ascending sizes to which the graph-node axis G / the view axis V are padded
        (masked rows: the model's outputs on the real rows do not change) so that the per-step shapes come from a small
        set -- what graph.NavigationGraphs keys its captured graphs by.  None: the reference's shapes (max over the batch)."""
        self.args, self.device = args, torch.device(device)
        self.node_slots = node_slots
        self.node_buckets, self.view_buckets = node_buckets, view_buckets
        self.pool = None              # (B, slots, H) running sums; slot 0 stays zero (stop token / padding)
        self.cnt = None               # (B, slots) host counts

    def reset(self, batch_size):
        self.pool = None
        self.cnt = np.zeros((batch_size, self.node_slots), dtype=np.int32)

    @staticmethod
    def _bucket(n, buckets):
        for b in buckets or ():
            if n <= b:
                return int(b)
        return n

    def _dev(self, a):
        return torch.from_numpy(a).to(self.device)

    # ---- agent.py:51-94 ---------------------------------------------------------------------------
    def panorama(self, obs):
        fs, B = self.args.image_feat_size, len(obs)
        rows, cands_all, n_cand = [], [], []
        for ob in obs:
            cands = ob["candidate"]
            used = {int(cc["pointId"]) for cc in cands}
            rest = [k for k in range(36) if k not in used]
            feat = ob["feature"]
            parts = [cc["feature"][None] for cc in cands]
            parts.append(np.asarray(feat)[rest] if isinstance(feat, np.ndarray) else np.stack([feat[k] for k in rest]))
            rows.append(np.concatenate(parts, 0))
            cands_all.append([cc["viewpointId"] for cc in cands])
            n_cand.append(len(cands))
        lens = np.array([r.shape[0] for r in rows], dtype=np.int64)
        self._view_lens = lens
        V, W = self._bucket(int(lens.max()), self.view_buckets), rows[0].shape[1]
        full = np.zeros((B, V, W + 3), dtype=np.float32)
        types = np.zeros((B, V), dtype=np.int64)
        for i, r in enumerate(rows):
            full[i, :lens[i], :W] = r
            full[i, :lens[i], W:] = 1.0           # the constant "box" columns of valid rows
            types[i, :n_cand[i]] = 1
        return {
            "view_img_fts": self._dev(np.ascontiguousarray(full[:, :, :fs])),
            "loc_fts": self._dev(np.ascontiguousarray(full[:, :, fs:])),
            "nav_types": self._dev(types), "view_lens": self._dev(lens),
            "cand_vpids": cands_all, "obj_img_fts": None, "obj_lens": None,
        }

    # ---- agent.py:300-311 (graph node embeddings) ---------------------------------------------------
    def _grow(self, need):
        slots = self.cnt.shape[1]
        while slots <= need:
            slots *= 2
        cnt = np.zeros((self.cnt.shape[0], slots), dtype=np.int32)
        cnt[:, :self.cnt.shape[1]] = self.cnt
        self.cnt = cnt
        if self.pool is not None:
            pool = self.pool.new_zeros(self.pool.shape[0], slots, self.pool.shape[2])
            pool[:, :self.pool.shape[1]] = self.pool
            self.pool = pool

    def _apply(self, src, sets, adds):
        for ops, accumulate in ((sets, False), (adds, True)):
            if not ops:
                continue
            ix = self._dev(np.asarray(ops, dtype=np.int64).T.copy())
            vals = src[ix[0], ix[2]]
            if torch.is_grad_enabled() and (vals.requires_grad or self.pool.requires_grad):
                self.pool = self.pool.index_put((ix[0], ix[1]), vals, accumulate=accumulate)
            else:
                self.pool.index_put_((ix[0], ix[1]), vals, accumulate=accumulate)

    def update_embeddings(self, obs, gmaps, ended, pano_embeds, pano_masks, cand_vpids):
        """Current node := mean of its panorama (overwrite); every unvisited candidate += its view embedding."""
        avg = torch.sum(pano_embeds * pano_masks.unsqueeze(2), 1) / torch.sum(pano_masks, 1, keepdim=True)
        src = torch.cat([avg.unsqueeze(1), pano_embeds], 1)               # row 0: the panorama mean
        if self.pool is None:
            self.pool = src.new_zeros(src.shape[0], self.cnt.shape[1], src.shape[2])
        sets, adds, touched = [], [], set()
        for i, g in enumerate(gmaps):
            if ended[i]:
                continue
            if g.n + 1 >= self.cnt.shape[1]:
                self._grow(g.n + 1)
            items = [(g.index(obs[i]["viewpoint"]) + 1, 0, True)]
            items += [(g.index(c) + 1, j + 1, False) for j, c in enumerate(cand_vpids[i]) if not g.visited(c)]
            for slot, row, overwrite in items:
                if (i, slot) in touched:                                   # same node twice in one step: keep the order
                    self._apply(src, sets, adds)
                    sets, adds, touched = [], [], set()
                touched.add((i, slot))
                if overwrite or self.cnt[i, slot] == 0:
                    sets.append((i, slot, row))
                    self.cnt[i, slot] = 1
                else:
                    adds.append((i, slot, row))
                    self.cnt[i, slot] += 1
        self._apply(src, sets, adds)

    # ---- agent.py:96-205 ----------------------------------------------------------------------------
    def navigation(self, obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types, grid_memory=None):
        a, B = self.args, len(obs)
        recs, deltas, graphs, hopss, base_h, base_e = [], [], [], [], [], []
        for i, g in enumerate(gmaps):
            cur = g.index(obs[i]["viewpoint"])
            n = g.n
            if a.act_visited_nodes:
                vis = np.array([cur], dtype=np.int64)
                unv = np.array([k for k in range(n) if k != cur], dtype=np.int64)
            else:
                seen = g.seen[:n]
                vis, unv = np.flatnonzero(seen), np.flatnonzero(~seen)
            ids = np.concatenate([vis, unv]) if a.enc_full_graph else unv
            n_vis = len(vis) if a.enc_full_graph else 0
            cid = np.fromiter((g.index(c) for c in cand_vpids[i]), dtype=np.int64, count=len(cand_vpids[i]))
            tgt = np.concatenate([ids, cid, np.array([g.index(g.start_vp)], dtype=np.int64)])
            d, gr, hp = g.relative_geometry(cur, tgt)
            deltas.append(d)
            graphs.append(gr)
            hopss.append(hp)
            base_h.append(np.full(len(tgt), float(obs[i]["heading"])))
            base_e.append(np.full(len(tgt), float(obs[i]["elevation"])))
            recs.append((g, ids, n_vis, len(unv), len(cid)))
        feats = batched_pos_features(np.concatenate(deltas), np.concatenate(base_h), np.concatenate(base_e),
                                     np.concatenate(graphs), np.concatenate(hopss))
        F = feats.shape[1]
        G = self._bucket(1 + max(len(r[1]) for r in recs), self.node_buckets)
        V1 = pano_embeds.shape[1] + 1
        cand_of_node = np.full((B, G), -2, dtype=np.int32)     # integer form of the vpid-keyed fusion loops
        cand_visited = np.zeros((B, V1), dtype=np.uint8)       # (vilmodel.py:884-899; GlocalTextPathNavCMT._fusion_index_maps)
        gpos = np.zeros((B, G, F), dtype=np.float32)
        pair = np.zeros((B, G, G), dtype=np.float32)
        steps = np.zeros((B, G), dtype=np.int64)
        visited = np.zeros((B, G), dtype=bool)
        slot = np.zeros((B, G), dtype=np.int64)
        inv = np.ones((B, G), dtype=np.float32)
        vpos = np.zeros((B, V1, 2 * F), dtype=np.float32)
        lens = np.zeros(B, dtype=np.int64)
        stop_row = np.array([0, 1, 0, 1] + [0] * (F - 4), dtype=np.float32)  # the stop token: zero angles and distances
        vpids, no_vp_left, o = [], [], 0
        for i, (g, ids, n_vis, n_unv, nc) in enumerate(recs):
            m = len(ids)
            lens[i] = m + 1
            gpos[i, 0] = stop_row
            gpos[i, 1:m + 1] = feats[o:o + m]
            vpos[i, 1:nc + 1, F:] = feats[o + m:o + m + nc]
            vpos[i, :, :F] = feats[o + m + nc]
            o += m + nc + 1
            if m:
                sub = g.dist[np.ix_(ids, ids)]
                sub = np.where(np.isfinite(sub), sub, float(UNREACHABLE))
                np.fill_diagonal(sub, 0.0)
                pair[i, 1:m + 1, 1:m + 1] = sub
            names = [g.names[k] for k in ids]
            steps[i, 1:m + 1] = [g.step_id.get(v, 0) for v in names]
            visited[i, 1:1 + n_vis] = True
            slot[i, 1:m + 1] = ids + 1
            c = self.cnt[i, ids + 1]
            if (c == 0).any():
                raise KeyError("graph node without an embedding: %s" % names[int(np.flatnonzero(c == 0)[0])])
            inv[i, 1:m + 1] = np.float32(1.0) / c.astype(np.float32)
            vpids.append([None] + names)
            no_vp_left.append(n_unv == 0)
            seen_names = set(names[:n_vis])
            col = {}
            for j, cv in enumerate(cand_vpids[i]):
                if cv in seen_names:
                    cand_visited[i, j + 1] = 1
                else:
                    col[cv] = j + 1
            for j in range(n_vis, m):
                cand_of_node[i, j + 1] = col.get(names[j], -1)
        slot_d, inv_d = self._dev(slot), self._dev(inv)
        rows = torch.arange(B, device=self.device).unsqueeze(1)
        gmap_img = self.pool[rows, slot_d] * inv_d.unsqueeze(2)
        out = {
            "gmap_vpids": vpids, "gmap_img_embeds": gmap_img, "gmap_step_ids": self._dev(steps),
            "gmap_pos_fts": self._dev(gpos), "gmap_visited_masks": self._dev(visited),
            "gmap_pair_dists": self._dev(pair),
            "gmap_masks": self._dev(np.arange(G)[None] < lens[:, None]), "no_vp_left": no_vp_left,
            "fusion_maps": (self._dev(cand_of_node), self._dev(cand_visited)),
        }
        vp_img = torch.cat([torch.zeros_like(pano_embeds[:, :1]), pano_embeds], 1)
        nav_masks = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=self.device), nav_types == 1], 1)
        vl = view_lens + 1
        v_max = V1                    # == max(view_lens) + 1 on the reference's shapes; the padded width with view buckets
        out.update({
            "vp_img_embeds": vp_img, "vp_pos_fts": self._dev(vpos),
            "vp_masks": torch.arange(v_max, device=vl.device).unsqueeze(0) < vl.unsqueeze(1),
            "vp_nav_masks": nav_masks, "vp_cand_vpids": [[None] + x for x in cand_vpids], "vp_obj_masks": None,
        })
        return out
