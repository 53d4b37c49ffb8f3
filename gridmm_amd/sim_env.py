"""SyntheticNavEnv: a simulator-free stand-in for R2RNavBatch/EnvBatch (map_nav_src/r2r/env.py).

There is no MatterSim, no Matterport data and no network on the build/GPU boxes, so the agent loop is
driven by a seeded fake building: a random planar viewpoint graph, per-viewpoint view features, depth and
CLIP patch tokens.  It emits observation dicts with exactly the fields the reference's `_get_obs` emits
(env.py:595-613) and the candidate dicts of `make_candidate` (env.py:507-575), so the agent loop in
gridmm_amd/agent.py sees the same interface it would see in front of MatterSim.

The grid memory is advanced for EVERY episode on EVERY `_get_obs()` call, as EnvBatch.getStates does
(env.py:392-398) -- including episodes that have already stopped.
"""
import math
import zlib

import numpy as np
import torch

from .synthetic import NATIVE


def angle_feature(heading, elevation, angle_feat_size=4):
    """map_nav_src/utils/data.py:114-117."""
    return np.array([math.sin(heading), math.cos(heading), math.sin(elevation), math.cos(elevation)]
                    * (angle_feat_size // 4), dtype=np.float32)


def all_point_angle_features(angle_feat_size=4):
    """get_all_point_angle_feature (utils/data.py:119-147): [base view][view] -> relative angle features."""
    out = np.empty((36, 36, angle_feat_size), np.float32)
    for base in range(36):
        bh, be = (base % 12) * math.radians(30), (base // 12 - 1) * math.radians(30)
        for ix in range(36):
            h, e = (ix % 12) * math.radians(30), (ix // 12 - 1) * math.radians(30)
            out[base, ix] = angle_feature(h - bh, e - be, angle_feat_size)
    return out


def _rs(*key):
    return np.random.RandomState(zlib.crc32(repr(key).encode()) & 0x7FFFFFFF)


class SyntheticScan:
    """One fake building: viewpoints on a plane, k-nearest-neighbour navigation graph (kept connected)."""

    def __init__(self, name, n_vp=24, seed=0):
        rs = _rs("scan", name, seed)
        self.name = name
        self.vps = ["%s_vp%02d" % (name, i) for i in range(n_vp)]
        xy = rs.uniform(-12, 12, size=(n_vp, 2))
        z = rs.uniform(-0.3, 0.3, size=n_vp)
        self.pos = {v: (float(xy[i, 0]), float(xy[i, 1]), float(z[i])) for i, v in enumerate(self.vps)}
        d = np.linalg.norm(xy[:, None] - xy[None], axis=-1)
        adj = {v: set() for v in self.vps}
        for i in range(n_vp):
            for j in np.argsort(d[i])[1:4]:
                adj[self.vps[i]].add(self.vps[j])
                adj[self.vps[j]].add(self.vps[i])
        # connect components (nearest pair between the first component and the rest)
        while True:
            comp = self._component(adj, self.vps[0])
            if len(comp) == n_vp:
                break
            rest = [v for v in self.vps if v not in comp]
            best = min(((d[self.vps.index(a), self.vps.index(b)], a, b) for a in comp for b in rest))
            adj[best[1]].add(best[2])
            adj[best[2]].add(best[1])
        self.adj = {v: sorted(n) for v, n in adj.items()}
        self._floyd(d)

    @staticmethod
    def _component(adj, start):
        seen, todo = {start}, [start]
        while todo:
            for n in adj[todo.pop()]:
                if n not in seen:
                    seen.add(n)
                    todo.append(n)
        return seen

    def _floyd(self, d):
        n = len(self.vps)
        ix = {v: i for i, v in enumerate(self.vps)}
        D = np.full((n, n), np.inf)
        nxt = -np.ones((n, n), int)
        for v in self.vps:
            D[ix[v], ix[v]] = 0
            for u in self.adj[v]:
                a, b = self.pos[v], self.pos[u]
                D[ix[v], ix[u]] = math.sqrt(sum((p - q) ** 2 for p, q in zip(a, b)))
                nxt[ix[v], ix[u]] = ix[u]
        for k in range(n):
            for i in range(n):
                via = D[i, k] + D[k]
                better = via < D[i]
                D[i][better] = via[better]
                nxt[i][better] = nxt[i, k]
        self.dist = {v: {u: float(D[ix[v], ix[u]]) for u in self.vps} for v in self.vps}
        self._nxt, self._ix = nxt, ix

    def shortest_path(self, a, b):
        path, i, j = [a], self._ix[a], self._ix[b]
        while i != j:
            i = int(self._nxt[i, j])
            path.append(self.vps[i])
        return path

    # ---- per-viewpoint "sensor" data: a pure function of the key, memoised like the reference's in-memory feature
    #      store (utils/data.py ImageFeaturesDB keeps every viewpoint it has read)
    def _memo(self, key, make, limit=160, store="_memo_store"):
        c = self.__dict__.setdefault(store, {})
        if key not in c:
            if len(c) >= limit:
                c.pop(next(iter(c)))
            c[key] = make()
        return c[key]

    def view_features(self, vp, dim=768):
        return self._memo(("view", vp, dim), lambda: _rs("view", vp).standard_normal((36, dim)).astype(np.float32))

    def depth(self, vp, geom=NATIVE):
        def make():
            rs = _rs("depth", vp)
            d = rs.randint(0, 20000, size=(geom.n_views, geom.patches ** 2)).astype(np.uint16)
            d[rs.rand(*d.shape) < 0.1] = 0
            return d
        return self._memo(("depth", vp, geom.n_views, geom.patches), make)

    def patch_tokens(self, vp, geom=NATIVE):
        return self._memo(("clip", vp, geom.pts_per_obs, geom.feat_dim), lambda: (
            _rs("clip", vp).standard_normal((geom.pts_per_obs, geom.feat_dim)) * 0.35).astype(np.float16))


class _SimState:
    def __init__(self):
        self.scan = self.vp = None
        self.heading = self.elevation = 0.0

    @property
    def view_index(self):
        return 12 * (int(round(self.elevation / math.radians(30))) + 1) + int(round(self.heading / math.radians(30))) % 12


class SyntheticNavEnv:
    """Stand-in for R2RNavBatch (env.py:403-650) over SyntheticScan buildings.

    grid_memory: an object with reset() / step(depth, feats, poses, headings) / as_reference_obs()
    (gridmm_amd.grid_memory.GridMemoryBatch on the GPU; tests inject a CPU oracle adapter).
    """

    def __init__(self, batch_size, grid_memory, n_scans=2, n_episodes=8, seed=0, geom=NATIVE,
                 image_feat_size=768, angle_feat_size=4, vocab=2000, name="synthetic"):
        self.batch_size, self.geom, self.name = batch_size, geom, name
        self.image_feat_size, self.angle_feat_size = image_feat_size, angle_feat_size
        self.grid_memory = grid_memory
        self.scans = {("s%d" % k): SyntheticScan("s%d" % k, seed=seed + k) for k in range(n_scans)}
        self.shortest_distances = {s: sc.dist for s, sc in self.scans.items()}
        self.angle_feature = all_point_angle_features(angle_feat_size)
        rs = _rs("episodes", seed)
        self.data = []
        for e in range(n_episodes):
            scan = self.scans["s%d" % (e % n_scans)]
            while True:
                a, b = (str(v) for v in rs.choice(scan.vps, 2, replace=False))   # plain str: the per-viewpoint
                # generators key on repr(name), and numpy's str_ has another repr than the same name as a str
                path = scan.shortest_path(a, b)
                if 3 <= len(path) <= 7:
                    break
            n_tok = int(rs.randint(12, 40))
            self.data.append({
                "instr_id": "%d_0" % e, "path_id": e, "scan": scan.name, "path": path,
                "heading": float(rs.randint(0, 12)) * math.radians(30),
                "instruction": "synthetic", "instr_encoding": [101] + rs.randint(1000, vocab, size=n_tok).tolist() + [102],
            })
        self.gt_trajs = {x["instr_id"]: (x["scan"], x["path"]) for x in self.data}
        self.ix = 0
        self.batch = None
        self.sims = [_SimState() for _ in range(batch_size)]
        self.device_store = None          # feature_store.DeviceStore: observations resident in HBM (build_device_store)

    def build_device_store(self, device):
        """All viewpoints of all buildings -> one DeviceStore (patch tokens + sampled depth + pose); from then on an
        environment step gathers its observations on the device and the observation dicts carry no grid fields (the
        agent hands the grid memory to the model as a handle, agent.py:150-152)."""
        from .feature_store import DeviceStore
        keys, tok, dep, poses = [], [], [], []
        for sname, sc in self.scans.items():
            for vp in sc.vps:
                keys.append(vp)
                tok.append(sc.patch_tokens(vp, self.geom))
                dep.append(sc.depth(vp, self.geom).reshape(-1))
                poses.append(sc.pos[vp])
        self.device_store = DeviceStore(keys, np.stack(tok), np.stack(dep), poses, device)
        self.grid_memory.track_cmax = True        # the varlen navigation path then needs no mid-step read-back
        return self.device_store

    # ---- simulator surface used by the agent (agent.py:255: sims[i].newEpisode)
    def teleport(self, i, scan, vp, heading, elevation):
        s = self.sims[i]
        s.scan, s.vp, s.heading, s.elevation = scan, vp, float(heading), float(elevation)

    def size(self):
        return len(self.data)

    def reset_epoch(self):
        self.ix = 0

    def _next_minibatch(self):
        batch = self.data[self.ix:self.ix + self.batch_size]
        if len(batch) < self.batch_size:
            self.ix = self.batch_size - len(batch)
            batch = batch + self.data[:self.ix]
        else:
            self.ix += self.batch_size
        self.batch = batch

    def reset(self):
        """env.py:625-634."""
        self._next_minibatch()
        for i, item in enumerate(self.batch):
            self.teleport(i, item["scan"], item["path"][0], item["heading"], 0.0)
        self.grid_memory.reset()
        return self._get_obs()

    def make_candidate(self, feature, scan, vp, view_id):
        """env.py:507-575 (buffered branch): neighbours seen from the view that faces them best."""
        base_heading = (view_id % 12) * math.radians(30)
        base_elevation = (view_id // 12 - 1) * math.radians(30)
        sc = self.scans[scan]
        px, py, pz = sc.pos[vp]
        cands = []
        for j, nb in enumerate(sc.adj[vp]):
            x, y, z = sc.pos[nb]
            dx, dy, dz = x - px, y - py, z - pz
            nh = math.atan2(dx, dy) % (2 * math.pi)           # heading is measured from +y (graph_utils.py:24-27)
            ne = math.atan2(dz, math.hypot(dx, dy))
            ix = 12 + int(round(nh / math.radians(30))) % 12   # horizon view facing the neighbour
            h, e = np.float32(np.float32(nh) - base_heading), np.float32(np.float32(ne) - base_elevation)
            cands.append({
                "heading": h, "elevation": e, "scanId": scan, "viewpointId": nb, "pointId": np.int32(ix),
                "idx": np.int32(j + 1),
                "feature": np.concatenate((feature[ix].astype(np.float32),
                                           angle_feature(h, e, self.angle_feat_size)), -1).astype(np.float32),
                "position": (np.float32(x), np.float32(y), np.float32(z)),
            })
        return cands

    def _static_obs(self, sc, s, vi):
        feature = sc.view_features(s.vp, self.image_feat_size)
        return (feature, self.make_candidate(feature, s.scan, s.vp, vi),
                np.concatenate((feature, self.angle_feature[vi]), -1).astype(np.float32))

    def _get_obs(self):
        """env.py:583-623 + EnvBatch.getStates (env.py:377-400)."""
        B = self.batch_size
        states = self.sims
        if self.device_store is not None:
            depth, poses = self.device_store.append(self.grid_memory, [s.vp for s in states])
            self.grid_memory.step(depth, None, poses, [s.heading for s in states])
            grid_fts = grid_map = pos_fts = [None] * B
        else:
            depth = np.stack([self.scans[s.scan].depth(s.vp, self.geom).reshape(-1) for s in states])
            feats = np.stack([self.scans[s.scan].patch_tokens(s.vp, self.geom) for s in states])
            poses = [self.scans[s.scan].pos[s.vp][:2] for s in states]
            self.grid_memory.step(depth, feats, poses, [s.heading for s in states])
            grid_fts, grid_map, pos_fts = self.grid_memory.as_reference_obs()
        obs = []
        for i, s in enumerate(states):
            item = self.batch[i]
            sc = self.scans[s.scan]
            vi = s.view_index
            # The fields that are pure functions of (viewpoint, view index) -- candidates, the angle-extended view features,
            # the position -- are kept like the reference keeps its buffered_state_dict (env.py:529-575), the per-episode
            # constants (instruction, ground truth) are converted once per item; consumers treat them as read-only.
            static = sc._memo(("obs", s.vp, vi, self.image_feat_size), lambda: self._static_fields(sc, s, vi),
                              limit=4096, store="_obs_store")
            ep = item.get("_obs_fields")
            if ep is None:
                ep = item["_obs_fields"] = {
                    "instr_id": item["instr_id"], "instruction": item["instruction"],
                    "instr_encoding": [np.int32(t) for t in item["instr_encoding"]], "gt_path": item["path"],
                    "path_id": item["path_id"]}
            ob = dict(static)
            ob.update(ep)
            ob["heading"], ob["elevation"] = np.float32(s.heading), np.float32(s.elevation)
            ob["grid_fts"], ob["grid_map"], ob["gridmap_pos_fts"] = grid_fts[i], grid_map[i], pos_fts[i]
            ob["distance"] = np.float32(self.shortest_distances[s.scan][s.vp][item["path"][-1]])
            obs.append(ob)
        return obs

    def _static_fields(self, sc, s, vi):
        feature, cand, full_feature = self._static_obs(sc, s, vi)
        x, y, z = sc.pos[s.vp]
        return {"scan": s.scan, "viewpoint": s.vp, "viewIndex": vi, "position": (np.float32(x), np.float32(y), np.float32(z)),
                "feature": full_feature, "candidate": cand}
