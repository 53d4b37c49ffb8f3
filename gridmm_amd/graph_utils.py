"""Topological map of one episode on dense arrays (host side).

Contract (what the reference's map_nav_src/models/graph_utils.py:43-151 gives its caller, r2r/agent.py:96-320):
the viewpoints seen so far with their positions, all-pairs shortest distances over the edges observed so far
(refreshed only through the node the agent stands on, i.e. a node-at-a-time Floyd-Warshall relaxation), the hop
sequence between two nodes, running means of node embeddings, and the 7 position features per node
(sin/cos heading, sin/cos elevation, line distance / 30, graph distance / 30, hops / 10).

Own design: viewpoint ids are interned to integers; distances and the relaxation pivot ("via") live in dense
numpy matrices that grow geometrically; one relaxation step is a vectorised outer sum; routes are unrolled with an
explicit stack; `pair_distances` / `pos_features` are computed for whole id vectors at once.  Unreachable pairs
read back as UNREACHABLE (the reference's 95959595 sentinel ends up in `gmap_pair_dists`, so it is part of the
data contract); internally they are +inf.  Pinned by tests/golden/topo_map.npz (reference driven over a scripted walk).
"""
import numpy as np

MAX_DIST = 30.0   # metres: scale of the two distance features
MAX_STEP = 10.0   # hops:   scale of the hop feature
UNREACHABLE = 95959595


def heading_elevation_distance(origin, targets, base_heading=0.0, base_elevation=0.0):
    """origin (3,), targets (n, 3) -> heading (n,), elevation (n,), distance (n,) in float64.  The simulator measures
    heading from the +y axis, hence asin(dx / r) mirrored for targets behind the origin."""
    d = np.asarray(targets, dtype=np.float64).reshape(-1, 3) - np.asarray(origin, dtype=np.float64)
    flat = np.maximum(np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2), 1e-8)
    full = np.maximum(np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2 + d[:, 2] ** 2), 1e-8)
    heading = np.arcsin(d[:, 0] / flat)
    heading = np.where(d[:, 1] < 0, np.pi - heading, heading) - base_heading
    elevation = np.arcsin(d[:, 2] / full) - base_elevation
    return heading, elevation, full


def angle_features(headings, elevations, angle_feat_size=4):
    """(n,) float32 angles -> (n, angle_feat_size): [sin h, cos h, sin e, cos e] tiled."""
    h = np.asarray(headings, dtype=np.float32)
    e = np.asarray(elevations, dtype=np.float32)
    quad = np.stack([np.sin(h), np.cos(h), np.sin(e), np.cos(e)], 1).astype(np.float32)
    return np.tile(quad, (1, max(1, angle_feat_size // 4)))


def batched_pos_features(delta, base_heading, base_elevation, graph, hops, angle_feat_size=4):
    """pos_features of many (origin, target) pairs at once: delta (N, 3) float64 target - origin, base heading /
    elevation (N,), graph distance and hops (N,) -> (N, angle_feat_size + 3) float32.  Same arithmetic, in the same
    precisions, as heading_elevation_distance + angle_features + TopoMap.pos_features."""
    flat = np.maximum(np.sqrt(delta[:, 0] ** 2 + delta[:, 1] ** 2), 1e-8)
    full = np.maximum(np.sqrt(delta[:, 0] ** 2 + delta[:, 1] ** 2 + delta[:, 2] ** 2), 1e-8)
    heading = np.arcsin(delta[:, 0] / flat)
    heading = np.where(delta[:, 1] < 0, np.pi - heading, heading) - base_heading
    elevation = np.arcsin(delta[:, 2] / full) - base_elevation
    h, e = heading.astype(np.float32), elevation.astype(np.float32)
    out = np.empty((len(h), angle_feat_size + 3), dtype=np.float32)
    quad = np.stack([np.sin(h), np.cos(h), np.sin(e), np.cos(e)], 1)
    out[:, :angle_feat_size] = np.tile(quad, (1, max(1, angle_feat_size // 4)))
    out[:, angle_feat_size] = (full / MAX_DIST).astype(np.float32)
    out[:, angle_feat_size + 1] = (graph / MAX_DIST).astype(np.float32)
    out[:, angle_feat_size + 2] = (hops / MAX_STEP).astype(np.float32)
    return out


class TopoMap:
    def __init__(self, start_vp, capacity=32, batch=None, row=0):
        """batch / row: the arrays are row `row` of a TopoMapBatch's (B, cap, ...) storage (views), so that a caller can
        run one vectorised pass over all episodes of a lock-step batch; alone, the map owns its arrays."""
        self.start_vp = start_vp
        self._id = {}                  # viewpoint name -> dense integer id (insertion order)
        self.names = []                # id -> name
        self._batch, self._row = batch, row
        if batch is None:
            self._alloc(capacity)
        else:
            batch._attach(self)
        self._n = 0
        self._emb = []                 # id -> [running sum (tensor), count] or None
        self.step_id = {}              # name -> navigation step at which it was last visited
        self.stop_score = {}           # name -> {'stop': p}

    # ---- storage -------------------------------------------------------------------------------
    def _alloc(self, cap):
        self.pos = np.zeros((cap, 3), dtype=np.float64)
        self.dist = np.full((cap, cap), np.inf, dtype=np.float64)   # diagonal stays inf: a node is no pivot of itself
        self.via = np.full((cap, cap), -1, dtype=np.int32)          # -1: direct edge (or nothing known)
        self.seen = np.zeros(cap, dtype=bool)                       # visited = has been a relaxation pivot

    @property
    def n(self):
        return self._n

    @n.setter
    def n(self, v):
        self._n = v
        if self._batch is not None:
            self._batch.n[self._row] = v

    def _grow(self):
        if self._batch is not None:
            self._batch.grow()         # re-points the arrays of every map of the batch
            return
        old = (self.pos, self.dist, self.via, self.seen)
        c = old[0].shape[0]
        self._alloc(2 * c)
        self.pos[:c], self.dist[:c, :c], self.via[:c, :c], self.seen[:c] = old

    def intern(self, vp):
        i = self._id.get(vp)
        if i is None:
            if self.n == self.pos.shape[0]:
                self._grow()
            i = self._id[vp] = self.n
            self.names.append(vp)
            self._emb.append(None)
            self.n += 1
        return i

    def __contains__(self, vp):
        return vp in self._id

    def nodes(self):
        """Viewpoint names in order of first appearance."""
        return list(self.names)

    def position(self, vp):
        return self.pos[self._id[vp]]

    # ---- graph ---------------------------------------------------------------------------------
    def observe(self, ob):
        """Add the observation's viewpoint, its candidates and the edges between them, then relax all pairs through
        the viewpoint (the agent stands on it: it becomes 'visited')."""
        k = self.intern(ob["viewpoint"])
        self.pos[k] = ob["position"]
        for cand in ob["candidate"]:
            c = self.intern(cand["viewpointId"])
            self.pos[c] = cand["position"]
            delta = self.pos[c] - self.pos[k]
            w = np.sqrt(delta[0] ** 2 + delta[1] ** 2 + delta[2] ** 2)
            if w < self.dist[k, c]:
                self.dist[k, c] = self.dist[c, k] = w
                self.via[k, c] = self.via[c, k] = -1
        n = self.n
        d = self.dist[:n, :n]
        through = d[:, k, None] + d[None, k, :]        # row/column k are untouched by the update (d[k, k] = inf)
        better = through < d
        np.fill_diagonal(better, False)
        d[better] = through[better]
        self.via[:n, :n][better] = k
        self.seen[k] = True

    def visited(self, vp):
        i = self._id.get(vp)
        return bool(i is not None and self.seen[i])

    def distance(self, a, b):
        if a == b:
            return 0
        ia, ib = self._id.get(a), self._id.get(b)
        if ia is None or ib is None or not np.isfinite(self.dist[ia, ib]):
            return UNREACHABLE
        return self.dist[ia, ib]

    def _route_ids(self, a, b):
        out, stack = [], [(a, b)]
        while stack:
            x, y = stack.pop()
            if x == y:
                continue
            k = self.via[x, y]
            if k < 0:
                out.append(y)
            else:
                stack.append((k, y))      # second leg is emitted after the first
                stack.append((x, k))
        return out

    def route(self, a, b):
        """Viewpoints from a (excluded) to b (included) along the relaxation pivots."""
        if a == b:
            return []
        return [self.names[i] for i in self._route_ids(self._id[a], self._id[b])]

    def hops(self, a, b):
        return 0 if a == b else len(self._route_ids(self._id[a], self._id[b]))

    def index(self, vp):
        """Dense id of a known viewpoint (insertion order)."""
        return self._id[vp]

    def hops_from(self, cur, ids):
        """Route lengths from node id `cur` to every id in `ids` (0 for cur itself): direct edges are read off the
        pivot matrix, only multi-leg routes are unrolled."""
        v = self.via[cur, ids]
        out = np.where(ids == cur, 0.0, 1.0)
        for j in np.flatnonzero(v >= 0):
            if ids[j] != cur:
                out[j] = len(self._route_ids(cur, int(ids[j])))
        return out

    def relative_geometry(self, cur, ids):
        """Raw inputs of pos_features for node ids `ids` seen from node `cur`: (position deltas (n, 3), graph distance
        (n,) with UNREACHABLE for unknown routes and 0 for cur itself, hops (n,)) -- so that a caller can run the
        trigonometry of many episodes in one vectorised pass (batched_pos_features)."""
        graph = np.where(ids == cur, 0.0, self.dist[cur, ids])
        graph = np.where(np.isfinite(graph), graph, float(UNREACHABLE))
        return self.pos[ids] - self.pos[cur], graph, self.hops_from(cur, ids)

    def pair_distances(self, vpids):
        """(n, n) float32 matrix of graph distances between the named nodes; rows/columns of `None` entries (the
        stop token) and the diagonal are 0."""
        n = len(vpids)
        out = np.zeros((n, n), dtype=np.float32)
        idx = np.array([self._id[v] for v in vpids if v is not None], dtype=np.int64)
        slots = np.array([j for j, v in enumerate(vpids) if v is not None], dtype=np.int64)
        if len(idx):
            sub = self.dist[np.ix_(idx, idx)]
            sub = np.where(np.isfinite(sub), sub, float(UNREACHABLE))
            np.fill_diagonal(sub, 0.0)
            out[np.ix_(slots, slots)] = sub
        return out

    # ---- node embeddings -----------------------------------------------------------------------
    def add_embedding(self, vp, embed, overwrite=False):
        i = self.intern(vp)
        cur = self._emb[i]
        if overwrite or cur is None:
            self._emb[i] = [embed, 1]
        else:
            cur[0] = embed + cur[0]
            cur[1] += 1

    def embedding(self, vp):
        s, c = self._emb[self._id[vp]]
        return s if c == 1 else s / c      # mean over the views that saw the node; x / 1 == x exactly

    # ---- features ------------------------------------------------------------------------------
    def pos_features(self, cur_vp, vpids, cur_heading, cur_elevation, angle_feat_size=4):
        """(len(vpids), angle_feat_size + 3) float32; all-zero angles / distances for `None` (the stop token)."""
        n = len(vpids)
        ang = np.zeros((n, 2), dtype=np.float32)
        rel = np.zeros((n, 3), dtype=np.float32)
        slots = [j for j, v in enumerate(vpids) if v is not None]
        if slots:
            cur = self._id[cur_vp]
            ids = np.array([self._id[vpids[j]] for j in slots], dtype=np.int64)
            h, e, line = heading_elevation_distance(self.pos[cur], self.pos[ids], cur_heading, cur_elevation)
            graph = np.where(ids == cur, 0.0, self.dist[cur, ids])
            graph = np.where(np.isfinite(graph), graph, float(UNREACHABLE))
            hops = np.array([len(self._route_ids(cur, i)) for i in ids], dtype=np.float64)
            ang[slots, 0], ang[slots, 1] = h, e
            rel[slots, 0], rel[slots, 1], rel[slots, 2] = line / MAX_DIST, graph / MAX_DIST, hops / MAX_STEP
        return np.concatenate([angle_features(ang[:, 0], ang[:, 1], angle_feat_size), rel], 1)


class TopoMapBatch:
    """Storage of the B topological maps of a lock-step batch as (B, cap, ...) arrays; `maps[b]` is a TopoMap whose
    arrays are views of row b.  Per-episode calls (observe, route, pos_features, ...) work as before; a collator reads
    the batch arrays directly (collate.NavCollator.navigation)."""

    def __init__(self, start_vps, capacity=64):
        self.B, self.cap = len(start_vps), capacity
        self._alloc(capacity)
        self.n = np.zeros(self.B, dtype=np.int64)
        self.step = np.zeros((self.B, capacity), dtype=np.int64)     # last navigation step at which a node was visited
        self.maps = []
        self.maps = [TopoMap(vp, batch=self, row=b) for b, vp in enumerate(start_vps)]

    def _alloc(self, cap):
        B = self.B
        self.pos = np.zeros((B, cap, 3), dtype=np.float64)
        self.dist = np.full((B, cap, cap), np.inf, dtype=np.float64)
        self.via = np.full((B, cap, cap), -1, dtype=np.int32)
        self.seen = np.zeros((B, cap), dtype=bool)

    def _attach(self, m):
        b = m._row
        m.pos, m.dist, m.via, m.seen = self.pos[b], self.dist[b], self.via[b], self.seen[b]

    def grow(self):
        old, c = (self.pos, self.dist, self.via, self.seen, self.step), self.cap
        self.cap = 2 * c
        self._alloc(self.cap)
        self.pos[:, :c], self.dist[:, :c, :c], self.via[:, :c, :c], self.seen[:, :c] = old[:4]
        self.step = np.zeros((self.B, self.cap), dtype=np.int64)
        self.step[:, :c] = old[4]
        for m in self.maps:
            self._attach(m)

    def observe_all(self, obs, active=None):
        """TopoMap.observe(obs[b]) for every (active) episode, with the all-pairs relaxation of the B maps done as ONE pass
        over the (B, n, n) arrays: same arithmetic (float64 edge lengths, strict `<` improvements, pivot ids), same result
        as the per-map calls (tests/test_topo_map.py)."""
        bs, ks, kpos = [], [], []
        eb, ec, epos = [], [], []                    # candidate edges: row of bs, candidate node id, candidate position
        for b, ob in enumerate(obs):
            if active is not None and not active[b]:
                continue
            m = self.maps[b]
            k = m.intern(ob["viewpoint"])
            cands = ob["candidate"]
            cs = [m.intern(c["viewpointId"]) for c in cands]                # (may re-point the arrays: intern first)
            if cs:
                eb.extend([len(bs)] * len(cs))
                ec.extend(cs)
                epos.extend(c["position"] for c in cands)
            bs.append(b)
            ks.append(k)
            kpos.append(ob["position"])
        if bs:
            # positions and observed edges of all episodes in whole-array passes (same float64 arithmetic as TopoMap.observe:
            # edge length from the stored positions, strict `<` improvements; a candidate listed twice keeps the shorter edge)
            bsa, ksa = np.asarray(bs, dtype=np.int64), np.asarray(ks, dtype=np.int64)
            self.pos[bsa, ksa] = np.asarray(kpos, dtype=np.float64)
            if ec:
                e = np.asarray(eb, dtype=np.int64)
                bi, ki, ci = bsa[e], ksa[e], np.asarray(ec, dtype=np.int64)
                self.pos[bi, ci] = np.asarray(epos, dtype=np.float64)
                delta = self.pos[bi, ci] - self.pos[bi, ki]
                w = np.sqrt(delta[:, 0] ** 2 + delta[:, 1] ** 2 + delta[:, 2] ** 2)
                imp = w < self.dist[bi, ki, ci]
                if imp.any():
                    bi, ki, ci, w = bi[imp], ki[imp], ci[imp], w[imp]
                    np.minimum.at(self.dist, (bi, ki, ci), w)
                    np.minimum.at(self.dist, (bi, ci, ki), w)
                    self.via[bi, ki, ci] = -1
                    self.via[bi, ci, ki] = -1
        if not bs:
            return
        bs, ks = np.asarray(bs, dtype=np.int64), np.asarray(ks, dtype=np.int64)
        n = int(self.n[bs].max())
        ar = np.arange(len(bs))
        d = self.dist[bs, :n, :n]                            # (nb, n, n) copies; rows / columns past an episode's n are inf
        through = d[ar, :, ks][:, :, None] + d[ar, ks, :][:, None, :]
        better = through < d
        better[:, np.arange(n), np.arange(n)] = False
        d[better] = through[better]
        self.dist[bs, :n, :n] = d
        v = self.via[bs, :n, :n]
        v[better] = np.broadcast_to(ks[:, None, None].astype(np.int32), better.shape)[better]
        self.via[bs, :n, :n] = v
        self.seen[bs, ks] = True

    def mark_step(self, b, vp, t):
        """The agent's `gmap.step_id[vp] = t` (agent.py:277-279) mirrored into the batch array."""
        m = self.maps[b]
        m.step_id[vp] = t
        self.step[b, m.index(vp)] = t
