"""Host-side topological map of one episode (restates map_nav_src/models/graph_utils.py:43-151).

Pure bookkeeping (python dicts + a few floats per node); node embeddings are device tensors owned by the
caller.  Feeds the `gmap_*` inputs of forward('navigation') (map_nav_src/r2r/agent.py:96-169).
"""
from collections import defaultdict

import numpy as np

MAX_DIST = 30   # graph_utils.py:4
MAX_STEP = 10   # graph_utils.py:5
UNREACHABLE = 95959595  # graph_utils.py:45


def calc_position_distance(a, b):
    dx, dy, dz = b[0] - a[0], b[1] - a[1], b[2] - a[2]
    return np.sqrt(dx ** 2 + dy ** 2 + dz ** 2)


def calculate_vp_rel_pos_fts(a, b, base_heading=0, base_elevation=0):
    """graph_utils.py:15-33 (the simulator's x-y axes are transposed: heading is measured from +y)."""
    dx, dy, dz = b[0] - a[0], b[1] - a[1], b[2] - a[2]
    xy_dist = max(np.sqrt(dx ** 2 + dy ** 2), 1e-8)
    xyz_dist = max(np.sqrt(dx ** 2 + dy ** 2 + dz ** 2), 1e-8)
    heading = np.arcsin(dx / xy_dist)
    if b[1] < a[1]:
        heading = np.pi - heading
    heading -= base_heading
    elevation = np.arcsin(dz / xyz_dist)
    elevation -= base_elevation
    return heading, elevation, xyz_dist


def get_angle_fts(headings, elevations, angle_feat_size=4):
    ang = np.vstack([np.sin(headings), np.cos(headings), np.sin(elevations), np.cos(elevations)])
    ang = ang.transpose().astype(np.float32)
    rep = angle_feat_size // 4
    return np.concatenate([ang] * rep, 1) if rep > 1 else ang


class FloydGraph:
    """Incremental all-pairs shortest paths over the visited sub-graph (graph_utils.py:43-92)."""

    def __init__(self):
        self._dis = defaultdict(lambda: defaultdict(lambda: UNREACHABLE))
        self._point = defaultdict(lambda: defaultdict(lambda: ""))
        self._visited = set()

    def distance(self, x, y):
        return 0 if x == y else self._dis[x][y]

    def add_edge(self, x, y, dis):
        if dis < self._dis[x][y]:
            self._dis[x][y] = dis
            self._dis[y][x] = dis
            self._point[x][y] = ""
            self._point[y][x] = ""

    def update(self, k):
        for x in self._dis:
            for y in self._dis:
                if x != y and self._dis[x][k] + self._dis[k][y] < self._dis[x][y]:
                    self._dis[x][y] = self._dis[x][k] + self._dis[k][y]
                    self._dis[y][x] = self._dis[x][y]
                    self._point[x][y] = k
                    self._point[y][x] = k
        self._visited.add(k)

    def visited(self, k):
        return k in self._visited

    def path(self, x, y):
        """[v1, ..., y] from x to y (x excluded)."""
        if x == y:
            return []
        if self._point[x][y] == "":
            return [y]
        k = self._point[x][y]
        return self.path(x, k) + self.path(k, y)


class GraphMap:
    def __init__(self, start_vp):
        self.start_vp = start_vp
        self.node_positions = {}
        self.graph = FloydGraph()
        self.node_embeds = {}        # vp -> [sum of embeddings (device tensor), count]
        self.node_stop_scores = {}
        self.node_nav_scores = {}
        self.node_step_ids = {}

    def update_graph(self, ob):
        self.node_positions[ob["viewpoint"]] = ob["position"]
        for cc in ob["candidate"]:
            self.node_positions[cc["viewpointId"]] = cc["position"]
            self.graph.add_edge(ob["viewpoint"], cc["viewpointId"], calc_position_distance(ob["position"], cc["position"]))
        self.graph.update(ob["viewpoint"])

    def update_node_embed(self, vp, embed, rewrite=False):
        if rewrite or vp not in self.node_embeds:
            self.node_embeds[vp] = [embed, 1]
        else:
            self.node_embeds[vp][0] = embed + self.node_embeds[vp][0]
            self.node_embeds[vp][1] += 1

    def get_node_embed(self, vp):
        e, n = self.node_embeds[vp]
        return e if n == 1 else e / n          # x / 1 == x exactly: skip the kernel + autograd node (most nodes: one visit)

    def get_pos_fts(self, cur_vp, gmap_vpids, cur_heading, cur_elevation, angle_feat_size=4):
        """(len, 7): sin/cos heading, sin/cos elevation, line dist/30, graph dist/30, hops/10 (graph_utils.py:127-151)."""
        rel_angles, rel_dists = [], []
        for vp in gmap_vpids:
            if vp is None:
                rel_angles.append([0, 0])
                rel_dists.append([0, 0, 0])
            else:
                h, e, d = calculate_vp_rel_pos_fts(self.node_positions[cur_vp], self.node_positions[vp],
                                                   base_heading=cur_heading, base_elevation=cur_elevation)
                rel_angles.append([h, e])
                rel_dists.append([d / MAX_DIST, self.graph.distance(cur_vp, vp) / MAX_DIST,
                                  len(self.graph.path(cur_vp, vp)) / MAX_STEP])
        rel_angles = np.array(rel_angles).astype(np.float32)
        rel_dists = np.array(rel_dists).astype(np.float32)
        return np.concatenate([get_angle_fts(rel_angles[:, 0], rel_angles[:, 1], angle_feat_size), rel_dists], 1)
