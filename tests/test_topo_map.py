"""gridmm_amd.graph_utils.TopoMap against tests/golden/topo_map.npz -- the reference's FloydGraph / GraphMap
(map_nav_src/models/graph_utils.py:43-151) driven over a scripted 20-step walk by oracle/gen_golden.py."""
import os

import numpy as np
import torch

from gridmm_amd.graph_utils import TopoMap, UNREACHABLE

GOLD = os.path.join(os.path.dirname(__file__), "golden", "topo_map.npz")


def _obs(g, t):
    i = int(g["in_walk"][t])
    return {"viewpoint": "vp%02d" % i, "position": tuple(g["in_pos"][i]),
            "candidate": [{"viewpointId": "vp%02d" % j, "position": tuple(g["in_pos"][j])}
                          for j in np.nonzero(g["in_adj"][i])[0]]}


def test_topo_map_matches_reference_walk():
    g = np.load(GOLD)
    T = len(g["in_walk"])
    tm = TopoMap("vp%02d" % g["in_walk"][0], capacity=4)     # small capacity: exercises the growth path
    for t in range(T):
        ob = _obs(g, t)
        tm.observe(ob)
        cur = ob["viewpoint"]
        tm.add_embedding(cur, torch.from_numpy(g["in_embeds"][t, 0]), overwrite=True)
        for c, cc in enumerate(ob["candidate"]):
            if not tm.visited(cc["viewpointId"]):
                tm.add_embedding(cc["viewpointId"], torch.from_numpy(g["in_embeds"][t, 1 + c]))
        names = tm.nodes()
        n = len(names)
        assert [int(v[2:]) for v in names] == list(g["order"][t, :n]) and (g["order"][t, n:] == -1).all()
        for a, va in enumerate(names):
            assert tm.visited(va) == bool(g["visited"][t, a])
            np.testing.assert_allclose(tm.embedding(va).numpy(), g["emb"][t, a], rtol=0, atol=1e-6)
            for b, vb in enumerate(names):
                want = g["dist"][t, a, b]
                assert tm.distance(va, vb) == want, (t, va, vb)          # same sums in the same order: exact
                assert tm.hops(va, vb) == g["hops"][t, a, b]
            r = [int(v[2:]) for v in tm.route(cur, va)]
            assert r == [x for x in g["route"][t, a] if x >= 0]
        pair = tm.pair_distances([None] + names)
        want = g["dist"][t, :n, :n].astype(np.float32)
        assert (pair[0] == 0).all() and (pair[:, 0] == 0).all()
        np.testing.assert_array_equal(pair[1:, 1:], want)
        fts = tm.pos_features(cur, [None] + names, g["in_heading"][t], g["in_elevation"][t])
        assert fts.dtype == np.float32 and fts.shape == (n + 1, 7)
        np.testing.assert_allclose(fts, g["pos_fts"][t, :n + 1], rtol=0, atol=2e-6)


def test_topo_map_degenerate_queries():
    tm = TopoMap("a")
    tm.observe({"viewpoint": "a", "position": (0.0, 0.0, 0.0), "candidate": []})
    assert tm.distance("a", "a") == 0 and tm.route("a", "a") == [] and tm.hops("a", "a") == 0
    assert tm.distance("a", "zz") == UNREACHABLE
    f = tm.pos_features("a", [None, "a"], 0.3, 0.1)
    assert np.allclose(f[0], [0, 1, 0, 1, 0, 0, 0])                       # stop token: zero angles / distances
    assert np.allclose(f[1, 4:], 0) and np.isclose(f[1, 0], np.sin(-0.3), atol=1e-6)


def test_topo_map_batch_rows_equal_standalone_maps():
    """graph_utils.TopoMapBatch: B maps as rows of one set of arrays (what the batched collation reads).  Every row must
    behave exactly like a stand-alone TopoMap -- including growth of the shared storage past its initial capacity."""
    from gridmm_amd.graph_utils import TopoMapBatch
    g = np.load(GOLD)
    T = len(g["in_walk"])
    starts = ["vp%02d" % g["in_walk"][0]] * 3
    tb = TopoMapBatch(starts, capacity=4)                    # grows several times over the 20-step walk
    alone = [TopoMap(s, capacity=4) for s in starts]
    for t in range(T):
        for b in range(3):
            ob = _obs(g, (t + 3 * b) % T if b else t)        # the rows walk differently
            if ob["viewpoint"] not in alone[b] and t:        # (keep walks connected: only move to known nodes)
                ob = _obs(g, t)
            tb.maps[b].observe(ob)
            alone[b].observe(ob)
            tb.mark_step(b, ob["viewpoint"], t + 1)
            alone[b].step_id[ob["viewpoint"]] = t + 1
        for b in range(3):
            m, a = tb.maps[b], alone[b]
            n = a.n
            assert m.n == n == tb.n[b] and m.nodes() == a.nodes()
            assert np.array_equal(m.dist[:n, :n], a.dist[:n, :n]) and np.array_equal(m.via[:n, :n], a.via[:n, :n])
            assert np.array_equal(m.seen[:n], a.seen[:n]) and np.array_equal(m.pos[:n], a.pos[:n])
            assert np.array_equal(tb.dist[b, :n, :n], a.dist[:n, :n])           # the batch arrays ARE the rows' storage
            assert [int(tb.step[b, m.index(v)]) for v in a.nodes()] == [a.step_id.get(v, 0) for v in a.nodes()]
            cur = a.nodes()[-1]
            np.testing.assert_array_equal(m.pos_features(cur, a.nodes(), 0.3, 0.1), a.pos_features(cur, a.nodes(), 0.3, 0.1))
    assert tb.cap > 4


def test_observe_all_equals_per_map_observe():
    """TopoMapBatch.observe_all: the relaxation of all episodes in one pass over the (B, n, n) arrays == TopoMap.observe
    per episode (inactive rows untouched), including storage growth."""
    from gridmm_amd.graph_utils import TopoMapBatch
    g = np.load(GOLD)
    T = len(g["in_walk"])
    starts = ["vp%02d" % g["in_walk"][0]] * 4
    tb = TopoMapBatch(starts, capacity=4)
    alone = [TopoMap(s, capacity=4) for s in starts]
    for t in range(T):
        obs, active = [], []
        for b in range(4):
            ob = _obs(g, (t + 2 * b) % T if b else t)
            if ob["viewpoint"] not in alone[b] and t:
                ob = _obs(g, t)
            obs.append(ob)
            active.append(not (b == 3 and t % 3 == 1))      # row 3 sits out every third step
        tb.observe_all(obs, active)
        for b in range(4):
            if active[b]:
                alone[b].observe(obs[b])
        for b in range(4):
            m, a = tb.maps[b], alone[b]
            n = a.n
            assert m.n == n and m.nodes() == a.nodes()
            assert np.array_equal(m.dist[:n, :n], a.dist[:n, :n]) and np.array_equal(m.via[:n, :n], a.via[:n, :n])
            assert np.array_equal(m.seen[:n], a.seen[:n]) and np.array_equal(m.pos[:n], a.pos[:n])
            for x in range(n):
                for y in range(n):
                    assert m._route_ids(x, y) == a._route_ids(x, y)
    assert tb.cap > 4
