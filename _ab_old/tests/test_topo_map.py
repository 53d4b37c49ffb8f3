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
