"""CPU: packed observation store: bit-exact round trip, converter slicing == the reference's slicing
(env.py:80-113, 279-303), and the stored depth drives the grid memory exactly like the full depth maps do."""
import json
import os

import numpy as np

from gridmm_amd import feature_store as FS
from oracle import gridmap_oracle as G


class _H5Like(dict):
    """key -> object with [...] like an h5py dataset."""

    class _DS:
        def __init__(self, a):
            self.a = a

        def __getitem__(self, idx):
            return self.a[idx]

    def __getitem__(self, k):
        return self._DS(dict.__getitem__(self, k))


def _fake_reference_files(tmp_path, n=5):
    rs = np.random.RandomState(3)
    clip, depth, info = _H5Like(), _H5Like(), {}
    for i in range(n):
        key = "scan%d_vp%02d" % (i % 2, i)
        # float64-typed, fp16-valued, 2 junk columns behind the 50 tokens (the reference slices [:, :50])
        clip[key] = rs.standard_normal((12, 52, 768)).astype(np.float16).astype(np.float64)
        d = rs.randint(0, 20000, size=(36, 128 * 128 + 3)).astype(np.float64)     # uint16-valued, 3 junk columns
        d[rs.rand(*d.shape) < 0.1] = 0
        depth[key] = d
        info[key] = {"x": float(rs.uniform(-20, 20)), "y": float(rs.uniform(-20, 20)), "z": float(rs.uniform(0, 3))}
    p = os.path.join(str(tmp_path), "viewpoint_info.json")
    json.dump(info, open(p, "w"))
    return clip, depth, info, p


def test_round_trip_and_converter_matches_reference_slicing(tmp_path):
    clip, depth, info, info_path = _fake_reference_files(tmp_path)
    files = {"clip": clip, "depth": depth}
    out = os.path.join(str(tmp_path), "obs.gmm")
    FS.convert_reference_files("clip", "depth", info_path, out, opener=lambda p: files[p])
    st = FS.PackedStore(out)
    assert len(st) == len(info) and sorted(st.keys) == sorted(info)
    for key in info:
        d, tok, (x, y, z) = st.get(key)
        # the reference's own expressions
        sem = clip[key][...][:, :50].astype(np.float16)[:, 1:].reshape(-1, 768)                       # env.py:109,299
        full = depth[key][...][:, :128 * 128].astype(np.uint16).reshape(36, 128, 128)                   # env.py:91
        want = G.sample_depth(full, G.NATIVE, slice(12, 24)).reshape(-1)                                # env.py:279-285
        assert tok.dtype == np.float16 and np.array_equal(tok, sem)
        assert d.dtype == np.uint16 and np.array_equal(d, want)
        assert (x, y, z) == (info[key]["x"], info[key]["y"], info[key]["z"])                            # doubles, exact
    keys = list(info)[:3]
    dd, ff, pp = st.gather(keys)
    assert dd.shape == (3, 588) and ff.shape == (3, 588, 768) and len(pp) == 3


def test_store_drives_the_grid_memory_like_the_full_maps(tmp_path):
    clip, depth, info, info_path = _fake_reference_files(tmp_path, n=4)
    files = {"clip": clip, "depth": depth}
    out = os.path.join(str(tmp_path), "obs.gmm")
    FS.convert_reference_files("clip", "depth", info_path, out, opener=lambda p: files[p])
    st = FS.PackedStore(out)
    a, b = G.GridMemory(G.NATIVE), G.GridMemory(G.NATIVE)
    for t, key in enumerate(info):
        d, tok, (x, y, _) = st.get(key)
        heading = 0.5 * t
        ra = a.step(np.asarray(d).reshape(12, 49), np.asarray(tok), x, y, heading)
        full = depth[key][...][:, :128 * 128].astype(np.uint16).reshape(36, 128, 128)
        sem = clip[key][...][:, :50].astype(np.float16)[:, 1:].reshape(-1, 768)
        rb = b.step(G.sample_depth(full, G.NATIVE, slice(12, 24)), sem, info[key]["x"], info[key]["y"], heading)
        assert np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]) and np.array_equal(ra[0], rb[0])


def test_rejects_foreign_files(tmp_path):
    p = os.path.join(str(tmp_path), "x.bin")
    open(p, "wb").write(b"not a store" * 10)
    try:
        FS.PackedStore(p)
    except ValueError:
        return
    raise AssertionError("foreign file accepted")
