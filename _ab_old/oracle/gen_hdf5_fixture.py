"""Writes tests/golden/hdf5/features_small.hdf5 (+ .npz of the same arrays) with a real h5py / libhdf5.

Test infrastructure.  The datasets are created exactly the way the reference's feature writer creates them
(preprocess/get_map_feature.py:171-187: one dataset per "<scan>_<viewpoint>" key in the root group,
create_dataset(key, shape, dtype='float', compression='gzip'), five attributes per dataset), so that
gridmm_amd.hdf5_lite is pinned on the file structure the reference's files have: gzip chunks, attribute messages and
object-header continuations, and enough keys for a multi-node group B-tree.  Two extra datasets cover the other on-disk
types the readers cast to (uint16 depth: r2r/env.py:92-94; float16 semantic: :110-112) with shuffle + gzip.

h5py is not importable in the product interpreter of this image; the conda python3.9 that ships with it has it:
    /opt/conda/bin/python3.9 oracle/gen_hdf5_fixture.py
"""
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "hdf5")


def main():
    rng = np.random.RandomState(7)
    arrays = {}
    path = os.path.join(OUT, "features_small.hdf5")
    with h5py.File(path, "w") as outf:
        for s in range(3):
            for v in range(14):
                key = "scan%02d_vp%04x" % (s, v * 977)
                data = np.round(rng.randn(12, 20) * 4) / 8           # compressible, exactly representable
                outf.create_dataset(key, data.shape, dtype="float", compression="gzip")
                outf[key][...] = data
                outf[key].attrs["scanId"] = "scan%02d" % s
                outf[key].attrs["viewpointId"] = "vp%04x" % (v * 977)
                outf[key].attrs["image_w"] = 224
                outf[key].attrs["image_h"] = 224
                outf[key].attrs["vfov"] = 60
                arrays[key] = data
        d = rng.randint(0, 40000, size=(36, 50)).astype(np.uint16)
        outf.create_dataset("depth_u16", data=d, compression="gzip", shuffle=True, chunks=(8, 16))
        arrays["depth_u16"] = d
        h = (rng.randn(36, 50) * 2).astype(np.float16)
        outf.create_dataset("sem_f16", data=h, compression="gzip", shuffle=True, fletcher32=True)
        arrays["sem_f16"] = h
        outf.create_dataset("plain_f32", data=np.arange(24, dtype=np.float32).reshape(2, 3, 4))
        arrays["plain_f32"] = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    np.savez_compressed(os.path.join(OUT, "features_small.npz"), **arrays)
    print(path, os.path.getsize(path), "bytes,", len(arrays), "datasets; h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)


def reference_shaped():
    """clip_small.hdf5 / depth_small.hdf5 / viewpoint_info_small.json: three viewpoints in the reference's real shapes
    (clip rows (12, 52, 768) 'float', depth rows (36, 128*128+3) 'float'; r2r/env.py:80-113), with low-entropy content so
    that the gzip'd files stay small.  The expected packed records are recomputed in the test with the reference's own
    slicing expressions from the .npz copies of the same arrays."""
    import json
    rng = np.random.RandomState(11)
    info, clip_arrays, depth_arrays = {}, {}, {}
    with h5py.File(os.path.join(OUT, "clip_small.hdf5"), "w") as fc, h5py.File(os.path.join(OUT, "depth_small.hdf5"), "w") as fd:
        for i in range(3):
            key = "scanA_vp%02d" % i
            base = rng.randint(-24, 24, size=(12, 52, 1)) / 8.0
            ramp = (np.arange(768) % 16 - 8)[None, None] / 16.0
            clip = (base + ramp * (i + 1)).astype(np.float16).astype(np.float64)
            blocks = rng.randint(0, 2500, size=(36, 8, 8)) * 8
            blocks[rng.rand(36, 8, 8) < 0.15] = 0                         # invalid depth readings
            depth = np.repeat(np.repeat(blocks, 16, 1), 16, 2).reshape(36, -1)
            depth = np.concatenate([depth, np.full((36, 3), 7)], 1).astype(np.float64)
            for f, a in ((fc, clip), (fd, depth)):
                f.create_dataset(key, a.shape, dtype="float", compression="gzip")
                f[key][...] = a
                f[key].attrs["scanId"] = "scanA"
                f[key].attrs["viewpointId"] = "vp%02d" % i
            clip_arrays[key], depth_arrays[key] = clip.astype(np.float16), depth.astype(np.uint16)
            info[key] = {"x": float(rng.uniform(-20, 20)), "y": float(rng.uniform(-20, 20)), "z": float(rng.uniform(0, 3))}
    json.dump(info, open(os.path.join(OUT, "viewpoint_info_small.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(OUT, "clip_small.npz"), **clip_arrays)
    np.savez_compressed(os.path.join(OUT, "depth_small.npz"), **depth_arrays)
    for n in ("clip_small.hdf5", "depth_small.hdf5", "clip_small.npz", "depth_small.npz"):
        print(n, os.path.getsize(os.path.join(OUT, n)))


if __name__ == "__main__":
    main()
    reference_shaped()
