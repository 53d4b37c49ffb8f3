"""Packed, mmap-able observation store for the grid memory + converter from the reference's files.

What the reference reads on every step (map_nav_src/r2r/env.py):
  clip_p32.hdf5        key "<scan>_<vp>" -> (12, 50+, 768), float64-TYPED gzip h5 holding fp16 VALUES; used as
                       f[key][...][:, :50].astype(np.float16) (env.py:97-113) and then semantic[:, 1:] (CLS dropped,
                       env.py:299-303): 12 horizon views x 49 patch tokens
  depth.hdf5           key "<scan>_<vp>" -> (36, 128*128[+]) uint16-valued, 0.25 mm units; used as
                       f[key][...][:, :16384].astype(np.uint16) (env.py:80-94), reshaped (36,128,128), sampled at
                       idx = [9,27,...,117] on both axes (env.py:279-281), horizon views [12:24] only (env.py:283-285)
  viewpoint_info.json  key -> {"x","y","z"} python floats (env.py:168,286)
All of it is decompressed / type-converted per viewpoint on first touch and cached in python dicts.

Here: ONE file, fixed-size records, no decompression and no per-step conversion -- exactly the bytes the grid memory consumes:
  tokens (n, 588, 768) fp16   depth (n, 588) uint16   pose (n, 3) float64 (x, y, z as the JSON doubles)
memory-mapped; `gather(keys)` returns zero-copy views in the layout GridMemoryBatch.step() takes.

File layout (little endian):  b"GMMSTORE" | u32 version | u32 header_len | header JSON (utf-8, padded to 64 B) |
                              pose block | depth block | token block      (each block 64-B aligned)
"""
import json
import os
import struct

import numpy as np

MAGIC = b"GMMSTORE"
VERSION = 1
_ALIGN = 64


def _pad(n):
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


class PackedStoreWriter:
    """Two-pass-free writer: records are appended to three growing temp arrays and laid out on close()."""

    def __init__(self, path, n_views=12, patches=49, feat_dim=768):
        self.path, self.n_views, self.patches, self.feat_dim = path, n_views, patches, feat_dim
        self.keys, self._tok, self._dep, self._pose = [], [], [], []

    def add(self, key, tokens, depth, x, y, z=0.0):
        """tokens (n_views, patches, feat_dim) fp16-valued; depth (n_views, patches) uint16-valued."""
        pts = self.n_views * self.patches
        t = np.asarray(tokens).astype(np.float16).reshape(pts, self.feat_dim)
        d = np.asarray(depth).astype(np.uint16).reshape(pts)
        self.keys.append(str(key))
        self._tok.append(t)
        self._dep.append(d)
        self._pose.append((float(x), float(y), float(z)))

    def close(self):
        n, pts = len(self.keys), self.n_views * self.patches
        header = {"n": n, "n_views": self.n_views, "patches": self.patches, "feat_dim": self.feat_dim, "keys": self.keys}
        hb = json.dumps(header).encode("utf-8")
        head_len = _pad(16 + len(hb))
        pose_off = head_len
        dep_off = pose_off + _pad(n * 3 * 8)
        tok_off = dep_off + _pad(n * pts * 2)
        with open(self.path, "wb") as f:
            f.write(MAGIC + struct.pack("<II", VERSION, len(hb)) + hb)
            f.write(b"\0" * (head_len - 16 - len(hb)))
            pose = np.asarray(self._pose, np.float64).reshape(n, 3)
            f.write(pose.tobytes())
            f.write(b"\0" * (dep_off - pose_off - pose.nbytes))
            for d in self._dep:
                f.write(d.tobytes())
            f.write(b"\0" * (tok_off - dep_off - n * pts * 2))
            for t in self._tok:
                f.write(t.tobytes())
        return self.path


class PackedStore:
    def __init__(self, path):
        with open(path, "rb") as f:
            head = f.read(16)
            if head[:8] != MAGIC:
                raise ValueError("%s: not a GMMSTORE file" % path)
            version, hlen = struct.unpack("<II", head[8:16])
            if version != VERSION:
                raise ValueError("%s: store version %d, expected %d" % (path, version, VERSION))
            h = json.loads(f.read(hlen).decode("utf-8"))
        self.n, self.n_views, self.patches, self.feat_dim = h["n"], h["n_views"], h["patches"], h["feat_dim"]
        self.keys = h["keys"]
        self.index = {k: i for i, k in enumerate(self.keys)}
        pts = self.n_views * self.patches
        head_len = _pad(16 + hlen)
        pose_off = head_len
        dep_off = pose_off + _pad(self.n * 3 * 8)
        tok_off = dep_off + _pad(self.n * pts * 2)
        self.pose = np.memmap(path, np.float64, "r", pose_off, (self.n, 3))
        self.depth = np.memmap(path, np.uint16, "r", dep_off, (self.n, pts))
        self.tokens = np.memmap(path, np.float16, "r", tok_off, (self.n, pts, self.feat_dim))

    def __len__(self):
        return self.n

    def __contains__(self, key):
        return key in self.index

    def get(self, key):
        """-> (depth (pts,) uint16, tokens (pts, D) fp16, (x, y, z) python floats); zero-copy views."""
        i = self.index[key]
        p = self.pose[i]
        return self.depth[i], self.tokens[i], (float(p[0]), float(p[1]), float(p[2]))

    def gather(self, keys):
        """Observations of a batch of "<scan>_<vp>" keys in the layout of GridMemoryBatch.step():
        depth (B, pts) uint16, feats (B, pts, D) fp16, poses [(x, y)] python floats."""
        ids = [self.index[k] for k in keys]
        depth = np.stack([self.depth[i] for i in ids])
        feats = np.stack([self.tokens[i] for i in ids])
        poses = [(float(self.pose[i, 0]), float(self.pose[i, 1])) for i in ids]
        return depth, feats, poses


class DeviceStore:
    """The packed store resident in HBM (row f1 + f3 together): tokens (n, pts, D) fp16, depth (n, pts) uint16 (or fp32
    metres for VLN-CE geometry) and poses on the device; an environment step then moves NO observation bytes over PCIe --
    `append(mem, keys)` gathers the B observations straight into the grid memory's next slot (device copy) and returns
    the depth rows + host poses that GridMemoryBatch.step() takes.  288 GB of HBM hold ~39 000 native observations
    (588 x 768 fp16 = 0.9 MB each: all of R2R's 10 567 panoramas take 9.6 GB) or ~40 000 BASELINE-shape ones."""

    def __init__(self, keys, tokens, depth, poses, device):
        import torch
        self.keys = list(keys)
        self.index = {k: i for i, k in enumerate(self.keys)}
        self.device = torch.device(device)
        self._u16 = np.dtype(depth.dtype) == np.uint16
        # (uint16 rows are gathered through an int16 view of the same bytes: index_select has no uint16 kernel)
        self.tokens = self._upload(tokens, torch.float16, np.float16)
        self.depth = self._upload(depth, torch.int16 if self._u16 else torch.float32, np.int16 if self._u16 else np.float32)
        self.poses = [tuple(float(v) for v in p) for p in poses]

    def _upload(self, src, tdtype, ndtype, chunk=512):
        """Host array (possibly a read-only mmap of the packed store) -> device tensor, `chunk` records at a time through
        a writable staging copy: bounded host memory whatever the store's size."""
        import torch
        n = len(src)
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=tdtype, device=self.device)
        for i in range(0, n, chunk):
            blk = np.array(src[i:i + chunk])                  # writable copy of this block
            blk = blk.view(ndtype) if blk.dtype.itemsize == np.dtype(ndtype).itemsize and blk.dtype != ndtype else blk.astype(ndtype, copy=False)
            out[i:i + chunk].copy_(torch.from_numpy(blk))
        return out

    @classmethod
    def from_packed(cls, store, device):
        """PackedStore (mmap) -> HBM, one upload."""
        return cls(store.keys, store.tokens, store.depth, [tuple(store.pose[i]) for i in range(store.n)], device)

    def _upload_index(self, ids):
        """Row ids -> device through a small ring of pinned buffers (asynchronous: a `torch.tensor(ids, device=...)` is a pageable
        upload, i.e. the host would wait for everything queued on the stream -- a whole backward pass in a training loop)."""
        import torch
        if torch.device(self.device).type != "cuda":
            return torch.tensor(ids, dtype=torch.int64, device=self.device)
        ring = getattr(self, "_idx_ring", None)
        if ring is None or ring.numel < len(ids):
            from .pinned import PinnedRing
            ring = self._idx_ring = PinnedRing(max(len(ids), 64), dtype=torch.int64, max_slots=16)
        k = ring.acquire()
        host = ring.host(k)
        host[:len(ids)] = torch.as_tensor(ids, dtype=torch.int64)
        dev = torch.empty(len(ids), dtype=torch.int64, device=self.device)
        dev.copy_(host[:len(ids)], non_blocking=True)
        ring.record(k)
        return dev

    def append(self, mem, keys):
        """Write the observations of `keys` (one per episode of the lock-step batch) into mem.next_slot(); returns
        (depth (B, pts) device tensor, [(x, y)] host floats) for mem.step(depth, None, poses, headings)."""
        import torch
        idx = self._upload_index([self.index[k] for k in keys])
        slot = mem.next_slot()
        slot.copy_(self.tokens.index_select(0, idx).view(slot.shape))       # device-side gather, no PCIe
        d = self.depth.index_select(0, idx)
        return (d.view(torch.uint16) if self._u16 else d), [(self.poses[self.index[k]][0], self.poses[self.index[k]][1]) for k in keys]


DEPTH_W = 128
SAMPLE_IDX = [9, 27, 45, 63, 81, 99, 117]        # env.py:279


def reference_record(clip_rows, depth_rows):
    """The exact slices the reference takes from one viewpoint's h5 rows (env.py:80-113, 279-303):
    clip_rows (12, >=50, 768) -> tokens (12, 49, 768) fp16;  depth_rows (36, >=16384) -> depth (12, 49) uint16."""
    sem = np.asarray(clip_rows)[:, :50].astype(np.float16)[:, 1:]
    d = np.asarray(depth_rows)[:, :DEPTH_W * DEPTH_W].astype(np.uint16).reshape(-1, DEPTH_W, DEPTH_W)
    d = d[:, SAMPLE_IDX][:, :, SAMPLE_IDX].reshape(d.shape[0], -1)[12:24]
    return sem, d


def convert_reference_files(clip_h5, depth_h5, viewpoint_info_json, out_path, opener=None):
    """clip_p32.hdf5 + depth.hdf5 + viewpoint_info.json -> one packed store.  `opener(path)` must return a mapping
    key -> array-like: h5py.File when h5py is importable, else gridmm_amd.hdf5_lite.File (pure Python, reads the
    old-style groups + gzip chunks h5py writes by default; pinned on real h5py-written files in tests/golden/hdf5/)."""
    if opener is None:
        try:
            import h5py
            opener = lambda p: h5py.File(p, "r")   # noqa: E731
        except ImportError:
            from . import hdf5_lite
            opener = hdf5_lite.File
    info = json.load(open(viewpoint_info_json))
    clip, depth = opener(clip_h5), opener(depth_h5)
    w = PackedStoreWriter(out_path)
    for key in sorted(info):
        if key not in clip or key not in depth:
            continue
        tokens, d = reference_record(clip[key][...], depth[key][...])
        w.add(key, tokens, d, info[key]["x"], info[key]["y"], info[key].get("z", 0.0))
    return w.close()
