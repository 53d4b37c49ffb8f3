"""Device-resident grid memory: the HIP-backed mirror of the reference's per-episode map state.

Reference (relative to /root/reference):
  state lists                     map_nav_src/r2r/env.py:141-151 (global_semantic, global_position_x/y,
                                  global_mask, max/min_x/y, heading, global_map)
  EnvBatch.getGlobalMap           map_nav_src/r2r/env.py:267-374
  EnvBatch.get_gridmap_pos_fts    map_nav_src/r2r/env.py:242-265
  H2D of the whole history/step   map_nav_src/r2r/agent.py:168  (O(t^2) PCIe bytes) -- removed here:
                                  the slab and the point history stay in HBM, only pose/heading
                                  (a few floats per episode) cross PCIe each step.

Layout in HBM for B episodes, capacity `cap = max_steps * pts_per_obs` points each:
  slab       (B, cap, D)  fp16   CLIP patch tokens, appended per step (never re-copied)
  hist_x/y   (B, cap)     fp32   world XY of every point
  hist_valid (B, cap)     uint8  depth != 0
  cell_id    (B, cap)     int16  current egocentric cell (x*14+y) or -1       -> `grid_map`
  perm       (B, cap)     int32  point indices sorted by (cell, index)
  cell_start (B, 198)     int32  per-cell ranges of perm
  bbox (B,4), half_len (B,), pos_fts (B,196,5)
"""
import math
import os

import numpy as np
import torch

from . import ops
from .synthetic import GridGeometry, NATIVE  # noqa: F401  (geometry description only)


class GridMemoryBatch:
    MAX_BIN_SLICES = 16

    def __init__(self, batch_size, geom=NATIVE, max_steps=16, device="cuda"):
        """max_steps: observations per episode the slab holds.  A rollout appends max_action_len + 1 of them (one from
        env.reset(), one after every action incl. the last, r2r/agent.py:268-451): 16 for the default max_action_len = 15."""
        self.B, self.geom, self.max_steps = batch_size, geom, max_steps
        self.device = torch.device(device)
        self.n_new = geom.pts_per_obs
        self.cap = max_steps * self.n_new
        B, cap, dev = batch_size, self.cap, self.device
        self.slab = torch.zeros(B, cap, geom.feat_dim, dtype=torch.float16, device=dev)
        self.slab._gridmm_epoch = [0]
        self.hist_x = torch.zeros(B, cap, dtype=torch.float32, device=dev)
        self.hist_y = torch.zeros(B, cap, dtype=torch.float32, device=dev)
        self.hist_valid = torch.zeros(B, cap, dtype=torch.uint8, device=dev)
        self.cell_id = torch.full((B, cap), -1, dtype=torch.int16, device=dev)
        self.perm = torch.zeros(B, cap, dtype=torch.int32, device=dev)
        self.cell_start = torch.zeros(B, 198, dtype=torch.int32, device=dev)
        self.bbox = torch.empty(B, 4, dtype=torch.float32, device=dev)
        self.half_len = torch.zeros(B, dtype=torch.float32, device=dev)
        self.pos_fts = torch.zeros(B, 196, 5, dtype=torch.float32, device=dev)
        self.n_pts = torch.zeros(B, dtype=torch.int32, device=dev)
        self._bin_ws = torch.empty(B, self.MAX_BIN_SLICES * 17, 197, dtype=torch.int32, device=dev)
        self.n_pts_host = np.zeros(B, np.int64)
        self.keep_for_backward = False
        self._rel = None                   # per-point relevance kept across the steps of an episode (relevance_cache)
        # static per-step inputs (pinned host -> device): the only bytes that cross PCIe each step.  ONE staging buffer
        # (pose | cos/sin(-heading) | active flags | per-episode view tables (VLN-CE) | EXTRA bytes for the caller, e.g.
        # graph.GraphedNavStep's fused-logit index maps) and ONE async copy per step: every H2D is a ~4 us copy kernel on
        # the step's stream (rocprofv3: six of them per step were 1 % of the headline step)
        nv = geom.n_views if geom.vlnce else 0
        self.STAGE_EXTRA = 1 << 16
        o_pose, o_head, o_act = 0, B * 8, B * 16
        o_vc = (o_act + B + 15) // 16 * 16
        o_vs = o_vc + B * nv * 4
        self._stage_extra_off = (o_vs + B * nv * 4 + 63) // 64 * 64
        nbytes = self._stage_extra_off + self.STAGE_EXTRA
        self._stage_host = torch.zeros(nbytes, dtype=torch.uint8)
        if dev.type == "cuda":
            self._stage_host = self._stage_host.pin_memory()
        self._stage_dev = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self._stage_used = self._stage_extra_off           # bytes uploaded per step (grows when the extra area is used)

        def views(buf):
            f = lambda o, n, shape: buf[o:o + 4 * n].view(torch.float32).view(*shape)   # noqa: E731
            return (f(o_pose, B * 2, (B, 2)), f(o_head, B * 2, (B, 2)), buf[o_act:o_act + B],
                    f(o_vc, B * nv, (B, nv)) if nv else None, f(o_vs, B * nv, (B, nv)) if nv else None)
        self._pose_host, self._head_host, self._act_host, self._vcos_host, self._vsin_host = views(self._stage_host)
        self.pose_d, self.head_d, self.act_d, vcos_d, vsin_d = views(self._stage_dev)
        self._act_host.fill_(1)
        self.act_d.fill_(1)
        self._active = None
        self._h2d_done = None
        # A RING of pinned host buffers behind the one device buffer (the device side is overwritten in stream order; only a
        # pinned source has to outlive its copy): with a single buffer set_pose() waits for the previous step's copy, which sits
        # behind everything queued before it -- in a training loop the whole backward of the previous iteration.  Slot 0 is the
        # buffer above; callers that claimed bytes of its EXTRA area (stage_extra) keep the single-buffer behaviour.
        self._views_of = views
        self._stage_ring = None            # pinned.PinnedRing, created with the first step on a CUDA device (slot 0 = the buffer above)
        self._stage_slot = 0
        self.STAGE_RING = 16
        self._cmax_event = None
        self._cmax_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._cmax_host = torch.zeros(1, dtype=torch.int32).pin_memory() if dev.type == "cuda" else torch.zeros(1, dtype=torch.int32)
        self._bbox_init = torch.tensor([-10000.0, 10000.0, -10000.0, 10000.0], device=dev).repeat(B, 1)
        # host-side constants, rounded exactly as NumPy rounds them in env.py:118, 290
        P = geom.patches
        base = np.array([(2 * c + 1 - P) / P for c in range(P)] * P, np.float32)
        self.x_off = torch.from_numpy(base * np.float32(geom.tan_half_fov)).to(dev)
        self._view_ang = [v * math.pi / (geom.n_views / 2) for v in range(geom.n_views)]
        self.flags = ops.FLAG_VLNCE if geom.vlnce else 0
        if geom.vlnce:   # per-episode view tables: angle = v*pi/6 - heading (Policy_ViewSelection_GridMap.py:734)
            self.view_cos, self.view_sin = vcos_d, vsin_d
        else:
            ang = self._view_ang
            self.view_cos = torch.tensor([np.float32(math.cos(a)) for a in ang], dtype=torch.float32, device=dev)
            self.view_sin = torch.tensor([np.float32(math.sin(a)) for a in ang], dtype=torch.float32, device=dev)
        self.reset()

    def reset(self):
        """EnvBatch.newEpisodes (env.py:178-194).  Device-only work (graph-capturable).

        With `self.keep_for_backward` set (training rollouts), the feature slab of the finished rollout stays
        untouched -- autograd nodes of that rollout still read it in backward -- and a fresh one is allocated."""
        if self.keep_for_backward or getattr(self.slab, "_gridmm_in_graph", False):
            self.slab = torch.zeros_like(self.slab)      # a live autograd graph reads the old rows (autograd._GridAggregate)
            self.slab._gridmm_epoch = [0]
        else:
            self.slab._gridmm_epoch[0] += 1               # rows are recycled in place: stale backward passes must fail
        self.bbox.copy_(self._bbox_init)
        self.n_pts.zero_()
        self.n_pts_host[:] = 0
        self.cell_id.fill_(-1)
        self._cmax_event = None
        if self._rel is not None:                         # the rows are recycled: no point has a relevance value any more
            self._rel["valid"].zero_()
            self._rel["slab"] = self.slab

    # ---- per-point relevance kept across the steps of an episode (two-pass aggregation shapes: D = 768, long instructions)
    relevance_cache_enabled = True
    relevance_cache_in_graphs = False     # whole-step hipGraphs (graph.GraphedNavStep) bake the host-side key check in: opt-in

    def relevance_cache(self, key, keep_alive=None):
        """State of gridmm_grid_aggregate_incremental for this memory: w_j = max_l <x_j, text_l> (vilmodel.py:797-798) depends
        only on a point's slab row and on the instruction, so the values of the points appended at earlier steps are kept
        ('hist', by history index; 'valid' = leading points per episode that have one) and a step computes the new
        observation's only.  `key` identifies the instruction side (the text tensor + the text_proj parameters, their
        versions): a different key, reset() or a replaced slab clears 'valid' -- a stream-ordered memset, the launch path
        itself decides everything on the device.  Rows of the slab must not be rewritten once a step has used them (the
        rows a step appends are recomputed whatever 'valid' says)."""
        st = self._rel
        if st is None:
            from . import _lib
            B, cap, dev = self.B, self.cap, self.device
            n = int(_lib.load().gridmm_grid_aggregate_incremental_scratch(B, cap))
            st = self._rel = {"hist": torch.zeros(B, cap, dtype=torch.float32, device=dev),
                              "valid": torch.zeros(B, dtype=torch.int32, device=dev),
                              "rel": torch.zeros(B, cap, dtype=torch.float32, device=dev),
                              "scratch": torch.empty(n, dtype=torch.uint8, device=dev), "key": None, "slab": self.slab,
                              "alive": None, "clears": 0}
        if st["key"] != key or st["slab"] is not self.slab:
            st["valid"].zero_()
            st["key"], st["slab"], st["alive"] = key, self.slab, keep_alive
            st["clears"] += 1
            st["streak"] = st.get("streak", 0) + 1
            if st["streak"] >= 4:
                # a caller that hands over a NEW instruction tensor at every call (e.g. it re-runs forward('language') per
                # step) gets nothing from keeping values and pays the two bookkeeping launches: back to the plain passes
                self.relevance_cache_enabled = False
        else:
            st["streak"] = 0
        return st

    # ---- host half of a step: a few floats per episode into static (pinned -> device) buffers
    def set_pose(self, poses, headings, active=None):
        """poses: B x (x, y) python floats (viewpoint_info, env.py:286); headings: B python floats.
        Rounded to fp32 on the host exactly as NumPy does (env.py:118-120, 344-348)."""
        self._cmax_event = None                   # a new pose re-bins the memory: the tracked cell count is stale
        if self._stage_used == self._stage_extra_off and self.device.type == "cuda":
            # a slot of the pinned ring whose copy has completed (a loop that reads results back every step keeps using one or two
            # slots; a training loop running ahead of the device grows the ring up to STAGE_RING -- pinned allocations cost
            # milliseconds, so never eagerly)
            if self._stage_ring is None:
                from .pinned import PinnedRing
                self._stage_ring = PinnedRing(self._stage_host.numel(), max_slots=self.STAGE_RING, first=self._stage_host)
                if self._h2d_done is not None:
                    self._stage_ring.record(0, self._h2d_done)
            k = self._stage_ring.acquire()
            if k != self._stage_slot or self._stage_ring.host(k) is not self._stage_host:
                self._stage_slot, self._stage_host = k, self._stage_ring.host(k)
                self._pose_host, self._head_host, self._act_host, self._vcos_host, self._vsin_host = self._views_of(self._stage_host)
        elif self._h2d_done is not None:
            self._h2d_done.synchronize()          # the previous step's async H2D has consumed the pinned buffers
        # whole-array writes into the pinned buffers; cos / sin stay libm scalars in double (math.cos, as the reference
        # computes them) and are rounded to fp32 once, exactly like np.float32(math.cos(a))
        self._pose_host.numpy()[:] = np.asarray([(pz[0], pz[1]) for pz in poses], dtype=np.float64).astype(np.float32)
        ang = [(-hd + math.pi) if self.geom.vlnce else -hd for hd in headings]       # env.py:337 / VLN-CE :785
        self._head_host.numpy()[:] = np.array([(math.cos(a), math.sin(a)) for a in ang], dtype=np.float64).astype(np.float32)
        if self.geom.vlnce:
            rel = [[a0 - hd for a0 in self._view_ang] for hd in headings]
            self._vcos_host.numpy()[:] = np.array([[math.cos(a) for a in r] for r in rel], dtype=np.float64).astype(np.float32)
            self._vsin_host.numpy()[:] = np.array([[math.sin(a) for a in r] for r in rel], dtype=np.float64).astype(np.float32)
        if active is None:
            self._active = None
        else:
            self._act_host.copy_(torch.from_numpy(np.asarray(active, bool).astype(np.uint8)))
            self._active = np.asarray(active, bool)
        n = self._stage_used
        self._stage_dev[:n].copy_(self._stage_host[:n], non_blocking=True)       # the step's one H2D
        if self.device.type == "cuda":
            self._h2d_done = torch.cuda.Event()
            self._h2d_done.record()
            if self._stage_ring is not None:
                self._stage_ring.record(self._stage_slot, self._h2d_done)

    def stage_extra(self, nbytes):
        """(pinned host view, device view) of `nbytes` of the staging buffer's caller area: whatever the caller writes into
        the host view before set_pose() / step() travels with the pose in the same copy.  Regions are never reused: a
        second claimant (another GraphedNavStep on this memory) gets the bytes behind the first one's."""
        o = (self._stage_used + 15) // 16 * 16        # bump allocator: every caller gets its own region
        if o + nbytes > self._stage_extra_off + self.STAGE_EXTRA:
            raise ValueError("grid memory staging buffer: %d bytes requested, %d of %d already claimed"
                             % (nbytes, o - self._stage_extra_off, self.STAGE_EXTRA))
        self._stage_used = o + nbytes
        return self._stage_host[o:o + nbytes], self._stage_dev[o:o + nbytes]

    # ---- device half: kernel launches only (replayable from a hipGraph)
    def project_and_bin(self, depth):
        """depth (B, n_views*ppv) uint16 on the device.  Uses the pose/heading set by set_pose()."""
        self._cmax_event = None                   # only step() re-records the count behind these launches (cmax_hint)
        ops.grid_project(depth, self.x_off, self.view_cos, self.view_sin, self.pose_d, self.n_pts, self.hist_x,
                         self.hist_y, self.hist_valid, self.bbox, self.half_len, self.pos_fts,
                         None if self._active is None else self.act_d,
                         self.geom.n_views, self.geom.patches ** 2, self.geom.depth_div, self.flags,
                         self.geom.max_dist)
        # one workgroup per episode for small memories, else 8 / 16 slices of the history on their own workgroups
        # (tools/bench_bin.py: N = 7056: 33 -> 18 us, N = 105840: 274 -> 46 us).  Host-known depth: the choice is static
        # inside a captured graph.
        n_after = int(self.n_pts_host.max()) + self.n_new
        slices = int(os.environ.get("GRIDMM_BIN_SLICES", 0)) or (1 if n_after < 3000 else 8 if n_after < 60000 else 16)
        ops.grid_bin(self.hist_x, self.hist_y, self.hist_valid, self.n_pts, self.pose_d, self.head_d, self.half_len,
                     self.cell_id, self.perm, self.cell_start, self.flags, workspace=self._bin_ws, slices=slices)

    def step(self, depth, feats, poses, headings, active=None):
        """Append one observation per episode and re-bin the whole history (getGlobalMap for all i).

        depth  (B, n_views*ppv) uint16  sampled patch-centre depth (host or device)
        feats  (B, n_views*ppv, D) fp16 patch tokens (host or device), or None when the producer has
               already written them in place into `next_slot()` (zero-copy append)
        poses  B x (x, y) python floats; headings B python floats
        active optional B bools: inactive episodes are left untouched
        """
        B, dev, n_new = self.B, self.device, self.n_new
        act_host = np.ones(B, bool) if active is None else np.asarray(active, bool)
        if (self.n_pts_host[act_host] + n_new > self.cap).any():
            raise ValueError("grid memory capacity exceeded (max_steps=%d)" % self.max_steps)
        depth = _to_device(depth, dev)
        if self.geom.vlnce:
            depth = depth.to(torch.float32)                     # habitat depth, metres
        elif depth.dtype != torch.uint16:
            depth = depth.to(torch.int32).to(torch.uint16)
        depth = depth.reshape(B, n_new).contiguous()
        if feats is not None:
            feats = _to_device(feats, dev).reshape(B, n_new, self.geom.feat_dim)
            if act_host.all() and (self.n_pts_host == self.n_pts_host[0]).all():
                n0 = int(self.n_pts_host[0])
                self.slab[:, n0:n0 + n_new].copy_(feats)
            else:
                for b in np.nonzero(act_host)[0]:
                    n0 = int(self.n_pts_host[b])
                    self.slab[b, n0:n0 + n_new].copy_(feats[b])
        self.set_pose(poses, headings, active)
        self.project_and_bin(depth)
        self.n_pts_host[act_host] += n_new            # host mirror of the device-side counter
        self._cmax_event = None
        if self.track_cmax and dev.type == "cuda":
            # the batch's largest occupied-cell count (the reference's max_cell_num, vilmodel.py:809-823) goes to a pinned
            # word behind the binning kernels: by the time the caller has collated the navigation inputs it is on the host,
            # and the varlen path picks its sequence bucket without stalling the stream (cmax_hint)
            ops.grid_cell_count_max(self.cell_start, self._cmax_dev)      # one launch (was five torch ops per step)
            self._cmax_host.copy_(self._cmax_dev, non_blocking=True)
            self._cmax_event = torch.cuda.Event()
            self._cmax_event.record()
        return self.pos_fts

    track_cmax = False

    def cmax_hint(self):
        """Largest occupied-cell count over the batch after the last step(), or None when it was not tracked -- or when
        the memory has been re-binned since by another route (set_pose / project_and_bin / reset, a graph replay after
        set_pose): the count then belongs to an older binning and callers must read the occupancy of their own call."""
        if self._cmax_event is None:
            return None
        self._cmax_event.synchronize()
        return int(self._cmax_host[0])

    def points_upper_bound(self):
        """Host-known bound of the points per episode after the step in flight (graph replays append one observation
        to the restored history without touching the host mirror)."""
        return min(self.cap, int(self.n_pts_host.max()) + self.n_new)

    def next_slot(self):
        """(B, n_new, D) view of the slab where the next observation's tokens go (lock-step batches)."""
        n0 = int(self.n_pts_host[0])
        assert (self.n_pts_host == n0).all(), "next_slot() needs lock-step episodes"
        return self.slab[:, n0:n0 + self.n_new]

    # ---- the reference's observation form (env.py:610-612), for drop-in callers and tests
    def grid_fts(self, b):
        return self.slab[b, :int(self.n_pts_host[b])]

    def grid_map(self, b):
        return self.cell_id[b, :int(self.n_pts_host[b])].to(torch.float64)

    def as_reference_obs(self):
        return ([self.grid_fts(b) for b in range(self.B)], [self.grid_map(b) for b in range(self.B)],
                self.pos_fts)


def _to_device(x, dev):
    """Host -> device: asynchronous only from PINNED memory.  An asynchronous copy from pageable memory is not ordered
    against the host on ROCm (the runtime may pin the source in place and copy it after the call has returned): a caller
    that passes a temporary array would have it freed under the copy."""
    t = torch.as_tensor(x)
    return t.to(dev, non_blocking=bool(t.device.type == "cpu" and t.is_pinned()))


def pack_reference_lists(grid_fts, grid_map):
    """List form (agent.py:168: per-episode (N_b, D) fp16 + (N_b,) float64 ids) -> packed slab + sorted lists."""
    B = len(grid_fts)
    dev = grid_fts[0].device
    D = grid_fts[0].shape[1]
    n = [int(t.shape[0]) for t in grid_fts]
    cap = max(max(n), 1)
    slab = torch.zeros(B, cap, D, dtype=torch.float16, device=dev)
    ids = torch.full((B, cap), -1, dtype=torch.int16, device=dev)
    for b in range(B):
        if n[b]:
            slab[b, :n[b]].copy_(grid_fts[b].to(torch.float16))
            ids[b, :n[b]].copy_(grid_map[b].to(torch.int16))
    n_pts = torch.tensor(n, dtype=torch.int32, device=dev)
    perm = torch.empty(B, cap, dtype=torch.int32, device=dev)
    cell_start = torch.empty(B, 198, dtype=torch.int32, device=dev)
    ops.grid_sort_ids(ids, n_pts, perm, cell_start)
    return slab, perm, cell_start
