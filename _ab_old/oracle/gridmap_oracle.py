"""CPU oracle for the grid-memory projection ("fill_gridmap").  TEST INFRASTRUCTURE.

A NumPy restatement of the reference's per-step top-down grid projection.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this;
the product path (gridmm_amd/) never does.

Restates (file:line relative to /root/reference):
  get_rel_position           map_nav_src/r2r/env.py:115-121
  EnvBatch.getGlobalMap      map_nav_src/r2r/env.py:267-374
  EnvBatch.get_gridmap_pos_fts  map_nav_src/r2r/env.py:242-265
  calculate_vp_rel_pos_fts   map_nav_src/r2r/env.py:60-77
  get_angle_fts              map_nav_src/r2r/env.py:52-58

Parity pin: tests/golden/fill_gridmap_*.npz were produced by importing the
reference itself (oracle/gen_golden.py, numpy 2.2.6 / NEP-50 promotion) and this
file is checked against them bit-for-bit (tests/test_oracle_gridmap.py).

dtype notes (NEP-50, numpy>=2): `f32_array (op) python_float` is computed in
float32 with the scalar rounded to float32 first; `python_float - np.float32`
is float32.  Every rounding below is therefore an explicit np.float32.
The geometry is parameterised (views, patches per side, feature dim) so the
BASELINE 36x196x512 slab uses the same code as the native 12x49x768 one.
"""
import math
from dataclasses import dataclass

import numpy as np

GRID = 14  # GLOBAL_WIDTH == GLOBAL_HEIGHT, env.py:43-44
MAX_DIST = 30.0  # env.py:47


@dataclass(frozen=True)
class GridGeometry:
    """Shape of one observation slab.

    native (reference): n_views=12 horizon views, patches=7 (7x7), feat_dim=768,
    depth_w=128 -> sample indices [9+18k] (env.py:279).
    """
    n_views: int = 12
    patches: int = 7
    feat_dim: int = 768
    depth_w: int = 128
    depth_div: float = 4000.0  # env.py:116
    tan_half_fov: float = math.tan(math.pi / 6)  # env.py:118
    sample_offset: int = -1    # -1: stride // 2 (env.py:279: 9 + 18k); VLN-CE uses 19 + 36k
    max_dist: float = MAX_DIST
    # VLN-CE twin (VLN_CE/vlnce_baselines/models/Policy_ViewSelection_GridMap.py:632-641, 689-825):
    vlnce: bool = False        # depth float32 metres (no /4000); view angle = v*pi/6 - heading; gy = -ry + y;
    #                            re-binning angle = -heading + pi and map_x = -(tx cos + ty sin)

    @property
    def pts_per_view(self):
        return self.patches * self.patches

    @property
    def pts_per_obs(self):
        return self.n_views * self.patches * self.patches

    def sample_index(self):
        stride = self.depth_w // self.patches
        off = stride // 2 if self.sample_offset < 0 else self.sample_offset
        return np.array([off + k * stride for k in range(self.patches)])

    def x_offsets(self):
        """f32 vector of per-patch lateral offsets * tan(fov/2)   (env.py:118)."""
        P = self.patches
        base = np.array([(2 * c + 1 - P) / P for c in range(P)] * P, np.float32)
        return base * self.tan_half_fov  # f32 * python double -> f32

    def view_angles(self):
        """python doubles; (ix-12)*math.pi/6 in the reference (env.py:290) == v*pi/(12/2)."""
        return [v * math.pi / (self.n_views / 2) for v in range(self.n_views)]


NATIVE = GridGeometry()
BASELINE = GridGeometry(n_views=36, patches=14, feat_dim=512)
VLNCE_R2R = GridGeometry(depth_w=256, depth_div=1.0, tan_half_fov=math.tan(math.pi / 4.), sample_offset=19,
                         max_dist=25.0, vlnce=True)
VLNCE_RXR = GridGeometry(depth_w=256, depth_div=1.0, tan_half_fov=math.tan(math.pi * 79. / 360.), sample_offset=19,
                         max_dist=40.0, vlnce=True)


def sample_depth(depth_full, geom=NATIVE, horizon_slice=None):
    """(V_all, W, W[,1]) uint16 -> (n_views, P*P) uint16, row-major patches (env.py:279-281)."""
    d = np.asarray(depth_full)
    if d.ndim == 4:
        d = d[..., 0]
    idx = geom.sample_index()
    d = d[:, idx][:, :, idx].reshape(d.shape[0], -1)
    if horizon_slice is not None:
        d = d[horizon_slice]
    return d


def rel_position(depth_row, angle, geom=NATIVE):
    """env.py:115-121 for one view.  depth_row: (P*P,) uint16 (float32 metres for VLN-CE); angle: python double."""
    depth_y = depth_row.astype(np.float32)
    if not geom.vlnce:
        depth_y = depth_y / np.float32(geom.depth_div)
    depth_x = depth_y * geom.x_offsets()
    c = np.float32(math.cos(angle))
    s = np.float32(math.sin(angle))
    rel_x = depth_x * c + depth_y * s
    rel_y = depth_y * c - depth_x * s
    return rel_x, rel_y


def project_observation(depth_s, pos_x, pos_y, geom=NATIVE, heading=0.0):
    """World XY of one observation's points (env.py:289-294, 306-307; VLN-CE :733-741).

    depth_s: (n_views, P*P) uint16 sampled depth.  Returns gx, gy (n_pts,) f32 and
    valid (n_pts,) bool (depth != 0, env.py:283-285).
    """
    gx, gy = [], []
    px, py = np.float32(pos_x), np.float32(pos_y)
    for v, a in enumerate(geom.view_angles()):
        if geom.vlnce:
            rx, ry = rel_position(depth_s[v], a - heading, geom)     # ix*math.pi/6 - self.headings[i]
            gx.append(rx + px)
            gy.append(-ry + py)                                      # global_y = -rel_y + position["y"]
        else:
            rx, ry = rel_position(depth_s[v], a, geom)
            gx.append(rx + px)
            gy.append(ry + py)
    return np.concatenate(gx), np.concatenate(gy), (depth_s.reshape(-1) != 0)


def trunc_i32(x):
    """float32 -> int32 the way NumPy/x86 does it (cvttss2si): NaN/inf/out-of-range -> INT_MIN."""
    x = np.asarray(x, np.float32)
    ok = np.isfinite(x) & (x > np.float32(-2147483648.0)) & (x < np.float32(2147483648.0))
    out = np.full(x.shape, np.iinfo(np.int32).min, np.int32)
    out[ok] = x[ok].astype(np.int32)
    return out


def gridmap_pos_fts(half_len, max_dist=MAX_DIST):
    """env.py:242-265 (+ :60-77, :52-58).  half_len: np.float32 scalar.  -> (196,5) f32.

    The reference computes cell centres with `half_len` as np.float32 and python
    floats mixed: `half_len*2 / 14` is f32; `i*cell_len - half_len + cell_len/2.` is f32;
    dx**2 on np.float32 stays f32; np.sqrt f32; arcsin f32; `np.pi - heading` f32.
    """
    half_len = np.float32(half_len)
    cell_len = half_len * 2 / GRID
    ang, dist = [], []
    for i in range(GRID):
        for j in range(GRID):
            bx = i * cell_len - half_len + cell_len / 2.
            by = j * cell_len - half_len + cell_len / 2.
            bz = 0.
            dx, dy, dz = bx - 0., by - 0., bz - 0.
            xy = max(np.sqrt(dx ** 2 + dy ** 2), 1e-8)
            xyz = max(np.sqrt(dx ** 2 + dy ** 2 + dz ** 2), 1e-8)
            h = np.arcsin(dx / xy)
            if by < 0.:
                h = np.pi - h
            h -= 0.
            e = np.arcsin(dz / xyz)
            e -= 0.
            ang.append([h, e])
            dist.append([xyz / max_dist])
    ang = np.array(ang).astype(np.float32)
    dist = np.array(dist).astype(np.float32)
    fts = np.vstack([np.sin(ang[:, 0]), np.cos(ang[:, 0]), np.sin(ang[:, 1]), np.cos(ang[:, 1])])
    fts = fts.transpose().astype(np.float32)
    return np.concatenate([fts, dist], 1)


def gridmap_pos_fts_vlnce(half_len, max_dist):
    """VLN-CE twin: Policy_ViewSelection_GridMap.py:661-687 with vlnce_baselines/models/utils.py:125-144,
    whose calculate_vp_rel_pos_fts reads its points as (x, Z, y): the grid's second coordinate lands in the
    ELEVATION (dz = b[1]) and the "y" used for the heading is the constant 0 -- so heading = +-pi/2 and the
    cell's j coordinate only shows up in sin/cos(elevation).  Copied as is, not fixed."""
    half_len = np.float32(half_len)
    cell_len = half_len * 2 / GRID
    ang, dist = [], []
    for i in range(GRID):
        for j in range(GRID):
            bx = i * cell_len - half_len + cell_len / 2.
            bz = j * cell_len - half_len + cell_len / 2.
            dx, dz, dy = bx - 0., bz - 0., 0. - 0.
            if dx == dz == dy == 0:
                ang.append([0, 0]); dist.append([0 / max_dist]); continue
            xy = max(np.sqrt(dx ** 2 + dy ** 2), 1e-8)
            xyz = max(np.sqrt(dx ** 2 + dy ** 2 + dz ** 2), 1e-8)
            h = np.arcsin(dx / xy)
            h -= 0.
            e = np.arcsin(dz / xyz)
            e -= 0.
            ang.append([h, e])
            dist.append([xyz / max_dist])
    ang = np.array(ang).astype(np.float32)
    dist = np.array(dist).astype(np.float32)
    fts = np.vstack([np.sin(ang[:, 0]), np.cos(ang[:, 0]), np.sin(ang[:, 1]), np.cos(ang[:, 1])])
    return np.concatenate([fts.transpose().astype(np.float32), dist], 1)


class GridMemory:
    """Per-episode accumulated memory + per-step egocentric re-binning (env.py:267-374)."""

    def __init__(self, geom=NATIVE):
        self.geom = geom
        self.hist_x, self.hist_y, self.hist_valid, self.hist_fts = [], [], [], []
        # python ints until the first step, np.float32 afterwards (env.py:146-149, 312-319)
        self.max_x, self.min_x, self.max_y, self.min_y = -10000, 10000, -10000, 10000

    def step(self, depth_s, feats, pos_x, pos_y, heading):
        """depth_s (n_views,P*P) u16; feats (n_pts,D) f16; pos python floats; heading float.

        Returns grid_fts (N,D) f16, grid_map (N,) float64 in {-1,0..195}, pos_fts (196,5) f32,
        half_len f32.
        """
        g = self.geom
        gx, gy, valid = project_observation(depth_s, pos_x, pos_y, g, heading)
        self.hist_x.append(gx)
        self.hist_y.append(gy)
        self.hist_valid.append(valid)
        self.hist_fts.append(np.asarray(feats).reshape(-1, g.feat_dim))

        # running bbox over NEW points incl. invalid-depth ones (env.py:312-319)
        if gx.max() > self.max_x: self.max_x = gx.max()
        if gx.min() < self.min_x: self.min_x = gx.min()
        if gy.max() > self.max_y: self.max_y = gy.max()
        if gy.min() < self.min_y: self.min_y = gy.min()

        half_len = self.half_len(pos_x, pos_y)
        cell = self.bin_points(np.concatenate(self.hist_x), np.concatenate(self.hist_y),
                               np.concatenate(self.hist_valid), pos_x, pos_y, heading, half_len, g.vlnce)
        grid_map = cell.astype(np.float64)
        pf = gridmap_pos_fts_vlnce(half_len, g.max_dist) if g.vlnce else gridmap_pos_fts(half_len, g.max_dist)
        return (np.concatenate(self.hist_fts, 0), grid_map, pf, half_len)

    def half_len(self, pos_x, pos_y):
        """env.py:322-331: python float (op) np.float32 -> float32."""
        px, py = np.float32(pos_x), np.float32(pos_y)
        a, b = px - self.min_x, self.max_x - px
        xh = a if a > b else b
        a, b = py - self.min_y, self.max_y - py
        yh = a if a > b else b
        h = xh if xh > yh else yh
        return np.float32(np.float32(h * np.float32(2)) / np.float32(3))

    @staticmethod
    def bin_points(hx, hy, valid, pos_x, pos_y, heading, half_len, vlnce=False):
        """env.py:337-369 (VLN-CE: Policy_ViewSelection_GridMap.py:785-817).  -> int32 cell ids, -1 = invalid."""
        angle = (-heading + math.pi) if vlnce else -heading
        c = np.float32(math.cos(angle))
        s = np.float32(math.sin(angle))
        tx = hx - np.float32(pos_x)
        ty = hy - np.float32(pos_y)
        mx = tx * c + ty * s
        if vlnce:
            mx = -mx
        my = ty * c - tx * s
        two_h = np.float32(2) * half_len
        with np.errstate(divide="ignore", invalid="ignore"):
            cx = trunc_i32((mx + half_len) / two_h * np.float32(GRID - 1))
            cy = trunc_i32((my + half_len) / two_h * np.float32(GRID - 1))
        cx = np.clip(cx, 0, GRID - 1)
        cy = np.clip(cy, 0, GRID - 1)
        cell = cx * GRID + cy
        return np.where(valid, cell, -1).astype(np.int32)
