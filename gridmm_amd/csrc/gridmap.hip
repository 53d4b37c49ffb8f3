// Grid-memory projection ("fill_gridmap"): back-projection of a new observation, running bbox,
// map scale, cell-centre features, per-step egocentric re-binning of the whole history and the
// stable counting sort that turns cell ids into per-cell point lists.
//
// Bit-exactness contract: every fp32 operation is done in the reference's
// order with one rounding per operation -- this file is compiled with -ffp-contract=off (and
// repeats it as a pragma) so no product feeds an FMA; cos/sin arrive as host-rounded fp32;
// float->int conversion reproduces x86 cvttss2si (NaN/inf/out-of-range -> INT_MIN), which is
// what NumPy's .astype(np.int32) does on the reference's host.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int NBIN = GRIDMM_CELLS + 1;  // 196 cells + 1 bin for invalid-depth points
constexpr int SORT_WAVES = 16;          // 1024-thread workgroup per episode

__device__ __forceinline__ int trunc_x86(float v) {
  if (!(v > -2147483648.0f && v < 2147483648.0f)) return INT32_MIN;
  return (int)v;
}

// ---------------------------------------------------------------------------------------------
// project: one workgroup per episode
// ---------------------------------------------------------------------------------------------
constexpr int FLAG_VLNCE = 1;  // VLN-CE twin: gy = -ry + y; map_x = -(tx cos + ty sin); (x, Z, y) cell features

constexpr int PROJ_THREADS = 1024;   // one workgroup per episode: 16 waves walk the observation's points (7056 at the BASELINE shape)

template <bool DEPTH_F32>
__global__ __launch_bounds__(PROJ_THREADS) void grid_project_kernel(
    const void* __restrict__ depth_, const float* __restrict__ x_off,
    const float* __restrict__ view_cos, const float* __restrict__ view_sin, int view_stride,
    const float* __restrict__ pose, int32_t* __restrict__ n_old, float* __restrict__ hist_x,
    float* __restrict__ hist_y, uint8_t* __restrict__ hist_valid, float* __restrict__ bbox,
    float* __restrict__ half_len, float* __restrict__ pos_fts, const uint8_t* __restrict__ active,
    int n_views, int ppv, int cap, float depth_div, int flags, float max_dist) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (active && !active[b]) return;
  const int n_new = n_views * ppv, base = n_old[b];
  const float px = pose[2 * b], py = pose[2 * b + 1];
  float mxx = -__builtin_inff(), mnx = __builtin_inff(), mxy = -__builtin_inff(), mny = __builtin_inff();
  for (int i = tid; i < n_new; i += PROJ_THREADS) {
    const int v = i / ppv, p = i - v * ppv;
    float dy;
    bool ok;
    if (DEPTH_F32) {   // VLN-CE: habitat depth in metres, used as is (Policy_ViewSelection_GridMap.py:634)
      dy = reinterpret_cast<const float*>(depth_)[(size_t)b * n_new + i];
      ok = dy != 0.f;
    } else {
      const uint16_t d = reinterpret_cast<const uint16_t*>(depth_)[(size_t)b * n_new + i];
      dy = (float)d / depth_div;                 // env.py:116
      ok = d != 0;
    }
    const float dx = dy * x_off[p];              // env.py:118
    const float c = view_cos[b * view_stride + v], s = view_sin[b * view_stride + v];
    const float t0 = dx * c, t1 = dy * s, t2 = dy * c, t3 = dx * s;
    const float gx = (t0 + t1) + px;             // env.py:119, 291
    const float ry = t2 - t3;
    const float gy = (flags & FLAG_VLNCE) ? (-ry) + py : ry + py;   // env.py:120, 292 / VLN-CE :739
    const size_t o = (size_t)b * cap + base + i;
    hist_x[o] = gx;
    hist_y[o] = gy;
    hist_valid[o] = ok;
    mxx = fmaxf(mxx, gx); mnx = fminf(mnx, gx);
    mxy = fmaxf(mxy, gy); mny = fminf(mny, gy);
  }
  __shared__ float s_red[PROJ_THREADS / 64][4];
  __shared__ float s_half;
  mxx = wave_max(mxx); mnx = wave_min(mnx); mxy = wave_max(mxy); mny = wave_min(mny);
  if ((tid & 63) == 0) {
    s_red[tid >> 6][0] = mxx; s_red[tid >> 6][1] = mnx; s_red[tid >> 6][2] = mxy; s_red[tid >> 6][3] = mny;
  }
  __syncthreads();
  if (tid == 0) {
    float bb[4] = {bbox[4 * b + 0], bbox[4 * b + 1], bbox[4 * b + 2], bbox[4 * b + 3]};
    for (int w = 0; w < PROJ_THREADS / 64; ++w) {
      bb[0] = fmaxf(bb[0], s_red[w][0]); bb[1] = fminf(bb[1], s_red[w][1]);
      bb[2] = fmaxf(bb[2], s_red[w][2]); bb[3] = fminf(bb[3], s_red[w][3]);
    }
    for (int k = 0; k < 4; ++k) bbox[4 * b + k] = bb[k];
    // env.py:322-331
    const float ax = px - bb[1], bx = bb[0] - px;
    const float xh = ax > bx ? ax : bx;
    const float ay = py - bb[3], by = bb[2] - py;
    const float yh = ay > by ? ay : by;
    const float h = xh > yh ? xh : yh;
    const float hl = (h * 2.0f) / 3.0f;
    half_len[b] = hl;
    s_half = hl;
  }
  __syncthreads();
  if (tid == 0) n_old[b] = base + n_new;  // the history now holds the new observation (every read of n_old is above)
  // env.py:242-265 -- row i*14+j <-> cell (x=i, y=j)
  if (tid < GRIDMM_CELLS) {
    const float hl = s_half;
    const float cell_len = (hl * 2.0f) / 14.0f;
    const int i = tid / GRIDMM_GRID, j = tid - i * GRIDMM_GRID;
    const float bx = ((float)i * cell_len - hl) + cell_len / 2.0f;
    const float by = ((float)j * cell_len - hl) + cell_len / 2.0f;
    float* o = pos_fts + ((size_t)b * GRIDMM_CELLS + tid) * 5;
    if (flags & FLAG_VLNCE) {
      // vlnce_baselines/models/utils.py:125-144 reads points as (x, Z, y): the cell's j coordinate is an elevation
      if (bx == 0.f && by == 0.f) {
        o[0] = 0.f; o[1] = 1.f; o[2] = 0.f; o[3] = 1.f; o[4] = 0.f;
      } else {
        float xy = sqrtf(bx * bx);
        xy = xy >= 1e-8f ? xy : 1e-8f;
        float xyz = sqrtf(bx * bx + by * by);
        xyz = xyz >= 1e-8f ? xyz : 1e-8f;
        const float hd = asinf(bx / xy), el = asinf(by / xyz);
        o[0] = sinf(hd); o[1] = cosf(hd); o[2] = sinf(el); o[3] = cosf(el); o[4] = xyz / max_dist;
      }
    } else {
      float dist = sqrtf(bx * bx + by * by);
      dist = dist >= 1e-8f ? dist : 1e-8f;
      float hd = asinf(bx / dist);
      if (by < 0.f) hd = 3.14159265358979323846f - hd;
      o[0] = sinf(hd);
      o[1] = cosf(hd);
      o[2] = 0.f;
      o[3] = 1.f;
      o[4] = dist / max_dist;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bin + stable counting sort: one 1024-thread workgroup per episode; each wave owns a
// contiguous slice of the history so ranks are deterministic (ascending point index per cell,
// the order of the reference's boolean-mask gather, vilmodel.py:802).
// ---------------------------------------------------------------------------------------------
template <bool COMPUTE>
__global__ __launch_bounds__(1024) void grid_bin_sort_kernel(
    const float* __restrict__ hist_x, const float* __restrict__ hist_y,
    const uint8_t* __restrict__ hist_valid, const int32_t* __restrict__ n_pts,
    const float* __restrict__ pose, const float* __restrict__ head_cs,
    const float* __restrict__ half_len, int16_t* __restrict__ cell_id, int32_t* __restrict__ perm,
    int32_t* __restrict__ cell_start, int cap, int flags) {
  __shared__ int s_cur[SORT_WAVES][NBIN];
  __shared__ int s_start[NBIN + 1];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = n_pts[b];
  const int per_wave = ((n + SORT_WAVES * 64 - 1) / (SORT_WAVES * 64)) * 64;  // multiple of 64
  const int lo = wave * per_wave, hi = min(n, lo + per_wave);

  for (int i = tid; i < SORT_WAVES * NBIN; i += 1024) (&s_cur[0][0])[i] = 0;
  __syncthreads();

  float px = 0.f, py = 0.f, c = 0.f, s = 0.f, hl = 0.f, two_h = 0.f;
  if (COMPUTE) {
    px = pose[2 * b]; py = pose[2 * b + 1];
    c = head_cs[2 * b]; s = head_cs[2 * b + 1];
    hl = half_len[b]; two_h = 2.0f * hl;
  }
  int16_t* ids = cell_id + (size_t)b * cap;

  // pass 1: ids + per-wave histograms
  for (int i = lo + lane; i < hi; i += 64) {
    int id;
    if (COMPUTE) {
      const size_t o = (size_t)b * cap + i;
      const float tx = hist_x[o] - px, ty = hist_y[o] - py;                  // env.py:344-345
      const float a0 = tx * c, a1 = ty * s, a2 = ty * c, a3 = tx * s;
      const float sx = a0 + a1, my = a2 - a3;                                 // env.py:347-348
      const float mx = (flags & FLAG_VLNCE) ? -sx : sx;                       // VLN-CE :797
      int cx = trunc_x86(((mx + hl) / two_h) * 13.0f);                        // env.py:349
      int cy = trunc_x86(((my + hl) / two_h) * 13.0f);                        // env.py:351
      cx = cx < 0 ? 0 : (cx > 13 ? 13 : cx);                                  // env.py:353-357
      cy = cy < 0 ? 0 : (cy > 13 ? 13 : cy);
      id = hist_valid[o] ? cx * GRIDMM_GRID + cy : -1;                        // env.py:359-369
      ids[i] = (int16_t)id;
    } else {
      id = ids[i];
      if (id < -1 || id >= GRIDMM_CELLS) id = -1;
    }
    atomicAdd(&s_cur[wave][id < 0 ? GRIDMM_CELLS : id], 1);
  }
  __syncthreads();

  // exclusive scan: over waves within a bin, then over bins
  if (tid < NBIN) {
    int run = 0;
    for (int w = 0; w < SORT_WAVES; ++w) {
      const int v = s_cur[w][tid];
      s_cur[w][tid] = run;
      run += v;
    }
    s_start[tid] = run;  // total of the bin, scanned below
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < NBIN; ++k) {
      const int v = s_start[k];
      s_start[k] = run;
      run += v;
    }
    s_start[NBIN] = run;  // == n
  }
  __syncthreads();
  if (tid <= NBIN) cell_start[(size_t)b * (NBIN + 1) + tid] = s_start[tid];

  // pass 2: stable scatter, 64 points at a time per wave, in point order
  int32_t* pm = perm + (size_t)b * cap;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (int i0 = lo; i0 < hi; i0 += 64) {
    const int i = i0 + lane;
    const bool in = i < hi;
    int id = in ? (int)ids[i] : -2;
    if (in && (id < 0 || id >= GRIDMM_CELLS)) id = GRIDMM_CELLS;
    unsigned long long todo = __ballot(in);
    int rank = 0, cnt = 0;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int lid = __shfl(id, leader, 64);
      const unsigned long long m = __ballot(in && id == lid);
      if (in && id == lid) {
        rank = __popcll(m & lt_mask);
        cnt = __popcll(m);
      }
      todo &= ~m;
    }
    if (in) {
      const int base = s_start[id] + s_cur[wave][id];
      pm[base + rank] = i;
    }
    __builtin_amdgcn_wave_barrier();
    if (in && rank == 0) s_cur[wave][id] += cnt;
    __builtin_amdgcn_wave_barrier();
  }
}


// ---------------------------------------------------------------------------------------------
// The same sort for deep memories (tens of thousands of points per episode), cut into S slices per episode so that
// S x B workgroups work on it instead of B: histogram | scatter (the scan over slices and bins is the scatter's prologue).  Wave w of slice s owns the contiguous
// sub-slice (16 s + w) of the history, so the order inside a cell is still ascending point index.
// ws [B][S][16 + 1][NBIN] int32: rows 0..15 = exclusive prefix of the slice's 16 per-wave histograms, row 16 = the
// slice's total per bin (S rows per bin: short).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void slice_range(int n, int S, int gw, int& lo, int& hi) {
  const int per_wave = ((n + S * SORT_WAVES * 64 - 1) / (S * SORT_WAVES * 64)) * 64;   // multiple of 64
  lo = gw * per_wave;
  hi = min(n, lo + per_wave);
}

template <bool COMPUTE>
__global__ __launch_bounds__(1024) void grid_bin_hist_kernel(
    const float* __restrict__ hist_x, const float* __restrict__ hist_y, const uint8_t* __restrict__ hist_valid,
    const int32_t* __restrict__ n_pts, const float* __restrict__ pose, const float* __restrict__ head_cs,
    const float* __restrict__ half_len, int16_t* __restrict__ cell_id, int32_t* __restrict__ ws, int cap, int flags) {
  __shared__ int s_cur[SORT_WAVES][NBIN];
  const int sl = blockIdx.x, S = gridDim.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int lo, hi;
  slice_range(n_pts[b], S, sl * SORT_WAVES + wave, lo, hi);
  for (int i = tid; i < SORT_WAVES * NBIN; i += 1024) (&s_cur[0][0])[i] = 0;
  __syncthreads();
  float px = 0.f, py = 0.f, c = 0.f, s = 0.f, hl = 0.f, two_h = 0.f;
  if (COMPUTE) {
    px = pose[2 * b]; py = pose[2 * b + 1];
    c = head_cs[2 * b]; s = head_cs[2 * b + 1];
    hl = half_len[b]; two_h = 2.0f * hl;
  }
  int16_t* ids = cell_id + (size_t)b * cap;
  for (int i = lo + lane; i < hi; i += 64) {
    int id;
    if (COMPUTE) {   // the arithmetic of grid_bin_sort_kernel, operation for operation (bit-exact cell ids)
      const size_t o = (size_t)b * cap + i;
      const float tx = hist_x[o] - px, ty = hist_y[o] - py;                  // env.py:344-345
      const float a0 = tx * c, a1 = ty * s, a2 = ty * c, a3 = tx * s;
      const float sx = a0 + a1, my = a2 - a3;                                 // env.py:347-348
      const float mx = (flags & FLAG_VLNCE) ? -sx : sx;                       // VLN-CE :797
      int cx = trunc_x86(((mx + hl) / two_h) * 13.0f);                        // env.py:349
      int cy = trunc_x86(((my + hl) / two_h) * 13.0f);                        // env.py:351
      cx = cx < 0 ? 0 : (cx > 13 ? 13 : cx);                                  // env.py:353-357
      cy = cy < 0 ? 0 : (cy > 13 ? 13 : cy);
      id = hist_valid[o] ? cx * GRIDMM_GRID + cy : -1;                        // env.py:359-369
      ids[i] = (int16_t)id;
    } else {
      id = ids[i];
      if (id < -1 || id >= GRIDMM_CELLS) id = -1;
    }
    atomicAdd(&s_cur[wave][id < 0 ? GRIDMM_CELLS : id], 1);
  }
  __syncthreads();
  int32_t* out = ws + ((size_t)b * S + sl) * (SORT_WAVES + 1) * NBIN;
  if (tid < NBIN) {                      // exclusive scan over the 16 sub-slices of this slice (point order)
    int run = 0;
    for (int w = 0; w < SORT_WAVES; ++w) {
      const int v = s_cur[w][tid];
      out[w * NBIN + tid] = run;
      run += v;
    }
    out[SORT_WAVES * NBIN + tid] = run;
  }
}

// The scan over (slice, bin) totals is redone by every scatter workgroup in its prologue (S x 197 ints from L2 and a
// 197-wide prefix: ~1 us) instead of a launch of its own between the histogram and the scatter (8.6 us of launch, ramp
// and one-workgroup-per-episode work); slice 0 of an episode writes cell_start.
__global__ __launch_bounds__(1024) void grid_bin_scatter_kernel(const int16_t* __restrict__ cell_id,
                                                                const int32_t* __restrict__ n_pts,
                                                                const int32_t* __restrict__ ws,
                                                                int32_t* __restrict__ perm,
                                                                int32_t* __restrict__ cell_start, int cap) {
  __shared__ int s_cur[SORT_WAVES][NBIN];
  __shared__ int s_tot[256], s_base[NBIN];
  const int sl = blockIdx.x, S = gridDim.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int lo, hi;
  slice_range(n_pts[b], S, sl * SORT_WAVES + wave, lo, hi);
  const size_t stride = (size_t)(SORT_WAVES + 1) * NBIN;
  const int32_t* tot = ws + (size_t)b * S * stride + SORT_WAVES * NBIN;     // row 16 of slice 0: per-bin totals of a slice
  int before = 0, all = 0;               // points of this bin in the slices before mine / in all slices
  if (tid < 256) {
    if (tid < NBIN)
      for (int r = 0; r < S; ++r) {
        const int v = tot[r * stride + tid];
        if (r < sl) before += v;
        all += v;
      }
    s_tot[tid] = all;
  }
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {    // inclusive prefix over the bins (Hillis-Steele, 8 steps)
    int v = 0;
    if (tid < 256 && tid >= d) v = s_tot[tid - d];
    __syncthreads();
    if (tid < 256) s_tot[tid] += v;
    __syncthreads();
  }
  if (tid < NBIN) {
    const int start = s_tot[tid] - all;  // exclusive: first sorted position of the bin
    s_base[tid] = start + before;        // absolute base of this slice in the bin
    if (sl == 0) {
      cell_start[(size_t)b * (NBIN + 1) + tid] = start;
      if (tid == NBIN - 1) cell_start[(size_t)b * (NBIN + 1) + NBIN] = s_tot[tid];   // == n
    }
  }
  __syncthreads();
  const int32_t* in_ws = ws + ((size_t)b * S + sl) * stride;
  for (int i = tid; i < SORT_WAVES * NBIN; i += 1024)          // write cursor = slice base of the bin + wave prefix
    (&s_cur[0][0])[i] = in_ws[i] + s_base[i % NBIN];
  __syncthreads();
  const int16_t* ids = cell_id + (size_t)b * cap;
  int32_t* pm = perm + (size_t)b * cap;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (int i0 = lo; i0 < hi; i0 += 64) {       // stable scatter, 64 points at a time per wave, in point order
    const int i = i0 + lane;
    const bool in = i < hi;
    int id = in ? (int)ids[i] : -2;
    if (in && (id < 0 || id >= GRIDMM_CELLS)) id = GRIDMM_CELLS;
    unsigned long long todo = __ballot(in);
    int rank = 0, cnt = 0;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int lid = __shfl(id, leader, 64);
      const unsigned long long m = __ballot(in && id == lid);
      if (in && id == lid) {
        rank = __popcll(m & lt_mask);
        cnt = __popcll(m);
      }
      todo &= ~m;
    }
    if (in) pm[s_cur[wave][id] + rank] = i;
    __builtin_amdgcn_wave_barrier();
    if (in && rank == 0) s_cur[wave][id] += cnt;
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

extern "C" int gridmm_grid_project(const void* depth, int depth_f32, const float* x_off, const float* view_cos,
                                   const float* view_sin, int view_stride, const float* pose, int32_t* n_old,
                                   float* hist_x, float* hist_y, uint8_t* hist_valid, float* bbox,
                                   float* half_len, float* pos_fts, const uint8_t* active, int B,
                                   int n_views, int ppv, int cap, float depth_div, int flags, float max_dist,
                                   gridmm_stream_t stream) {
  if (B <= 0 || n_views <= 0 || ppv <= 0 || cap < n_views * ppv || (view_stride != 0 && view_stride != n_views))
    return GRIDMM_EINVAL;
  if (depth_f32)
    GRIDMM_LAUNCH(grid_project_kernel<true>, dim3(B), dim3(PROJ_THREADS), 0, as_stream(stream), depth, x_off, view_cos,
                  view_sin, view_stride, pose, n_old, hist_x, hist_y, hist_valid, bbox, half_len, pos_fts, active,
                  n_views, ppv, cap, depth_div, flags, max_dist);
  else
    GRIDMM_LAUNCH(grid_project_kernel<false>, dim3(B), dim3(PROJ_THREADS), 0, as_stream(stream), depth, x_off, view_cos,
                  view_sin, view_stride, pose, n_old, hist_x, hist_y, hist_valid, bbox, half_len, pos_fts, active,
                  n_views, ppv, cap, depth_div, flags, max_dist);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_grid_bin(const float* hist_x, const float* hist_y, const uint8_t* hist_valid,
                               const int32_t* n_pts, const float* pose, const float* head_cs,
                               const float* half_len, int16_t* cell_id, int32_t* perm,
                               int32_t* cell_start, int B, int cap, int flags, gridmm_stream_t stream) {
  if (B <= 0 || cap <= 0) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH((grid_bin_sort_kernel<true>), dim3(B), dim3(1024), 0, as_stream(stream), hist_x,
                hist_y, hist_valid, n_pts, pose, head_cs, half_len, cell_id, perm, cell_start, cap, flags);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

namespace {
// cmax[0] = max over the episodes of the number of non-empty cells (one workgroup: B x 196 comparisons, no atomics)
__global__ __launch_bounds__(256) void cell_count_max_kernel(const int32_t* __restrict__ cell_start, int32_t* __restrict__ cmax,
                                                             int B) {
  int best = 0;
  for (int b = 0; b < B; ++b) {
    const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
    const int c = threadIdx.x;
    const int n = __syncthreads_count(c < GRIDMM_CELLS && cs[c + 1] > cs[c]);
    best = max(best, n);
  }
  if (threadIdx.x == 0) cmax[0] = best;
}
}  // namespace

extern "C" int gridmm_grid_cell_count_max(const int32_t* cell_start, int32_t* cmax, int B, gridmm_stream_t stream) {
  if (B <= 0 || !cell_start || !cmax) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(cell_count_max_kernel, dim3(1), dim3(256), 0, as_stream(stream), cell_start, cmax, B);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_grid_sort_ids(const int16_t* cell_id, const int32_t* n_pts, int32_t* perm,
                                    int32_t* cell_start, int B, int cap, gridmm_stream_t stream) {
  if (B <= 0 || cap <= 0) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH((grid_bin_sort_kernel<false>), dim3(B), dim3(1024), 0, as_stream(stream), nullptr,
                nullptr, nullptr, n_pts, nullptr, nullptr, nullptr, const_cast<int16_t*>(cell_id), perm,
                cell_start, cap, 0);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_grid_bin_sliced(const float* hist_x, const float* hist_y, const uint8_t* hist_valid,
                                      const int32_t* n_pts, const float* pose, const float* head_cs,
                                      const float* half_len, int16_t* cell_id, int32_t* perm, int32_t* cell_start,
                                      int32_t* workspace, int slices, int B, int cap, int flags,
                                      gridmm_stream_t stream) {
  if (B <= 0 || cap <= 0 || slices < 1 || slices > 64) return GRIDMM_EINVAL;
  if (slices == 1 || !workspace)
    return gridmm_grid_bin(hist_x, hist_y, hist_valid, n_pts, pose, head_cs, half_len, cell_id, perm, cell_start, B,
                           cap, flags, stream);
  hipStream_t st = as_stream(stream);
  GRIDMM_LAUNCH((grid_bin_hist_kernel<true>), dim3(slices, B), dim3(1024), 0, st, hist_x, hist_y, hist_valid, n_pts,
                pose, head_cs, half_len, cell_id, workspace, cap, flags);
  GRIDMM_LAUNCH(grid_bin_scatter_kernel, dim3(slices, B), dim3(1024), 0, st, cell_id, n_pts, workspace, perm, cell_start,
                cap);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
