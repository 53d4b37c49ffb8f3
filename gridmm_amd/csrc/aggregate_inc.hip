// gridmm_grid_aggregate_incremental: the two-pass aggregation (D = 768, or instructions outside the one-pass window at
// D <= 512) for a DEVICE-RESIDENT memory, with the relevance pass restricted to the points that do not have a value yet.
//
// The relevance of a point, w_j = max_l <x_j, text_l> (map_nav_src/models/vilmodel.py:797-798), depends on the point's slab
// row and on the instruction only: both are constant over an episode, while the reference recomputes all N = n_new * t
// values at every step (O(t) per step, O(t^2) per episode) because it rebuilds its tensors from python lists.  Here the
// values live next to the slab in history order (`rel_hist`), `rel_valid[b]` says how many leading points of episode b
// have one, and a step computes only the rest -- normally the n_new points of the observation just appended:
//
//   prep     perm_new[b][i] = start_b + i,  start_b = min(rel_valid[b], n_pts[b] - appended_b): an identity "sorted order"
//            over the points without a value, and a cell table whose only entry the relevance kernels read (the valid
//            count) is their number
//   pass 1   the relevance kernels of the two-pass path, unchanged (aggregate_rel.hip / aggregate_relg.hip), on that order:
//            a point's value does not depend on its tile neighbours (MFMA columns are independent, the K-slices of a
//            token tile are summed in a fixed wave order), so it is bit-identical to the value the full pass computes
//   gather   relevance[b][p] = value of point perm[b][p] (new or kept), the new values are committed to rel_hist
//   pass 2   grid_aggregate_pipe_kernel<.., PREW> over the whole slab, as before: the ONE read of the slab per step
//
// Everything is decided on the device (no host-side validity logic in the launch path), so the call sequence is the same for
// a cold memory (rel_valid = 0: every point is "new", e.g. after reset() or a new instruction: the caller clears rel_valid)
// and a warm one, and it replays from a hipGraph.
#include "common.h"

int gridmm_grid_relevance_wide(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* relevance, int32_t* amax, int B, int cap, int D, int L, int n_chunks,
                               hipStream_t st);
int gridmm_grid_relevance_gemm(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* relevance, int32_t* amax, int B, int cap, int D, int L, int n_chunks,
                               hipStream_t st);
int gridmm_grid_aggregate_prew(const void* slab, const int32_t* perm, const int32_t* cell_start, const float* w,
                               float* cells, uint8_t* occ, float* ws, int B, int cap, int D, int n_chunks,
                               hipStream_t st);

namespace {

constexpr int CS = GRIDMM_CELLS + 2;   // ints per episode in a cell table

__global__ __launch_bounds__(256) void rel_inc_prep_kernel(const int32_t* __restrict__ n_pts,
                                                           const uint8_t* __restrict__ active,
                                                           const int32_t* __restrict__ rel_valid, int n_new, int cap,
                                                           int32_t* __restrict__ perm_new, int32_t* __restrict__ cs_new,
                                                           int32_t* __restrict__ n_start) {
  const int b = blockIdx.y;
  const int n = min(max(n_pts[b], 0), cap);
  const int appended = (!active || active[b]) ? n_new : 0;      // rows [n - appended, n) were written by this step
  const int start = min(max(rel_valid[b], 0), max(n - appended, 0));
  const int cnt = n - start;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) perm_new[(size_t)b * cap + i] = start + i;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < CS; c += 256) cs_new[(size_t)b * CS + c] = c == 0 ? 0 : cnt;
    if (threadIdx.x == 0) n_start[b] = start;
  }
}

__global__ __launch_bounds__(256) void rel_inc_gather_kernel(const int32_t* __restrict__ perm,
                                                             const int32_t* __restrict__ cell_start,
                                                             const int32_t* __restrict__ n_pts,
                                                             const int32_t* __restrict__ n_start,
                                                             const float* __restrict__ rel_new, float* __restrict__ rel_hist,
                                                             int32_t* __restrict__ rel_valid, float* __restrict__ relevance,
                                                             int cap) {
  const int b = blockIdx.y;
  const int n_valid = cell_start[(size_t)b * CS + GRIDMM_CELLS];
  const int start = n_start[b];
  const int n = min(max(n_pts[b], 0), cap);
  const size_t o = (size_t)b * cap;
  // reads of rel_hist touch rows < start, the commit below writes rows >= start: no ordering needed between the two loops
  for (int p = blockIdx.x * 256 + threadIdx.x; p < n_valid; p += gridDim.x * 256) {
    const int idx = perm[o + p];
    relevance[o + p] = idx >= start ? rel_new[o + idx - start] : rel_hist[o + idx];
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n - start; i += gridDim.x * 256) rel_hist[o + start + i] = rel_new[o + i];
  if (blockIdx.x == 0 && threadIdx.x == 0) rel_valid[b] = n;     // (read by the NEXT call's prep kernel only)
}

// full = 1: the relevance pass ran over ALL valid points in sorted order (the plain two-pass call): keep its values
__global__ __launch_bounds__(256) void rel_inc_commit_kernel(const int32_t* __restrict__ perm,
                                                             const int32_t* __restrict__ cell_start,
                                                             const int32_t* __restrict__ n_pts,
                                                             const float* __restrict__ relevance, float* __restrict__ rel_hist,
                                                             int32_t* __restrict__ rel_valid, int cap) {
  const int b = blockIdx.y;
  const int n_valid = cell_start[(size_t)b * CS + GRIDMM_CELLS];
  const size_t o = (size_t)b * cap;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < n_valid; p += gridDim.x * 256) rel_hist[o + perm[o + p]] = relevance[o + p];
  // (points with zero depth never get a cell, so their slot is never read: it needs no value)
  if (blockIdx.x == 0 && threadIdx.x == 0) rel_valid[b] = min(max(n_pts[b], 0), cap);
}

}  // namespace

extern "C" size_t gridmm_grid_aggregate_incremental_scratch(int B, int cap) {
  if (B <= 0 || cap <= 0) return 0;
  // perm_new [B][cap] i32 | rel_new [B][cap] f32 | cs_new [B][198] i32 | n_start [B] i32
  return ((size_t)B * cap * 8 + (size_t)B * (CS + 1) * 4 + 15) / 16 * 16;
}

extern "C" int gridmm_grid_aggregate_incremental(const void* slab, const int32_t* perm, const int32_t* cell_start,
                                                 const void* text_frag, const int32_t* n_pts, const uint8_t* active,
                                                 int n_new, float* rel_hist, int32_t* rel_valid, void* scratch,
                                                 float* cells, uint8_t* occ, float* relevance, void* workspace, int B,
                                                 int cap, int D, int L, int n_chunks, int full, gridmm_stream_t stream) {
  if (B <= 0 || cap <= 0 || L <= 0 || n_new < 0 || n_chunks <= 0 || n_chunks > GRIDMM_CELLS || !workspace || !scratch ||
      !rel_hist || !rel_valid || !relevance || !n_pts)
    return GRIDMM_EINVAL;
  if (D != 768 && L >= 33 && L <= 96) return GRIDMM_EINVAL;     // one-pass shapes: gridmm_grid_aggregate reads the slab once already
  const int Lt = (L + 15) / 16;
  if (Lt > 32) return GRIDMM_EINVAL;
  int32_t* perm_new = static_cast<int32_t*>(scratch);
  float* rel_new = reinterpret_cast<float*>(perm_new + (size_t)B * cap);
  int32_t* cs_new = reinterpret_cast<int32_t*>(rel_new + (size_t)B * cap);
  int32_t* n_start = cs_new + (size_t)B * CS;
  // workspace layout of gridmm_grid_aggregate_workspace: chunk table | split-cell records (the accumulation pass uses the records)
  const size_t table = ((size_t)B * (n_chunks + 1) * sizeof(int32_t) + 15) / 16 * 16;
  float* ws = reinterpret_cast<float*>(static_cast<char*>(workspace) + table);
  hipStream_t st = as_stream(stream);
  const int gx = cap >= 8192 ? 8 : cap >= 2048 ? 4 : 1;
  if (full) {
    // the caller knows that no point has a value (first step of an episode): the plain two passes in sorted order, then
    // one launch that files the values by history index -- two launches fewer than the general sequence below
    int rc = GRIDMM_EINVAL;
    if (D == 768 && L <= 80 && (size_t)cap <= 45000)
      rc = gridmm_grid_relevance_wide(slab, perm, cell_start, text_frag, relevance, nullptr, B, cap, D, L, n_chunks, st);
    if (rc != GRIDMM_OK)
      rc = gridmm_grid_relevance_gemm(slab, perm, cell_start, text_frag, relevance, nullptr, B, cap, D, L, n_chunks, st);
    if (rc != GRIDMM_OK) return rc;
    rc = gridmm_grid_aggregate_prew(slab, perm, cell_start, relevance, cells, occ, ws, B, cap, D, n_chunks, st);
    if (rc != GRIDMM_OK) return rc;
    GRIDMM_LAUNCH(rel_inc_commit_kernel, dim3(2 * gx, B), dim3(256), 0, st, perm, cell_start, n_pts, relevance, rel_hist,
                  rel_valid, cap);
    GRIDMM_CHECK_LAUNCH();
    return GRIDMM_OK;
  }
  GRIDMM_LAUNCH(rel_inc_prep_kernel, dim3(gx, B), dim3(256), 0, st, n_pts, active, rel_valid, n_new, cap, perm_new, cs_new,
                n_start);
  GRIDMM_CHECK_LAUNCH();
  int rc = GRIDMM_EINVAL;
  if (D == 768 && L <= 80 && (size_t)cap <= 45000)
    rc = gridmm_grid_relevance_wide(slab, perm_new, cs_new, text_frag, rel_new, nullptr, B, cap, D, L, n_chunks, st);
  if (rc != GRIDMM_OK)
    rc = gridmm_grid_relevance_gemm(slab, perm_new, cs_new, text_frag, rel_new, nullptr, B, cap, D, L, n_chunks, st);
  if (rc != GRIDMM_OK) return rc;
  GRIDMM_LAUNCH(rel_inc_gather_kernel, dim3(cap >= 8192 ? 16 : cap >= 2048 ? 4 : 1, B), dim3(256), 0, st, perm, cell_start,
                n_pts, n_start, rel_new, rel_hist, rel_valid, relevance, cap);
  GRIDMM_CHECK_LAUNCH();
  return gridmm_grid_aggregate_prew(slab, perm, cell_start, relevance, cells, occ, ws, B, cap, D, n_chunks, st);
}
