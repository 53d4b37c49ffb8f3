// Training counterparts of the row-wise stages: the backward of the logit masking / fusion, the backward of the cell
// compaction, and hidden-state dropout -- so that the differentiable forward (gridmm_amd/vilmodel_train.py) has no stock
// torch gather / where / dropout on the path SURVEY.md §8 covers.  Reference: map_nav_src/models/vilmodel.py.
#include "common.h"

namespace {

// One workgroup per episode.  Forward (fuse_logits_kernel, rowops.hip; vilmodel.py:859-899):
//   fw = sigmoid(fuse_raw);  global_j = g_j fw (-inf if visited / padded);  grid_j = gr_j (same mask)
//   local_k = l_k (1 - fw) (-inf unless navigable);  fused_j = global_j + add_j,
//   add_0 = local_0;  add_j = local_{con[j]} (con >= 0) | sum_{k >= 1, cand_visited[k]} local_k (con == -1) | 0 (con == -2)
// Backward: masked positions pass no gradient (masked_fill); a masked local that was added to a fused logit made it
// -inf, whose incoming gradient is 0 from any softmax / cross-entropy, so it is treated as 0 as well.
__global__ __launch_bounds__(64) void fuse_logits_bwd_kernel(
    const float* __restrict__ g_raw, const float* __restrict__ l_raw, const float* __restrict__ fuse_raw,
    const uint8_t* __restrict__ gmap_masks, const uint8_t* __restrict__ gmap_visited,
    const uint8_t* __restrict__ vp_nav_masks, const int32_t* __restrict__ cand_of_node,
    const uint8_t* __restrict__ cand_visited, const float* __restrict__ d_global, const float* __restrict__ d_local,
    const float* __restrict__ d_grid, const float* __restrict__ d_fused, float* __restrict__ d_g_raw,
    float* __restrict__ d_l_raw, float* __restrict__ d_grid_raw, float* __restrict__ d_fuse_raw, int G, int V) {
  extern __shared__ float sm[];          // dL[V]: total gradient on the masked local logits
  float* dL = sm;
  __shared__ float s_part[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float fw = fuse_raw ? 1.0f / (1.0f + expf(-fuse_raw[b])) : 0.5f;
  for (int k = tid; k < V; k += blockDim.x) dL[k] = d_local ? d_local[b * V + k] : 0.f;
  __syncthreads();
  float dfw = 0.f, dbw = 0.f;            // dbw: gradient on the sum of visited candidates' local logits
  // nodes: one thread per node; the scatter of d_fused into dL is done afterwards by one thread in node order
  // (G, V <= a few dozen; no float atomics: bit-reproducible whatever cand_of_node looks like)
  for (int j = tid; j < G; j += blockDim.x) {
    const bool ok = gmap_masks[b * G + j] && !gmap_visited[b * G + j];
    const float df = d_fused ? d_fused[b * G + j] : 0.f;
    const float dG = ok ? (d_global ? d_global[b * G + j] : 0.f) + df : 0.f;
    d_g_raw[b * G + j] = dG * fw;
    dfw += dG * g_raw[b * G + j];
    d_grid_raw[b * G + j] = (ok && d_grid) ? d_grid[b * G + j] : 0.f;
    if (j > 0 && cand_of_node[b * G + j] == -1) dbw += df;
  }
  if (tid == 0 && d_fused) {
    dL[0] += d_fused[b * G];
    for (int j = 1; j < G; ++j) {
      const int k = cand_of_node[b * G + j];
      if (k >= 0) dL[k] += d_fused[b * G + j];
    }
  }
  s_part[tid] = dbw;
  __syncthreads();
  float dbw_all = 0.f;
  for (int i = 0; i < (int)blockDim.x; ++i) dbw_all += s_part[i];     // fixed order: deterministic
  __syncthreads();
  for (int k = tid; k < V; k += blockDim.x) {
    float d = dL[k];
    if (k >= 1 && cand_visited[b * V + k]) d += dbw_all;
    if (!vp_nav_masks[b * V + k]) d = 0.f;
    d_l_raw[b * V + k] = d * (1.0f - fw);
    dfw -= d * l_raw[b * V + k];
  }
  s_part[tid] = dfw;
  __syncthreads();
  if (tid == 0 && d_fuse_raw) {
    float s = 0.f;
    for (int i = 0; i < (int)blockDim.x; ++i) s += s_part[i];
    d_fuse_raw[b] = s * fw * (1.0f - fw);
  }
}

// d_cells[b][c] = d_out[b][rank(c)] for occupied cells (rank = number of occupied cells before c), 0 for empty ones:
// the backward of the compaction of vilmodel.py:813-823 (the same gradient reaches grid_proj's output and the position
// embedding, which are summed before the compaction).
__global__ __launch_bounds__(256) void cells_compact_bwd_kernel(const float* __restrict__ d_out, int64_t d_out_bs,
                                                                const uint8_t* __restrict__ occ,
                                                                float* __restrict__ d_cells, int H) {
  __shared__ int s_rank[GRIDMM_CELLS];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    int base = 0;
    for (int c0 = 0; c0 < GRIDMM_CELLS; c0 += 64) {
      const int c = c0 + lane;
      const bool o = (c < GRIDMM_CELLS) && occ[b * GRIDMM_CELLS + c];
      const unsigned long long m = __ballot(o);
      if (c < GRIDMM_CELLS) s_rank[c] = o ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
      base += __popcll(m);
    }
  }
  __syncthreads();
  const int nv = H >> 2;
  const int c_lo = blockIdx.y * GRIDMM_GRID;
  for (int i = tid; i < GRIDMM_GRID * nv; i += blockDim.x) {
    const int c = c_lo + i / nv, q = i % nv;
    const int r = s_rank[c];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= 0) v = reinterpret_cast<const float4*>(d_out + (size_t)b * d_out_bs + (size_t)r * H)[q];
    reinterpret_cast<float4*>(d_cells + ((size_t)b * GRIDMM_CELLS + c) * H)[q] = v;
  }
}

// y = keep(seed, i) ? x / (1 - p) : 0 -- hidden-state dropout (BertSelfOutput / BertOutput / embeddings,
// vilmodel.py:86,166,205; transformer.py dropout1 / dropout2) from the same counter-based hash as the attention dropout
// (common.h): the backward is the same call on the incoming gradient with the same seed, nothing is stored.
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n4, float p,
                               unsigned long long seed, const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
  const float scale = 1.0f / (1.0f - p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const unsigned int e = (unsigned int)(i * 4);
    float4 o;
    o.x = dropout_keep(seed, e, p) ? v.x * scale : 0.f;
    o.y = dropout_keep(seed, e + 1, p) ? v.y * scale : 0.f;
    o.z = dropout_keep(seed, e + 2, p) ? v.z * scale : 0.f;
    o.w = dropout_keep(seed, e + 3, p) ? v.w * scale : 0.f;
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

// y = r + dropout(x) (p == 0: y = r + x; r == NULL: y = dropout(x)) [+ the bf16 hi / lo planes of y]: the residual sums of a
// pre-LayerNorm transformer layer (transformer.py:176-182: src = src + dropout1(src2)) and the gradient sums of its backward
// as ONE pass.  Two roundings, product then sum (no FMA contraction): bit-identical to gridmm_dropout followed by an fp32 add.
__global__ void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ r, float* __restrict__ y, size_t n4,
                                   float p, unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                   unsigned short* __restrict__ Ph, unsigned short* __restrict__ Pl) {
  if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
  const float scale = 1.0f / (1.0f - p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 o = reinterpret_cast<const float4*>(x)[i];
    if (p > 0.f) {
      const unsigned int e = (unsigned int)(i * 4);
      o.x = dropout_keep(seed, e, p) ? __fmul_rn(o.x, scale) : 0.f;
      o.y = dropout_keep(seed, e + 1, p) ? __fmul_rn(o.y, scale) : 0.f;
      o.z = dropout_keep(seed, e + 2, p) ? __fmul_rn(o.z, scale) : 0.f;
      o.w = dropout_keep(seed, e + 3, p) ? __fmul_rn(o.w, scale) : 0.f;
    }
    if (r) {
      const float4 b = reinterpret_cast<const float4*>(r)[i];
      o.x = __fadd_rn(b.x, o.x); o.y = __fadd_rn(b.y, o.y); o.z = __fadd_rn(b.z, o.z); o.w = __fadd_rn(b.w, o.w);
    }
    if (y) reinterpret_cast<float4*>(y)[i] = o;
    if (Ph) {
      uint2 hi, lo;
      split2_bf16(o.x, o.y, hi.x, lo.x);
      split2_bf16(o.z, o.w, hi.y, lo.y);
      reinterpret_cast<uint2*>(Ph)[i] = hi;
      reinterpret_cast<uint2*>(Pl)[i] = lo;
    }
  }
}

// ---- Linear layers with a handful of input features (the position / angle embeddings: K = 7 for loc_fts and the map /
// viewpoint position features, K = 14 for their concatenation; map_nav_src/models/vilmodel.py:454-470, 538-552, 640-655).  Too
// narrow for the MFMA tile GEMMs (their fp32-A fallback walks 64-wide tiles over a 7-deep contraction: 35-86 us per call);
// plain fp32 FMAs over the rows do it at the speed of writing Y / reading dY.
constexpr int SK_MAX = 16;
// Y[m][n] = b[n] + sum_k X[m][k] W[n][k].  Thread = 4 consecutive columns (W rows in registers), workgroup = 256 threads
// over the columns x SK_ROWS rows.
constexpr int SK_ROWS = 8;
__global__ __launch_bounds__(256) void linear_skinny_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                            const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                            int M, int N, int K) {
  __shared__ float s_x[SK_ROWS][SK_MAX];
  const int n0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int m0 = blockIdx.y * SK_ROWS;
  if (threadIdx.x < SK_ROWS * SK_MAX) {
    const int r = threadIdx.x / SK_MAX, k = threadIdx.x % SK_MAX;
    s_x[r][k] = (m0 + r < M && k < K) ? X[(size_t)(m0 + r) * ldx + k] : 0.f;
  }
  __syncthreads();
  if (n0 >= N) return;
  float w[4][SK_MAX];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < SK_MAX; ++k) w[j][k] = (k < K) ? W[(size_t)(n0 + j) * K + k] : 0.f;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) b4 = *reinterpret_cast<const float4*>(bias + n0);
  for (int r = 0; r < SK_ROWS && m0 + r < M; ++r) {
    float y[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int k = 0; k < SK_MAX; ++k) {
      const float x = s_x[r][k];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = fmaf(x, w[j][k], y[j]);
    }
    *reinterpret_cast<float4*>(Y + (size_t)(m0 + r) * ldy + n0) = make_float4(y[0], y[1], y[2], y[3]);
  }
}

// dW[n][k] = sum_m dY[m][n] X[m][k], db[n] = sum_m dY[m][n]: stage 1, one partial per 64-row block and column
// (part[blk][n][K + 1], the last entry = the bias partial); stage 2 sums the blocks in order (deterministic).
constexpr int SKB_ROWS = 64;
__global__ __launch_bounds__(256) void linear_skinny_bwd_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ X,
                                                                int ldx, float* __restrict__ part, int M, int N, int K) {
  __shared__ float s_x[SKB_ROWS][SK_MAX];
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int m0 = blockIdx.y * SKB_ROWS, rows = min(SKB_ROWS, M - m0);
  for (int i = threadIdx.x; i < SKB_ROWS * SK_MAX; i += 256) {
    const int r = i / SK_MAX, k = i % SK_MAX;
    s_x[r][k] = (r < rows && k < K) ? X[(size_t)(m0 + r) * ldx + k] : 0.f;
  }
  __syncthreads();
  if (n >= N) return;
  float acc[SK_MAX + 1];
#pragma unroll
  for (int k = 0; k <= SK_MAX; ++k) acc[k] = 0.f;
  for (int r = 0; r < rows; ++r) {
    const float d = dY[(size_t)(m0 + r) * ldy + n];
#pragma unroll
    for (int k = 0; k < SK_MAX; ++k) acc[k] = fmaf(d, s_x[r][k], acc[k]);
    acc[SK_MAX] += d;
  }
  float* p = part + ((size_t)blockIdx.y * N + n) * (K + 1);
#pragma unroll
  for (int k = 0; k < SK_MAX; ++k)
    if (k < K) p[k] = acc[k];
  p[K] = acc[SK_MAX];
}
__global__ void linear_skinny_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, float* __restrict__ db,
                                                int n_blk, int N, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over N * (K + 1)
  if (i >= N * (K + 1)) return;
  float s = 0.f;
  for (int b = 0; b < n_blk; ++b) s += part[(size_t)b * N * (K + 1) + i];
  const int n = i / (K + 1), k = i % (K + 1);
  if (k < K) { if (dW) dW[(size_t)n * K + k] = s; }
  else if (db) db[n] = s;
}

// ---- Linear layers with ONE output feature (the last Linear of the navigation heads' ClsPrediction, H -> 1;
// map_nav_src/models/vilmodel.py:437-446): a dot product per row.  Forward: a wave per row.  Backward: dX[m][:] = dY[m] w,
// dw = sum_m dY[m] X[m][:], db = sum_m dY[m] -- a workgroup per 64 rows writes one partial (K + 1 floats, the layout of
// linear_skinny_bwd_reduce_kernel with N = 1), summed in block order.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ Y, int M, int K) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int c = lane * 4; c < K; c += 256) {
    const float4 x = *reinterpret_cast<const float4*>(X + (size_t)row * ldx + c);
    const float4 v = *reinterpret_cast<const float4*>(w + c);
    s += (x.x * v.x + x.y * v.y) + (x.z * v.z + x.w * v.w);
  }
  s = wave_sum(s);
  if (lane == 0) Y[row] = s + (bias ? bias[0] : 0.f);
}
constexpr int RD_ROWS = 64;
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ X, int ldx,
                                                         const float* __restrict__ w, float* __restrict__ dX, int lddx,
                                                         float* __restrict__ part, int M, int K) {
  const int m0 = blockIdx.x * RD_ROWS, rows = min(RD_ROWS, M - m0);
  float* p = part + (size_t)blockIdx.x * (K + 1);
  for (int c = threadIdx.x * 4; c < K; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(w + c);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < rows; ++r) {
      const float d = dY[m0 + r];
      const float4 x = *reinterpret_cast<const float4*>(X + (size_t)(m0 + r) * ldx + c);
      a.x = fmaf(d, x.x, a.x); a.y = fmaf(d, x.y, a.y); a.z = fmaf(d, x.z, a.z); a.w = fmaf(d, x.w, a.w);
      if (dX) *reinterpret_cast<float4*>(dX + (size_t)(m0 + r) * lddx + c) = make_float4(d * v.x, d * v.y, d * v.z, d * v.w);
    }
    p[c] = a.x; p[c + 1] = a.y; p[c + 2] = a.z; p[c + 3] = a.w;
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += dY[m0 + r];
    p[K] = s;
  }
}

}  // namespace

extern "C" int gridmm_rowdot(const float* X, int ldx, const float* w, const float* bias, float* Y, int M, int K,
                             gridmm_stream_t stream) {
  if (!X || !w || !Y || M <= 0 || K <= 0 || K % 4 || ldx < K || ldx % 4) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(rowdot_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), X, ldx, w, bias, Y, M, K);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" size_t gridmm_rowdot_bwd_workspace(int M, int K) {
  return (size_t)((M + RD_ROWS - 1) / RD_ROWS) * (K + 1) * sizeof(float);
}

extern "C" int gridmm_rowdot_bwd(const float* dY, const float* X, int ldx, const float* w, float* dX, int lddx, float* dw,
                                 float* db, float* workspace, int M, int K, gridmm_stream_t stream) {
  if (!dY || !X || !w || !workspace || M <= 0 || K <= 0 || K % 4 || ldx < K || ldx % 4 || (dX && (lddx < K || lddx % 4)))
    return GRIDMM_EINVAL;
  const int n_blk = (M + RD_ROWS - 1) / RD_ROWS;
  hipStream_t st = as_stream(stream);
  GRIDMM_LAUNCH(rowdot_bwd_kernel, dim3(n_blk), dim3(256), 0, st, dY, X, ldx, w, dX, lddx, workspace, M, K);
  GRIDMM_CHECK_LAUNCH();
  if (dw || db) {
    GRIDMM_LAUNCH(linear_skinny_bwd_reduce_kernel, dim3((K + 1 + 255) / 256), dim3(256), 0, st, workspace, dw, db, n_blk, 1, K);
    GRIDMM_CHECK_LAUNCH();
  }
  return GRIDMM_OK;
}

extern "C" int gridmm_linear_skinny(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N,
                                    int K, gridmm_stream_t stream) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || N % 4 || K <= 0 || K > SK_MAX || ldx < K || ldy < N || ldy % 4) return GRIDMM_EINVAL;
  dim3 grid((N / 4 + 255) / 256, (M + SK_ROWS - 1) / SK_ROWS);
  GRIDMM_LAUNCH(linear_skinny_kernel, grid, dim3(256), 0, as_stream(stream), X, ldx, W, bias, Y, ldy, M, N, K);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" size_t gridmm_linear_skinny_bwd_workspace(int M, int N, int K) {
  return (size_t)((M + SKB_ROWS - 1) / SKB_ROWS) * N * (K + 1) * sizeof(float);
}

extern "C" int gridmm_linear_skinny_bwd(const float* dY, int ldy, const float* X, int ldx, float* dW, float* db, float* workspace,
                                        int M, int N, int K, gridmm_stream_t stream) {
  if (!dY || !X || !workspace || (!dW && !db) || M <= 0 || N <= 0 || K <= 0 || K > SK_MAX || ldx < K || ldy < N)
    return GRIDMM_EINVAL;
  const int n_blk = (M + SKB_ROWS - 1) / SKB_ROWS;
  hipStream_t st = as_stream(stream);
  GRIDMM_LAUNCH(linear_skinny_bwd_kernel, dim3((N + 255) / 256, n_blk), dim3(256), 0, st, dY, ldy, X, ldx, workspace, M, N, K);
  GRIDMM_CHECK_LAUNCH();
  const int tot = N * (K + 1);
  GRIDMM_LAUNCH(linear_skinny_bwd_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, workspace, dW, db, n_blk, N, K);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_dropout_add(const float* x, const float* r, float* y, void* y_hi, void* y_lo, int64_t n, float p,
                                  unsigned long long seed, const unsigned long long* seed_dev, gridmm_stream_t stream) {
  if (!x || (!y && !y_hi) || (y_hi && !y_lo) || n <= 0 || (n & 3) || n >= (1ll << 32) || p < 0.f || p >= 1.f)
    return GRIDMM_EINVAL;
  const size_t n4 = (size_t)n / 4;
  unsigned blocks = (unsigned)((n4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  GRIDMM_LAUNCH(dropout_add_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, r, y, n4, p, seed, seed_dev,
                (unsigned short*)y_hi, (unsigned short*)y_lo);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_fuse_logits_bwd(const float* g_raw, const float* l_raw, const float* fuse_raw,
                                      const uint8_t* gmap_masks, const uint8_t* gmap_visited,
                                      const uint8_t* vp_nav_masks, const int32_t* cand_of_node,
                                      const uint8_t* cand_visited, const float* d_global, const float* d_local,
                                      const float* d_grid, const float* d_fused, float* d_g_raw, float* d_l_raw,
                                      float* d_grid_raw, float* d_fuse_raw, int B, int G, int V,
                                      gridmm_stream_t stream) {
  if (B <= 0 || G <= 0 || V <= 0 || V > 4096 || !g_raw || !l_raw || !d_g_raw || !d_l_raw || !d_grid_raw ||
      (fuse_raw && !d_fuse_raw))
    return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(fuse_logits_bwd_kernel, dim3(B), dim3(64), V * sizeof(float), as_stream(stream), g_raw, l_raw, fuse_raw,
                gmap_masks, gmap_visited, vp_nav_masks, cand_of_node, cand_visited, d_global, d_local, d_grid, d_fused,
                d_g_raw, d_l_raw, d_grid_raw, d_fuse_raw, G, V);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_cells_compact_bwd(const float* d_out, int64_t d_out_bs, const uint8_t* occ, float* d_cells, int B,
                                        int H, gridmm_stream_t stream) {
  if (B <= 0 || H <= 0 || H % 4 || d_out_bs % 4 || !d_out || !occ || !d_cells) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(cells_compact_bwd_kernel, dim3(B, GRIDMM_GRID), dim3(256), 0, as_stream(stream), d_out, d_out_bs, occ,
                d_cells, H);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_dropout(const float* x, float* y, int64_t n, float p, unsigned long long seed,
                              const unsigned long long* seed_dev, gridmm_stream_t stream) {
  if (n <= 0 || n % 4 || n >= ((int64_t)1 << 32) || !(p >= 0.f && p < 1.f) || !x || !y) return GRIDMM_EINVAL;
  const size_t n4 = (size_t)n / 4;
  unsigned blocks = (unsigned)((n4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  GRIDMM_LAUNCH(dropout_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, n4, p, seed, seed_dev);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
