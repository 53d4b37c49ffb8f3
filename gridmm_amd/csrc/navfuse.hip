// Fused row-wise stages around the encoders of forward('navigation'): everything here is a few hundred KB of fp32 per
// step, so the cost is launches -- each kernel below replaces 3..12 of them (and the torch.cat / slice-assign nodes in
// between).  Reference lines: map_nav_src/models/vilmodel.py.
//   gridmm_cells_embed   :813-823  grid_pos_embeddings (Linear(5,H) + LN) computed inside the cell compaction
//   gridmm_node_embed    :828-833  gmap / vp position embeddings (Linear(7|14,H) + LN) + image embeds + step table,
//                        :846-851  and the byte masks of [cells | nodes | txt] and [nodes | views]
//   gridmm_nav_heads     :859-907  the LN . w tails of the five ClsPrediction heads + masking + logit fusion
#include "common.h"

namespace {

// Row kernels are instantiated for NV float4 per lane and FULL = (H == 256 * NV): without per-chunk guards the loads of a
// phase are issued together (with them: one branch + one s_waitcnt per chunk and operand, ~10 serialised memory round
// trips per row -- rowops.hip).
constexpr int MAXK = 16;     // position feature widths on this path: 5, 7, 14
constexpr int EMB_WAVES = 16;

#define ROW_OK(i) (FULL || lane + (i) * 64 < nv)

// occupied cells in a word of 4 occupancy bytes (any non-zero byte counts)
__device__ __forceinline__ int occ4(uint32_t w) {
  return ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
}

// W^T [K][H] -> LDS, coalesced 128-bit copies by the whole workgroup (K * H % 4 == 0)
__device__ __forceinline__ void stage_wt(const float* __restrict__ WT, int K, int H, float* __restrict__ wT) {
  const int n4 = (K * H) >> 2;
  for (int i = threadIdx.x; i < n4; i += blockDim.x)
    reinterpret_cast<float4*>(wT)[i] = reinterpret_cast<const float4*>(WT)[i];
}

// y[i] (chunk c = lane + 64 i) = LN(W f + b) * gamma + beta for one row, one wave; fval: lane k < K holds the row's k-th
// position feature (0 in lanes >= K); wT = W^T [K][H] (the nn.Linear weight transposed once on the host) staged in LDS once
// per 16-row workgroup (re-reading its 15-43 KB from L2 for every row was 60-90 MB of L2 -> L1 traffic per launch).
// ONE code path for K = 5 / 7 / 14: trips of 7 unrolled steps; steps beyond K re-read row K - 1 with a zero feature.
template <int NV, bool FULL>
__device__ __forceinline__ void pos_embed_row(float fval, int K, const float* __restrict__ wT,
                                              const float* __restrict__ bias, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, float eps, int H, int lane, float4* y) {
  const int nv = H >> 2;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 g[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    y[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(bias)[c] : zero4;
    g[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(gamma)[c] : zero4;     // consumed after the reductions
    bt[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(beta)[c] : zero4;
  }
  for (int k0 = 0; k0 < K; k0 += 7) {
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int k = k0 + u;                                   // wave-uniform
      const float fk = __shfl(fval, k, 64);                   // lanes >= K hold 0
      const float4* wr = reinterpret_cast<const float4*>(wT + (size_t)min(k, K - 1) * H);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (ROW_OK(i)) {
          const float4 w = wr[lane + i * 64];
          y[i].x += fk * w.x; y[i].y += fk * w.y; y[i].z += fk * w.z; y[i].w += fk * w.w;
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (y[i].x + y[i].y) + (y[i].z + y[i].w);     // absent chunks are zero
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (ROW_OK(i)) {
      const float a = y[i].x - mean, b = y[i].y - mean, cc = y[i].z - mean, d = y[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    y[i].x = (y[i].x - mean) * rstd * g[i].x + bt[i].x;
    y[i].y = (y[i].y - mean) * rstd * g[i].y + bt[i].y;
    y[i].z = (y[i].z - mean) * rstd * g[i].z + bt[i].z;
    y[i].w = (y[i].w - mean) * rstd * g[i].w + bt[i].w;
  }
}

// grid (B, ceil(c_pad / 16)): block (b, s) writes output rows [16 s, 16 s + 16) of episode b, one wave (of 16) per row.
// Compaction + mask exactly as cells_compact_kernel (rowops.hip; vilmodel.py:813-823 with its in-place view quirk); the
// position embedding of a compacted row is computed here from the K cell-centre features of its source cell.  Only the
// blocks of slice 0 count the occupied cells of the other episodes (cmax, for the mask): one 4-byte word of occupancy
// bits per lane and episode.
template <int NV, bool FULL>
__global__ __launch_bounds__(EMB_WAVES * 64) void cells_embed_kernel(
    const float* __restrict__ proj, const float* __restrict__ pos_fts, int K, const float* __restrict__ WpT,
    const float* __restrict__ bp, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const uint8_t* __restrict__ occ, float* __restrict__ out, uint8_t* __restrict__ mask, int mask_bs,
    const uint8_t* __restrict__ tail_mask, int n_tail, int32_t* __restrict__ n_cells, int32_t* __restrict__ cmax_out,
    int B, int H, int S_pad, int c_pad) {
  __shared__ int s_src[GRIDMM_CELLS];
  __shared__ int s_n, s_tail, s_cmax;
  __shared__ int s_wmax[EMB_WAVES];
  extern __shared__ __attribute__((aligned(16))) float s_wT[];   // K * H floats; 16-B aligned: ds_read_b128 (an unaligned
                                                                 // base behind the static arrays made every read 8x slower)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_wt(WpT, K, H, s_wT);
  constexpr int WORDS = GRIDMM_CELLS / 4;      // 49 words of 4 occupancy bytes (0 / 1) per episode
  const uint32_t* ow = reinterpret_cast<const uint32_t*>(occ);
  if (blockIdx.y == 0) {
    int wmax = 0;
    for (int e0 = wave; e0 < B; e0 += 2 * EMB_WAVES) {    // 2 episodes per trip: the loads are independent
      uint32_t w[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + EMB_WAVES * u;
        w[u] = (e < B && lane < WORDS) ? ow[(size_t)e * WORDS + lane] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int n = (int)wave_sum((float)occ4(w[u]));
        wmax = n > wmax ? n : wmax;
      }
    }
    if (lane == 0) s_wmax[wave] = wmax;
  }
  if (wave == 0) {                             // ranks of this episode's occupied cells: 4 cells per lane, one load
    const uint32_t w = lane < WORDS ? ow[(size_t)b * WORDS + lane] : 0u;
    const int cnt = occ4(w);
    int pre = cnt;                             // inclusive prefix over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(pre, o, 64);
      if (lane >= o) pre += t;
    }
    int r = pre - cnt;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((w >> (8 * e)) & 0xffu) s_src[r++] = lane * 4 + e;
    if (lane == 63) s_n = pre;
  }
  __syncthreads();
  if (blockIdx.y == 0) {
    if (wave == 0) {
      const int n = s_n;
      int tail = 0;
      for (int c0 = 0; c0 < GRIDMM_CELLS; c0 += 64) {
        const int c = c0 + lane;
        const bool o = (c >= n) && (c < GRIDMM_CELLS) && occ[b * GRIDMM_CELLS + c];
        tail += __popcll(__ballot(o));
      }
      if (lane == 0) {
        int cmax = s_wmax[0];
        for (int w = 1; w < EMB_WAVES; ++w) cmax = s_wmax[w] > cmax ? s_wmax[w] : cmax;
        s_tail = n + tail;
        s_cmax = cmax;
        n_cells[b] = n;
        if (b == 0) cmax_out[0] = cmax;
      }
    }
    __syncthreads();
    const int n = s_n, lim = s_tail, cmax = s_cmax;
    for (int p = tid; p < c_pad; p += blockDim.x) {
      uint8_t m;
      if (p < n) m = 1;
      else if (p < lim) m = occ[b * GRIDMM_CELLS + p] ? 1 : 0;
      else m = 0;
      if (p >= cmax) m = 0;
      mask[(size_t)b * mask_bs + p] = m;
    }
    if (tail_mask)
      for (int j = tid; j < n_tail; j += blockDim.x) mask[(size_t)b * mask_bs + c_pad + j] = tail_mask[b * n_tail + j];
  }
  const int n = s_n, nv = H >> 2;
  const int p = blockIdx.y * EMB_WAVES + wave;
  if (p >= c_pad) return;                      // a sequence padded to c_pad < 196 cell rows (valid when cmax <= c_pad)
  float* orow = out + ((size_t)b * S_pad + p) * H;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p < n) {
    const int c = s_src[p];
    const float fval = lane < K ? pos_fts[((size_t)b * GRIDMM_CELLS + c) * K + lane] : 0.f;
    const float* prow = proj + ((size_t)b * GRIDMM_CELLS + c) * H;
    float4 a[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(prow)[lane + i * 64] : zero4;
    float4 y[NV];
    pos_embed_row<NV, FULL>(fval, K, s_wT, bp, gamma, beta, eps, H, lane, y);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ROW_OK(i))
        reinterpret_cast<float4*>(orow)[lane + i * 64] =
            make_float4(a[i].x + y[i].x, a[i].y + y[i].y, a[i].z + y[i].z, a[i].w + y[i].w);
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ROW_OK(i)) reinterpret_cast<float4*>(orow)[lane + i * 64] = zero4;
  }
}

struct NodeSegs {
  gridmm_embed_seg_t s[2];
  int n;
};

// 16 rows of ONE segment per block, one wave per row (the segment's W^T staged in LDS once); blocks [0, B) also assemble
// the byte masks of their episode
template <int NV, bool FULL>
__global__ __launch_bounds__(EMB_WAVES * 64) void node_embed_kernel(const NodeSegs segs, int blocks0, int H,
                                                                    const uint8_t* __restrict__ gmap_m, int G,
                                                                    const uint8_t* __restrict__ vp_m, int V,
                                                                    const uint8_t* __restrict__ txt_m, int L,
                                                                    uint8_t* __restrict__ kv_masks, int kv_bs,
                                                                    int kv_col0, uint8_t* __restrict__ q_masks, int B) {
  extern __shared__ __attribute__((aligned(16))) float s_wT[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < B) {
    const int b = blockIdx.x;
    if (kv_masks) {
      for (int j = threadIdx.x; j < G; j += blockDim.x) kv_masks[(size_t)b * kv_bs + kv_col0 + j] = gmap_m[b * G + j];
      for (int j = threadIdx.x; j < L; j += blockDim.x) kv_masks[(size_t)b * kv_bs + kv_col0 + G + j] = txt_m[b * L + j];
    }
    if (q_masks) {
      for (int j = threadIdx.x; j < G; j += blockDim.x) q_masks[(size_t)b * (G + V) + j] = gmap_m[b * G + j];
      for (int j = threadIdx.x; j < V; j += blockDim.x) q_masks[(size_t)b * (G + V) + G + j] = vp_m[b * V + j];
    }
  }
  const bool second = (int)blockIdx.x >= blocks0;                 // block-uniform
  if (second && segs.n < 2) return;
  const gridmm_embed_seg_t S = second ? segs.s[1] : segs.s[0];
  const int row0 = ((int)blockIdx.x - (second ? blocks0 : 0)) * EMB_WAVES, row = row0 + wave;
  if (row0 >= S.M) return;
  const int K = S.K, nv = H >> 2;
  stage_wt(S.W, K, H, s_wT);
  __syncthreads();
  if (row >= S.M) return;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float fval = lane < K ? S.pos[(size_t)row * K + lane] : 0.f;
  float4 ad[NV], tb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { ad[i] = zero4; tb[i] = zero4; }
  if (S.add1) {
    const float4* ar = reinterpret_cast<const float4*>(S.add1 + (size_t)row * S.ld1);
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ROW_OK(i)) ad[i] = ar[lane + i * 64];
  }
  if (S.table && S.idx) {
    const float4* tr = reinterpret_cast<const float4*>(S.table + (size_t)S.idx[row] * H);
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ROW_OK(i)) tb[i] = tr[lane + i * 64];
  }
  float4 y[NV];
  pos_embed_row<NV, FULL>(fval, K, s_wT, S.bias, S.gamma, S.beta, S.eps, H, lane, y);
  size_t off = (size_t)row * H;
  if (S.out_rpb > 0) { const int eb = row / S.out_rpb; off = (size_t)eb * S.out_bs + (size_t)(row - eb * S.out_rpb) * H; }
  unsigned short *ohi = (unsigned short*)S.out_hi, *olo = (unsigned short*)S.out_lo;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (ROW_OK(i)) {
      float4 v = make_float4(y[i].x + ad[i].x, y[i].y + ad[i].y, y[i].z + ad[i].z, y[i].w + ad[i].w);
      v.x += tb[i].x; v.y += tb[i].y; v.z += tb[i].z; v.w += tb[i].w;
      if (S.out) reinterpret_cast<float4*>(S.out + off)[c] = v;
      if (ohi) {
        uint2 hi, lo;
        split2_bf16(v.x, v.y, hi.x, lo.x);
        split2_bf16(v.z, v.w, hi.y, lo.y);
        reinterpret_cast<uint2*>(ohi + off)[c] = hi;
        reinterpret_cast<uint2*>(olo + off)[c] = lo;
      }
    }
  }
}

struct HeadTails { gridmm_cls_tail_t t[5]; };

// <LN(x) * gamma + beta, w> + b0 of one row by one wave (x: row of H floats; x2 / xb: optional second addend + bias, then ReLU)
template <int NV, bool FULL>
__device__ __forceinline__ float ln_dot_row(const float* __restrict__ x, const float* __restrict__ x2,
                                            const float* __restrict__ xb, const gridmm_cls_tail_t& T, int H, int lane) {
  const int nv = H >> 2;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[NV], g[NV], bt[NV], ww[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    v[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(x)[c] : zero4;
    g[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(T.gamma)[c] : zero4;
    bt[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(T.beta)[c] : zero4;
    ww[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(T.w)[c] : zero4;
  }
  if (x2) {   // fuse head: the two K-halves of its Linear arrive as separate partial products
    float4 p[NV], q[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      p[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(x2)[c] : zero4;
      q[i] = ROW_OK(i) ? reinterpret_cast<const float4*>(xb)[c] : zero4;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x = fmaxf(v[i].x + p[i].x + q[i].x, 0.f); v[i].y = fmaxf(v[i].y + p[i].y + q[i].y, 0.f);
      v[i].z = fmaxf(v[i].z + p[i].z + q[i].z, 0.f); v[i].w = fmaxf(v[i].w + p[i].w + q[i].w, 0.f);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (ROW_OK(i)) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + T.eps);
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (ROW_OK(i)) {
      d += ((v[i].x - mean) * rstd * g[i].x + bt[i].x) * ww[i].x + ((v[i].y - mean) * rstd * g[i].y + bt[i].y) * ww[i].y +
           ((v[i].z - mean) * rstd * g[i].z + bt[i].z) * ww[i].z + ((v[i].w - mean) * rstd * g[i].w + bt[i].w) * ww[i].w;
    }
  }
  return wave_sum(d) + (T.b0 ? T.b0[0] : 0.f);
}

// Launch 1: one wave per head row over the whole batch -- rows of episode b: 1 (fuse) + G (global) + V (local) + G (grid)
// [+ V (object)] -> raw[b][.] (workspace).  Launch 2: the masking / fusion of fuse_logits_kernel (rowops.hip) per episode.
template <int NV, bool FULL>
__global__ __launch_bounds__(256) void nav_head_rows_kernel(
    const float* __restrict__ h_gl, int ld_gl, const float* __restrict__ fuse_a, const float* __restrict__ fuse_b,
    const float* __restrict__ fuse_bias, const float* __restrict__ h_grid, const HeadTails tails, int has_obj,
    float* __restrict__ raw, int B, int G, int V, int H) {
  const int lane = threadIdx.x & 63;
  const int rows = 1 + G + V + G + (has_obj ? V : 0);
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= B * rows) return;
  const int b = id / rows, r = id - b * rows, Sq = G + V;
  // one ln_dot_row call site (the row pointer and the tail record are selected first): the code is instantiated once
  const float *x, *x2 = nullptr, *xb = nullptr;
  int ti;
  if (r == 0) { x = fuse_a + (size_t)b * H; x2 = fuse_b + (size_t)b * H; xb = fuse_bias; ti = 0; }
  else if (r < 1 + G) { x = h_gl + ((size_t)b * Sq + (r - 1)) * ld_gl; ti = 1; }
  else if (r < 1 + G + V) { x = h_gl + ((size_t)b * Sq + G + (r - 1 - G)) * ld_gl + H; ti = 2; }
  else if (r < 1 + G + V + G) { x = h_grid + ((size_t)b * G + (r - 1 - G - V)) * H; ti = 3; }
  else { x = h_gl + ((size_t)b * Sq + G + (r - 1 - G - V - G)) * ld_gl + 2 * H; ti = 4; }
  float val = 0.f;
  if (r != 0 || fuse_a) {
    gridmm_cls_tail_t T = tails.t[0];
    if (ti == 1) T = tails.t[1]; else if (ti == 2) T = tails.t[2]; else if (ti == 3) T = tails.t[3]; else if (ti == 4) T = tails.t[4];
    val = ln_dot_row<NV, FULL>(x, x2, xb, T, H, lane);
  }
  if (lane == 0) raw[(size_t)b * rows + r] = val;
}

__global__ __launch_bounds__(64) void nav_fuse_kernel(
    const float* __restrict__ raw, int has_fuse, int has_obj, const uint8_t* __restrict__ gmap_masks,
    const uint8_t* __restrict__ gmap_visited, const uint8_t* __restrict__ vp_nav_masks,
    const uint8_t* __restrict__ vp_obj_masks, const int32_t* __restrict__ cand_of_node,
    const uint8_t* __restrict__ cand_visited, float* __restrict__ global_logits, float* __restrict__ local_logits,
    float* __restrict__ grid_logits, float* __restrict__ fused_logits, float* __restrict__ obj_logits, int G, int V) {
  extern __shared__ float s_l[];   // V masked local logits
  const int b = blockIdx.x;
  const int rows = 1 + G + V + G + (has_obj ? V : 0);
  const float* rb = raw + (size_t)b * rows;
  const float *r_g = rb + 1, *r_l = r_g + G, *r_gr = r_l + V, *r_o = r_gr + G;
  const float ninf = -__builtin_inff();
  const float fw = has_fuse ? 1.0f / (1.0f + expf(-rb[0])) : 0.5f;
  for (int k = threadIdx.x; k < V; k += blockDim.x) {
    float v = r_l[k] * (1.0f - fw);
    if (!vp_nav_masks[b * V + k]) v = ninf;
    s_l[k] = v;
    local_logits[b * V + k] = v;
    if (has_obj) {
      float o = r_o[k];
      if (!vp_obj_masks[b * V + k]) o = ninf;
      obj_logits[b * V + k] = o;
    }
  }
  __syncthreads();
  float bw = 0.f;   // sum of the local logits of visited candidates, in candidate order (python `bw_logits += ...`)
  for (int k = 1; k < V; ++k)
    if (cand_visited[b * V + k]) bw += s_l[k];
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    const bool vis = gmap_visited[b * G + j], valid = gmap_masks[b * G + j];
    float g = r_g[j] * fw;
    if (vis || !valid) g = ninf;
    float gr = r_gr[j];
    if (vis || !valid) gr = ninf;
    global_logits[b * G + j] = g;
    grid_logits[b * G + j] = gr;
    float f = g;
    if (j == 0) f += s_l[0];
    else {
      const int k = cand_of_node[b * G + j];
      if (k >= 0) f += s_l[k];
      else if (k == -1) f += bw;
    }
    fused_logits[b * G + j] = f;
  }
}

#undef ROW_OK

// instantiation switch over (NV, FULL) for a row kernel launch
#define GRIDMM_ROW_DISPATCH(H, LAUNCH)                                                         \
  do {                                                                                         \
    const int nvq_ = ((H) / 4 + 63) / 64;                                                      \
    if ((H) % 256 == 0) {                                                                      \
      if (nvq_ == 1) { LAUNCH(1, true); } else if (nvq_ == 2) { LAUNCH(2, true); }             \
      else if (nvq_ == 3) { LAUNCH(3, true); } else { LAUNCH(4, true); }                       \
    } else {                                                                                   \
      if (nvq_ == 1) { LAUNCH(1, false); } else if (nvq_ == 2) { LAUNCH(2, false); }           \
      else if (nvq_ == 3) { LAUNCH(3, false); } else { LAUNCH(4, false); }                     \
    }                                                                                          \
  } while (0)

}  // namespace

extern "C" int gridmm_cells_embed(const float* proj, const float* pos_fts, int K, const float* W_pos, const float* b_pos,
                                  const float* gamma, const float* beta, float eps, const uint8_t* occ, float* out,
                                  uint8_t* mask, int mask_bs, const uint8_t* tail_mask, int n_tail, int32_t* n_cells,
                                  int32_t* cmax, int B, int H, int S_pad, int c_pad, gridmm_stream_t stream) {
  if (c_pad <= 0) c_pad = GRIDMM_CELLS;
  if (B <= 0 || H <= 0 || H % 4 || H > 1024 || K <= 0 || K > MAXK || c_pad > GRIDMM_CELLS ||
      S_pad < c_pad + (tail_mask ? n_tail : 0) || mask_bs < c_pad + (tail_mask ? n_tail : 0))
    return GRIDMM_EINVAL;
#define GRIDMM_CE(NVQ, F)                                                                                              \
  GRIDMM_LAUNCH((cells_embed_kernel<NVQ, F>), dim3(B, (c_pad + EMB_WAVES - 1) / EMB_WAVES), dim3(EMB_WAVES * 64),      \
                (size_t)K * H * sizeof(float), as_stream(stream), proj, pos_fts, K, W_pos, b_pos, gamma, beta, eps, occ, \
                out, mask, mask_bs, tail_mask, n_tail, n_cells, cmax, B, H, S_pad, c_pad)
  GRIDMM_ROW_DISPATCH(H, GRIDMM_CE);
#undef GRIDMM_CE
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_node_embed(const gridmm_embed_seg_t* segs, int n_segs, int H, const uint8_t* gmap_masks, int G,
                                 const uint8_t* vp_masks, int V, const uint8_t* txt_masks, int L, uint8_t* kv_masks,
                                 int kv_bs, int kv_col0, uint8_t* q_masks, int B, gridmm_stream_t stream) {
  if (!segs || n_segs < 1 || n_segs > 2 || H <= 0 || H % 4 || H > 1024 || B < 0) return GRIDMM_EINVAL;
  NodeSegs a;
  a.n = n_segs;
  int rows = 0;
  for (int i = 0; i < n_segs; ++i) {
    const gridmm_embed_seg_t& s = segs[i];
    if (s.M <= 0 || s.K <= 0 || s.K > MAXK || !s.pos || !s.W || !s.bias || !s.gamma || !s.beta || (!s.out && !s.out_hi) ||
        (s.out_hi && !s.out_lo) || (s.add1 && s.ld1 % 4) || (s.out_rpb > 0 && s.out_bs % 4))
      return GRIDMM_EINVAL;
    a.s[i] = s;
    rows += s.M;
  }
  if ((kv_masks || q_masks) && (!gmap_masks || G <= 0)) return GRIDMM_EINVAL;
  if (kv_masks && (!txt_masks || kv_bs < kv_col0 + G + L)) return GRIDMM_EINVAL;
  if (q_masks && (!vp_masks || V <= 0)) return GRIDMM_EINVAL;
  const int blocks0 = (a.s[0].M + EMB_WAVES - 1) / EMB_WAVES;
  int blocks = blocks0 + (n_segs > 1 ? (a.s[1].M + EMB_WAVES - 1) / EMB_WAVES : 0), maxk = a.s[0].K;
  if (n_segs > 1 && a.s[1].K > maxk) maxk = a.s[1].K;
  if ((kv_masks || q_masks) && blocks < B) blocks = B;
  (void)rows;
#define GRIDMM_NE(NVQ, F)                                                                                         \
  GRIDMM_LAUNCH((node_embed_kernel<NVQ, F>), dim3(blocks), dim3(EMB_WAVES * 64), (size_t)maxk * H * sizeof(float),  \
                as_stream(stream), a, blocks0, H, gmap_masks, G, vp_masks, V, txt_masks, L, kv_masks, kv_bs, kv_col0, \
                q_masks, (kv_masks || q_masks) ? B : 0)
  GRIDMM_ROW_DISPATCH(H, GRIDMM_NE);
#undef GRIDMM_NE
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" size_t gridmm_nav_heads_workspace(int B, int G, int V) {
  return (size_t)B * (1 + 2 * G + 2 * V) * sizeof(float);
}

extern "C" int gridmm_nav_heads(const float* h_gl, int ld_gl, const float* fuse_a, const float* fuse_b,
                                const float* fuse_bias, const float* h_grid, const gridmm_cls_tail_t* tails,
                                const uint8_t* gmap_masks, const uint8_t* gmap_visited, const uint8_t* vp_nav_masks,
                                const uint8_t* vp_obj_masks, const int32_t* cand_of_node, const uint8_t* cand_visited,
                                float* global_logits, float* local_logits, float* grid_logits, float* fused_logits,
                                float* obj_logits, void* workspace, int B, int G, int V, int H, gridmm_stream_t stream) {
  if (!h_gl || !h_grid || !tails || !workspace || B <= 0 || G <= 0 || V <= 0 || V > 4096 || H <= 0 || H % 4 || H > 1024 ||
      ld_gl % 4)
    return GRIDMM_EINVAL;
  if ((fuse_a && (!fuse_b || !fuse_bias)) || (obj_logits && !vp_obj_masks)) return GRIDMM_EINVAL;
  const int has_obj = obj_logits ? 1 : 0;
  if (ld_gl < (2 + has_obj) * H) return GRIDMM_EINVAL;
  HeadTails t;
  for (int i = 0; i < 5; ++i) t.t[i] = tails[i < 4 || has_obj ? i : 3];
  const int rows = 1 + G + V + G + (has_obj ? V : 0);
  float* raw = (float*)workspace;
#define GRIDMM_HR(NVQ, F)                                                                                              \
  GRIDMM_LAUNCH((nav_head_rows_kernel<NVQ, F>), dim3((B * rows + 3) / 4), dim3(256), 0, as_stream(stream), h_gl, ld_gl, \
                fuse_a, fuse_b, fuse_bias, h_grid, t, has_obj, raw, B, G, V, H)
  GRIDMM_ROW_DISPATCH(H, GRIDMM_HR);
#undef GRIDMM_HR
  GRIDMM_CHECK_LAUNCH();
  GRIDMM_LAUNCH(nav_fuse_kernel, dim3(B), dim3(64), (size_t)V * sizeof(float), as_stream(stream), raw, fuse_a ? 1 : 0,
                has_obj, gmap_masks, gmap_visited, vp_nav_masks, vp_obj_masks, cand_of_node, cand_visited, global_logits,
                local_logits, grid_logits, fused_logits, obj_logits, G, V);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
