// Row-wise fp32 kernels around the GEMMs: LayerNorm (+fused residual / embedding adds),
// LayerNorm+dot heads, sequence assembly copies, cell compaction and logit fusion.
// All HBM-bound, one 64-lane wave per row, 128-bit loads.
#include "common.h"

namespace {

constexpr int MAX_H = 1024;  // hidden sizes on this path: 768

// One wave per row; H % 4 == 0; each lane holds H/256 float4 (3 for H = 768).
// FULL (H == 256 * NV: every lane owns exactly NV chunks): no per-chunk guards, so the loads of a phase are issued
// together -- with guards the compiler emits one branch + one s_waitcnt per chunk and operand, i.e. ~10 serialised memory
// round trips per row, which is what a 1824-row launch (456 workgroups, one round) costs; gamma / beta / add1 / table rows
// are fetched BEFORE the two wave reductions so that they fly under them.
// DROP (training, gridmm_layernorm_dropout): X is the output of a dense layer whose hidden-state dropout is applied HERE,
// on load -- keep(seed, row * H + col) ? x / (1 - p) : 0, the mask of gridmm_dropout on the contiguous (M, H) tensor --
// so LN(dropout(X) + R) costs no launch and no pass of its own (BertSelfOutput / BertOutput, vilmodel.py:160-170, 199-211).
template <int NV, bool FULL, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_kernel(
    const float* __restrict__ X, int ldx, const float* __restrict__ R, int ldr,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float* __restrict__ Y, int ldy, const float* __restrict__ add1, int ld1,
    const float* __restrict__ table, const int64_t* __restrict__ idx, unsigned short* __restrict__ Yhi,
    unsigned short* __restrict__ Ylo, int ldp, int p_rpb, long p_bs, int M, int H, float drop_p = 0.f,
    unsigned long long seed = 0, const unsigned long long* __restrict__ seed_dev = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  size_t poff = (size_t)row * ldp;   // planes through the batched row map (gridmm_layernorm_map)
  if (p_rpb > 0) { const int eb = row / p_rpb; poff = (size_t)eb * p_bs + (size_t)(row - eb * p_rpb) * ldp; }
  const int nv = H >> 2;  // float4 per row
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto ok = [&](int i) { return FULL || lane + i * 64 < nv; };
  float4 v[NV], g[NV], bt[NV], ex[NV], tb[NV];
  const float4* xr = reinterpret_cast<const float4*>(X + (size_t)row * ldx);
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = ok(i) ? xr[lane + i * 64] : zero4;
  if constexpr (DROP) {
    if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
    const float scale = 1.0f / (1.0f - drop_p);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const unsigned int e = (unsigned int)row * (unsigned int)H + 4u * (unsigned int)(lane + i * 64);
      v[i].x = dropout_keep(seed, e, drop_p) ? __fmul_rn(v[i].x, scale) : 0.f;
      v[i].y = dropout_keep(seed, e + 1, drop_p) ? __fmul_rn(v[i].y, scale) : 0.f;
      v[i].z = dropout_keep(seed, e + 2, drop_p) ? __fmul_rn(v[i].z, scale) : 0.f;
      v[i].w = dropout_keep(seed, e + 3, drop_p) ? __fmul_rn(v[i].w, scale) : 0.f;
    }
  }
  if (R) {
    const float4* rr = reinterpret_cast<const float4*>(R + (size_t)row * ldr);
    float4 r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = ok(i) ? rr[lane + i * 64] : zero4;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i].x += r[i].x; v[i].y += r[i].y; v[i].z += r[i].z; v[i].w += r[i].w; }
  }
  // operands of the output phase: issued now, consumed after the reductions
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g[i] = ok(i) ? reinterpret_cast<const float4*>(gamma)[lane + i * 64] : zero4;
    bt[i] = ok(i) ? reinterpret_cast<const float4*>(beta)[lane + i * 64] : zero4;
    ex[i] = zero4;
    tb[i] = zero4;
  }
  if (add1) {
    const float4* ar = reinterpret_cast<const float4*>(add1 + (size_t)row * ld1);
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ok(i)) ex[i] = ar[lane + i * 64];
  }
  if (table && idx) {
    const float4* tr = reinterpret_cast<const float4*>(table + (size_t)idx[row] * H);
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ok(i)) tb[i] = tr[lane + i * 64];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);     // absent chunks are zero
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (ok(i)) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (ok(i)) {
      float4 y;
      y.x = (v[i].x - mean) * rstd * g[i].x + bt[i].x;
      y.y = (v[i].y - mean) * rstd * g[i].y + bt[i].y;
      y.z = (v[i].z - mean) * rstd * g[i].z + bt[i].z;
      y.w = (v[i].w - mean) * rstd * g[i].w + bt[i].w;
      if (add1) { y.x += ex[i].x; y.y += ex[i].y; y.z += ex[i].z; y.w += ex[i].w; }
      if (table && idx) { y.x += tb[i].x; y.y += tb[i].y; y.z += tb[i].z; y.w += tb[i].w; }
      if (Y) reinterpret_cast<float4*>(Y + (size_t)row * ldy)[c] = y;
      if (Yhi) {
        const float x[4] = {y.x, y.y, y.z, y.w};
        u16x4_t hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned short h = f32_to_bf16_rne(x[e]);
          hi[e] = h;
          lo[e] = f32_to_bf16_rne(x[e] - bf16_bits_to_f32(h));
        }
        reinterpret_cast<u16x4_t*>(Yhi + poff)[c] = hi;
        reinterpret_cast<u16x4_t*>(Ylo + poff)[c] = lo;
      }
    }
  }
}

template <int NV, bool FULL>
__global__ __launch_bounds__(256) void ln_dot_kernel(
    const float* __restrict__ X, int ldx, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, const float* __restrict__ w,
    const float* __restrict__ b0, float* __restrict__ out, int M, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = H >> 2;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto ok = [&](int i) { return FULL || lane + i * 64 < nv; };
  float4 v[NV], g[NV], bt[NV], ww[NV];
  const float4* xr = reinterpret_cast<const float4*>(X + (size_t)row * ldx);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = ok(i) ? xr[lane + i * 64] : zero4;
    g[i] = ok(i) ? reinterpret_cast<const float4*>(gamma)[lane + i * 64] : zero4;
    bt[i] = ok(i) ? reinterpret_cast<const float4*>(beta)[lane + i * 64] : zero4;
    ww[i] = ok(i) ? reinterpret_cast<const float4*>(w)[lane + i * 64] : zero4;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (ok(i)) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (ok(i)) {
      d += ((v[i].x - mean) * rstd * g[i].x + bt[i].x) * ww[i].x + ((v[i].y - mean) * rstd * g[i].y + bt[i].y) * ww[i].y +
           ((v[i].z - mean) * rstd * g[i].z + bt[i].z) * ww[i].z + ((v[i].w - mean) * rstd * g[i].w + bt[i].w) * ww[i].w;
    }
  }
  d = wave_sum(d);
  if (lane == 0) out[row] = d + (b0 ? b0[0] : 0.f);
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, int64_t src_bs,
                                                        int src_rs, float* __restrict__ dst,
                                                        int64_t dst_bs, int dst_rs, int rows, int H) {
  const int b = blockIdx.y;
  const int nv = H >> 2;
  const size_t total = (size_t)rows * nv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / nv), c = (int)(i % nv);
    reinterpret_cast<float4*>(dst + b * dst_bs + (size_t)r * dst_rs)[c] =
        reinterpret_cast<const float4*>(src + b * src_bs + (size_t)r * src_rs)[c];
  }
}

// One workgroup per episode.  vilmodel.py:813-823:
//   embeds[b, :n_b] = (cells + pos)[b][occ == 1]   (cell order)
//   mask: first n_b set to 1; then -- because `grid_mask` is a VIEW of the row being
//   written -- everything from (n_b + #occupied positions >= n_b) on is cleared and the
//   positions in between keep their ORIGINAL occupancy bit.  Finally [:, :Cmax].
__global__ __launch_bounds__(256) void cells_compact_kernel(
    const float* __restrict__ proj, const float* __restrict__ pos_emb,
    const uint8_t* __restrict__ occ, float* __restrict__ out, uint8_t* __restrict__ mask,
    int32_t* __restrict__ n_cells, int32_t* __restrict__ cmax_out, int B, int H, int S_pad) {
  __shared__ int s_rank[GRIDMM_CELLS];   // rank of each occupied cell, -1 if empty
  __shared__ int s_src[GRIDMM_CELLS];    // source cell of each compacted slot
  __shared__ int s_n, s_tail, s_cmax;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int s_wmax[4];
  // Cmax = max_e #occupied(e): each wave counts whole episodes with ballots
  int wmax = 0;
  for (int e = wave; e < B; e += 4) {
    int n = 0;
    for (int c0 = 0; c0 < GRIDMM_CELLS; c0 += 64) {
      const int c = c0 + lane;
      const bool o = (c < GRIDMM_CELLS) && occ[e * GRIDMM_CELLS + c];
      n += __popcll(__ballot(o));
    }
    wmax = n > wmax ? n : wmax;
  }
  if (lane == 0) s_wmax[wave] = wmax;
  // ranks of this episode's occupied cells (wave 0, ballot prefix)
  if (wave == 0) {
    int base = 0;
    for (int c0 = 0; c0 < GRIDMM_CELLS; c0 += 64) {
      const int c = c0 + lane;
      const bool o = (c < GRIDMM_CELLS) && occ[b * GRIDMM_CELLS + c];
      const unsigned long long m = __ballot(o);
      const int r = base + __popcll(m & ((1ull << lane) - 1ull));
      if (c < GRIDMM_CELLS) s_rank[c] = o ? r : -1;
      if (o) s_src[r] = c;
      base += __popcll(m);
    }
    if (lane == 0) s_n = base;
  }
  __syncthreads();
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
      for (int w = 1; w < 4; ++w) cmax = s_wmax[w] > cmax ? s_wmax[w] : cmax;
      s_tail = n + tail;
      s_cmax = cmax;
      if (blockIdx.y == 0) n_cells[b] = n;
      if (b == 0 && blockIdx.y == 0) cmax_out[0] = cmax;
    }
  }
  __syncthreads();
  const int n = s_n, lim = s_tail, cmax = s_cmax;
  for (int p = tid; p < GRIDMM_CELLS && blockIdx.y == 0; p += blockDim.x) {
    uint8_t m;
    if (p < n) m = 1;
    else if (p < lim) m = occ[b * GRIDMM_CELLS + p] ? 1 : 0;
    else m = 0;
    if (p >= cmax) m = 0;
    mask[(size_t)b * S_pad + p] = m;
  }
  const int nv = H >> 2;
  float* ob = out + (size_t)b * S_pad * H;
  // blockIdx.y splits the 196 output rows into 14 slices (the row copy is the only real work)
  const int p_lo = blockIdx.y * GRIDMM_GRID, p_hi = p_lo + GRIDMM_GRID;
  for (int i = p_lo * nv + tid; i < p_hi * nv; i += blockDim.x) {
    const int p = i / nv, c = i % nv;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < n) {
      const size_t src = ((size_t)b * GRIDMM_CELLS + s_src[p]) * H;
      const float4 a = reinterpret_cast<const float4*>(proj + src)[c];
      const float4 q = reinterpret_cast<const float4*>(pos_emb + src)[c];
      v = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
    }
    reinterpret_cast<float4*>(ob + (size_t)p * H)[c] = v;
  }
}

// One thread per (b, j); vilmodel.py:859-907.
__global__ void fuse_logits_kernel(const float* __restrict__ g_raw, const float* __restrict__ l_raw,
                                   const float* __restrict__ grid_raw, const float* __restrict__ fuse_raw,
                                   const uint8_t* __restrict__ gmap_masks,
                                   const uint8_t* __restrict__ gmap_visited,
                                   const uint8_t* __restrict__ vp_nav_masks,
                                   const int32_t* __restrict__ cand_of_node,
                                   const uint8_t* __restrict__ cand_visited,
                                   float* __restrict__ global_logits, float* __restrict__ local_logits,
                                   float* __restrict__ grid_logits, float* __restrict__ fused_logits,
                                   int B, int G, int V) {
  const int b = blockIdx.x;
  const float ninf = -__builtin_inff();
  const float fw = fuse_raw ? 1.0f / (1.0f + expf(-fuse_raw[b])) : 0.5f;
  extern __shared__ float s_local[];  // V masked local logits
  for (int k = threadIdx.x; k < V; k += blockDim.x) {
    float v = l_raw[b * V + k] * (1.0f - fw);
    if (!vp_nav_masks[b * V + k]) v = ninf;
    s_local[k] = v;
    local_logits[b * V + k] = v;
  }
  __syncthreads();
  // sum of the local logits of visited candidates, in candidate order (python `bw_logits += ...`)
  float bw = 0.f;
  for (int k = 1; k < V; ++k)
    if (cand_visited[b * V + k]) bw += s_local[k];
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    const bool vis = gmap_visited[b * G + j], valid = gmap_masks[b * G + j];
    float g = g_raw[b * G + j] * fw;
    if (vis || !valid) g = ninf;
    float gr = grid_raw[b * G + j];
    if (vis || !valid) gr = ninf;
    global_logits[b * G + j] = g;
    grid_logits[b * G + j] = gr;
    float f = g;
    if (j == 0) f += s_local[0];
    else {
      const int k = cand_of_node[b * G + j];
      if (k >= 0) f += s_local[k];
      else if (k == -1) f += bw;
      /* k == -2: visited or padded node, nothing added */
    }
    fused_logits[b * G + j] = f;
  }
}

}  // namespace

extern "C" int gridmm_layernorm_map(const float* X, int ldx, const float* R, int ldr, const float* gamma,
                                    const float* beta, float eps, float* Y, int ldy, const float* add1,
                                    int ld1, const float* table, const int64_t* idx, void* Y_hi, void* Y_lo,
                                    int ldp, int p_rpb, int64_t p_bs, int M, int H, gridmm_stream_t stream) {
  if (p_rpb > 0 && (p_bs % 4)) return GRIDMM_EINVAL;
  if (M <= 0 || H <= 0 || H % 4 || H > MAX_H || ldx % 4 || (Y && ldy % 4) || (R && ldr % 4) || (add1 && ld1 % 4))
    return GRIDMM_EINVAL;
  if ((!Y && !Y_hi) || (Y_hi && (!Y_lo || ldp % 4))) return GRIDMM_EINVAL;
  unsigned short *Yhi = (unsigned short*)Y_hi, *Ylo = (unsigned short*)Y_lo;
  dim3 grid((M + 3) / 4), block(256);
  const int nv = (H / 4 + 63) / 64;
  const bool full = (H % 256) == 0;
#define GRIDMM_LN(NV, F)                                                                              \
  GRIDMM_LAUNCH((layernorm_kernel<NV, F>), grid, block, 0, as_stream(stream), X, ldx, R, ldr, gamma, \
                     beta, eps, Y, ldy, add1, ld1, table, idx, Yhi, Ylo, ldp, p_rpb, (long)p_bs, M, H)
  if (full) { if (nv == 1) GRIDMM_LN(1, true); else if (nv == 2) GRIDMM_LN(2, true); else if (nv == 3) GRIDMM_LN(3, true); else GRIDMM_LN(4, true); }
  else { if (nv == 1) GRIDMM_LN(1, false); else if (nv == 2) GRIDMM_LN(2, false); else if (nv == 3) GRIDMM_LN(3, false); else GRIDMM_LN(4, false); }
#undef GRIDMM_LN
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_layernorm(const float* X, int ldx, const float* R, int ldr, const float* gamma,
                                const float* beta, float eps, float* Y, int ldy, const float* add1,
                                int ld1, const float* table, const int64_t* idx, void* Y_hi, void* Y_lo,
                                int ldp, int M, int H, gridmm_stream_t stream) {
  return gridmm_layernorm_map(X, ldx, R, ldr, gamma, beta, eps, Y, ldy, add1, ld1, table, idx, Y_hi, Y_lo, ldp, 0, 0, M, H,
                              stream);
}

// _planes: also the bf16 hi/lo planes (M, H) of Y -- the next Linear's A operand and, in the backward, an operand of its
// weight gradient (gridmm_linear_planes_tn): no split pass over Y.
extern "C" int gridmm_layernorm_dropout_planes(const float* X, const float* R, int ldr, const float* gamma, const float* beta,
                                               float eps, float* Y, void* Y_hi, void* Y_lo, float p, unsigned long long seed,
                                               const unsigned long long* seed_dev, int M, int H, gridmm_stream_t stream) {
  if (M <= 0 || H <= 0 || H % 4 || H > MAX_H || !X || !Y || !(p >= 0.f && p < 1.f) || (R && ldr % 4) ||
      (size_t)M * H >= (1ull << 32) || (Y_hi && !Y_lo))
    return GRIDMM_EINVAL;
  dim3 grid((M + 3) / 4), block(256);
  const int nv = (H / 4 + 63) / 64;
  const bool full = (H % 256) == 0;
#define GRIDMM_LND(NV, F)                                                                                             \
  GRIDMM_LAUNCH((layernorm_kernel<NV, F, true>), grid, block, 0, as_stream(stream), X, H, R, ldr, gamma, beta, eps, Y, H, \
                (const float*)nullptr, 0, (const float*)nullptr, (const int64_t*)nullptr, (unsigned short*)Y_hi,     \
                (unsigned short*)Y_lo, H, 0, 0L, M, H, p, seed, seed_dev)
  if (full) { if (nv == 1) GRIDMM_LND(1, true); else if (nv == 2) GRIDMM_LND(2, true); else if (nv == 3) GRIDMM_LND(3, true); else GRIDMM_LND(4, true); }
  else { if (nv == 1) GRIDMM_LND(1, false); else if (nv == 2) GRIDMM_LND(2, false); else if (nv == 3) GRIDMM_LND(3, false); else GRIDMM_LND(4, false); }
#undef GRIDMM_LND
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_layernorm_dropout(const float* X, const float* R, int ldr, const float* gamma, const float* beta,
                                        float eps, float* Y, float p, unsigned long long seed,
                                        const unsigned long long* seed_dev, int M, int H, gridmm_stream_t stream) {
  return gridmm_layernorm_dropout_planes(X, R, ldr, gamma, beta, eps, Y, nullptr, nullptr, p, seed, seed_dev, M, H, stream);
}

extern "C" int gridmm_ln_dot(const float* X, int ldx, const float* gamma, const float* beta, float eps,
                             const float* w, const float* b0, float* out, int M, int H,
                             gridmm_stream_t stream) {
  if (M <= 0 || H <= 0 || H % 4 || H > MAX_H || ldx % 4) return GRIDMM_EINVAL;
  dim3 grid((M + 3) / 4), block(256);
  const int nv = (H / 4 + 63) / 64;
  const bool full = (H % 256) == 0;
#define GRIDMM_LD(NV, F)                                                                             \
  GRIDMM_LAUNCH((ln_dot_kernel<NV, F>), grid, block, 0, as_stream(stream), X, ldx, gamma, beta, eps, \
                     w, b0, out, M, H)
  if (full) { if (nv == 1) GRIDMM_LD(1, true); else if (nv == 2) GRIDMM_LD(2, true); else if (nv == 3) GRIDMM_LD(3, true); else GRIDMM_LD(4, true); }
  else { if (nv == 1) GRIDMM_LD(1, false); else if (nv == 2) GRIDMM_LD(2, false); else if (nv == 3) GRIDMM_LD(3, false); else GRIDMM_LD(4, false); }
#undef GRIDMM_LD
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_copy_rows(const float* src, int64_t src_bs, int src_rs, float* dst, int64_t dst_bs,
                                int dst_rs, int B, int rows, int H, gridmm_stream_t stream) {
  if (B <= 0 || rows <= 0 || H <= 0 || H % 4 || src_rs % 4 || dst_rs % 4 || src_bs % 4 || dst_bs % 4)
    return GRIDMM_EINVAL;
  const size_t total = (size_t)rows * (H / 4);
  unsigned gx = (unsigned)((total + 255) / 256);
  if (gx > 1024) gx = 1024;
  GRIDMM_LAUNCH(copy_rows_kernel, dim3(gx, B), dim3(256), 0, as_stream(stream), src, src_bs, src_rs,
                     dst, dst_bs, dst_rs, rows, H);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_cells_compact(const float* proj, const float* pos_emb, const uint8_t* occ, float* out,
                                    uint8_t* mask, int32_t* n_cells, int32_t* cmax, int B, int H, int S_pad,
                                    gridmm_stream_t stream) {
  if (B <= 0 || H <= 0 || H % 4 || S_pad < GRIDMM_CELLS) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(cells_compact_kernel, dim3(B, GRIDMM_GRID), dim3(256), 0, as_stream(stream), proj, pos_emb, occ,
                     out, mask, n_cells, cmax, B, H, S_pad);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_fuse_logits(const float* g_raw, const float* l_raw, const float* grid_raw,
                                  const float* fuse_raw, const uint8_t* gmap_masks,
                                  const uint8_t* gmap_visited, const uint8_t* vp_nav_masks,
                                  const int32_t* cand_of_node, const uint8_t* cand_visited,
                                  float* global_logits, float* local_logits, float* grid_logits,
                                  float* fused_logits, int B, int G, int V, gridmm_stream_t stream) {
  if (B <= 0 || G <= 0 || V <= 0 || V > 4096) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(fuse_logits_kernel, dim3(B), dim3(64), V * sizeof(float), as_stream(stream), g_raw,
                     l_raw, grid_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, cand_of_node,
                     cand_visited, global_logits, local_logits, grid_logits, fused_logits, B, G, V);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

// ---- patch tokens of an image encoder -> the grid memory's fp16 slab -------------------------------------------------
// X: (B * n_views, T, D) fp32 token rows of the vision tower (token 0 = class token, dropped).  Episode b's slot is
// slab + b * slab_bs: n_views * (T-1) rows of D fp16, view-major -- exactly the order getGlobalMap appends them in
// (VLN_CE/vlnce_baselines/models/Policy_ViewSelection_GridMap.py:343-357: batch_grid_fts.view(B,12,50,768), [1:] per view).
namespace {
__global__ __launch_bounds__(256) void tokens_to_slab_kernel(const float* __restrict__ X, int T, int D,
                                                             _Float16* __restrict__ slab, int64_t slab_bs, int n_views) {
  const int b = blockIdx.y;
  const int rows = n_views * (T - 1), d4 = D / 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * d4; i += gridDim.x * blockDim.x) {
    const int r = i / d4, c = (i - r * d4) * 4;
    const int v = r / (T - 1), t = r - v * (T - 1) + 1;
    const float4 x = *reinterpret_cast<const float4*>(X + ((size_t)(b * n_views + v) * T + t) * D + c);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 h = {(_Float16)x.x, (_Float16)x.y, (_Float16)x.z, (_Float16)x.w};
    *reinterpret_cast<h4*>(slab + b * slab_bs + (size_t)r * D + c) = h;
  }
}
}  // namespace

extern "C" int gridmm_tokens_to_slab(const float* X, int T, int D, void* slab, int64_t slab_bs, int B, int n_views,
                                     gridmm_stream_t stream) {
  if (!X || !slab || B <= 0 || n_views <= 0 || T < 2 || D <= 0 || D % 4 || slab_bs % 4) return GRIDMM_EINVAL;
  const int work = n_views * (T - 1) * (D / 4);
  GRIDMM_LAUNCH(tokens_to_slab_kernel, dim3((work + 255) / 256 > 256 ? 256 : (work + 255) / 256, B), dim3(256), 0,
                as_stream(stream), X, T, D, (_Float16*)slab, slab_bs, n_views);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
