// Backward-pass kernels for training (fine-tune DAgger loop / pre-training).
//
// GEMM backward reuses the forward tile kernel (gridmm_linear_planes: C = A B^T on MFMA bf16x3):
//   dX = dY W      -> A = dY planes [M][N],        "weights" = W^T planes [K][N]        (packed once per weight)
//   dW = dY^T X    -> A = dY^T planes [N][Mp],     "weights" = X^T planes [K][Mp]       (contraction over rows)
// so the only new GEMM-side kernel is the transpose+split below (which also yields db = column sums of dY).
// LayerNorm / GELU / softmax-attention backward are row- or tile-local fp32 kernels.
#include "common.h"

namespace {

// X fp32 [M][C] (row stride ldx) -> bf16 hi/lo planes [C][Mp] (Mp % 32 == 0, zero padded) and colsum[C]
// 64 x 64 tiles through LDS; grid (ceil(C/64), ceil(Mp/64)).
constexpr int TS_TILES = 4;   // 64-row tiles per workgroup (256 rows): one column-sum partial per 256 rows, loads of
                              // tile t+1 are in flight while tile t is split and stored
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ X, int ldx,
                                                              unsigned short* __restrict__ Th,
                                                              unsigned short* __restrict__ Tl, float* __restrict__ colpart,
                                                              unsigned short* __restrict__ Rh,
                                                              unsigned short* __restrict__ Rl, int ldp,
                                                              int M, int C, int Mp, int Rrows) {
  __shared__ float tile[2][64][65];
  const int c0 = blockIdx.x * 64, tid = threadIdx.x;
  const int lc = tid & 63, lr = tid >> 6;                 // load role: column lc, rows lr, lr+4, ...
  const int c = tid >> 2, r0 = (tid & 3) * 16;            // transposed-store role: column c, 16 rows from r0
  const int rr = tid >> 2, rc = (tid & 3) * 16;           // row-store role: row rr, 16 columns from rc
  float v[16];
  auto load = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + lr + 4 * i;
      v[i] = (m < M && c0 + lc < C) ? X[(size_t)m * ldx + c0 + lc] : 0.f;
    }
  };
  float s = 0.f;
  const int mbase = blockIdx.y * 64 * TS_TILES;
  load(mbase);
  for (int t = 0; t < TS_TILES; ++t) {
    const int m0 = mbase + 64 * t;
    if (m0 >= Mp) break;
    float(*tl)[65] = tile[t & 1];
#pragma unroll
    for (int i = 0; i < 16; ++i) tl[lr + 4 * i][lc] = v[i];
    __syncthreads();                                        // (the other buffer's readers finished one barrier ago)
    if (t + 1 < TS_TILES && m0 + 64 < Mp) load(m0 + 64);    // next tile's global loads fly over the stores below
    if (Rh && m0 + rr < Rrows) {                            // Rrows = Mp: rows [M, Mp) are written as zeros (the tile zero-fills
                                                            // them): the operand of the TN weight-gradient GEMM
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int c8 = c0 + rc + 8 * half;
        if (c8 < ldp) {           // ldp % 8 == 0; columns in [C, ldp) are zero (the tile zero-fills beyond C)
          unsigned int hi[4], lo[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            split2_bf16(tl[rr][rc + 8 * half + 2 * e], tl[rr][rc + 8 * half + 2 * e + 1], hi[e], lo[e]);
          *reinterpret_cast<uint4*>(Rh + (size_t)(m0 + rr) * ldp + c8) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
          *reinterpret_cast<uint4*>(Rl + (size_t)(m0 + rr) * ldp + c8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
      }
    }
    if (c0 + c < C) {
      unsigned int hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = tl[r0 + 2 * e][c], b = tl[r0 + 2 * e + 1][c];
        s += a + b;
        if (Th) split2_bf16(a, b, hi[e], lo[e]);
      }
      if (Th && m0 + r0 < Mp) {
        const size_t o = (size_t)(c0 + c) * Mp + m0 + r0;
        uint4* ph = reinterpret_cast<uint4*>(Th + o);
        uint4* pl = reinterpret_cast<uint4*>(Tl + o);
        ph[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]); ph[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
        pl[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]); pl[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
      }
    }
  }
  if (colpart) {           // the 4 row-groups of a column sit in adjacent lanes: one partial per column and workgroup,
    s += __shfl_xor(s, 1, 64);   // summed over the row blocks in a fixed order by colsum_reduce_kernel (no atomics:
    s += __shfl_xor(s, 2, 64);   // db is bit-reproducible from run to run)
    if ((tid & 3) == 0 && c0 + c < C) colpart[(size_t)blockIdx.y * C + c0 + c] = s;
  }
}

__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ colsum,
                                                            int n_part, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < n_part; ++r) s += part[(size_t)r * C + c];
  colsum[c] = s;
}

// LayerNorm backward, one wave per row.  y = (x - mean) * rstd * gamma + beta,  x = X (+ R)
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//   dgamma / dbeta: per-workgroup partial sums -> part[blockIdx][2][H], reduced by ln_param_reduce_kernel.
// DROP (gridmm_layernorm_dropout_bwd): the forward normalised dropout(X) + R; the mask is regenerated here, the residual
// branch receives the plain gradient (dR) and X's branch the masked, rescaled one (dX) -- the backward of the dropout
// costs one extra store instead of a launch and a pass.
template <int NV, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const float* __restrict__ X, int ldx, const float* __restrict__ R, int ldr, const float* __restrict__ gamma,
    float eps, const float* __restrict__ dY, int ldy, float* __restrict__ dX, int lddx, float* __restrict__ part,
    int M, int H, float* __restrict__ dR = nullptr, int lddr = 0, float drop_p = 0.f, unsigned long long seed = 0,
    const unsigned long long* __restrict__ seed_dev = nullptr, unsigned short* __restrict__ Ph = nullptr,
    unsigned short* __restrict__ Pl = nullptr) {
  __shared__ float s_g[4][MAX_H_BWD];
  __shared__ float s_b[4][MAX_H_BWD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  const int nv = H >> 2;
  float4 xv[NV], gv[NV], dv[NV];
  const bool live = row < M;
  float s = 0.f;
  float scale = 1.f;
  if constexpr (DROP) {
    if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
    scale = 1.0f / (1.0f - drop_p);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && c < nv) {
      x = reinterpret_cast<const float4*>(X + (size_t)row * ldx)[c];
      if constexpr (DROP) {
        const unsigned int e = (unsigned int)row * (unsigned int)H + 4u * (unsigned int)c;
        x.x = dropout_keep(seed, e, drop_p) ? __fmul_rn(x.x, scale) : 0.f;
        x.y = dropout_keep(seed, e + 1, drop_p) ? __fmul_rn(x.y, scale) : 0.f;
        x.z = dropout_keep(seed, e + 2, drop_p) ? __fmul_rn(x.z, scale) : 0.f;
        x.w = dropout_keep(seed, e + 3, drop_p) ? __fmul_rn(x.w, scale) : 0.f;
      }
      if (R) {
        const float4 r = reinterpret_cast<const float4*>(R + (size_t)row * ldr)[c];
        x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
      }
      s += (x.x + x.y) + (x.z + x.w);
    }
    xv[i] = x;
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (live && c < nv) {
      const float a = xv[i].x - mean, b = xv[i].y - mean, cc = xv[i].z - mean, d = xv[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), d = g, xh = g;
    if (live && c < nv) {
      d = reinterpret_cast<const float4*>(dY + (size_t)row * ldy)[c];
      const float4 gm = reinterpret_cast<const float4*>(gamma)[c];
      xh = make_float4((xv[i].x - mean) * rstd, (xv[i].y - mean) * rstd, (xv[i].z - mean) * rstd, (xv[i].w - mean) * rstd);
      g = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      sg += (g.x + g.y) + (g.z + g.w);
      sgx += (g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w);
    }
    xv[i] = xh; gv[i] = g; dv[i] = d;
  }
  const float mg = wave_sum(sg) / (float)H, mgx = wave_sum(sgx) / (float)H;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      if (live) {
        float4 o;
        o.x = rstd * (gv[i].x - mg - xv[i].x * mgx);
        o.y = rstd * (gv[i].y - mg - xv[i].y * mgx);
        o.z = rstd * (gv[i].z - mg - xv[i].z * mgx);
        o.w = rstd * (gv[i].w - mg - xv[i].w * mgx);
        if constexpr (DROP) {
          if (dR) reinterpret_cast<float4*>(dR + (size_t)row * lddr)[c] = o;
          const unsigned int e = (unsigned int)row * (unsigned int)H + 4u * (unsigned int)c;
          o.x = dropout_keep(seed, e, drop_p) ? o.x * scale : 0.f;
          o.y = dropout_keep(seed, e + 1, drop_p) ? o.y * scale : 0.f;
          o.z = dropout_keep(seed, e + 2, drop_p) ? o.z * scale : 0.f;
          o.w = dropout_keep(seed, e + 3, drop_p) ? o.w * scale : 0.f;
        }
        reinterpret_cast<float4*>(dX + (size_t)row * lddx)[c] = o;
        if (Ph) {          // the bf16 planes of dX (contiguous rows of H): the dY operand of the Linear in front of this LayerNorm
          uint2 hi, lo;
          split2_bf16(o.x, o.y, hi.x, lo.x);
          split2_bf16(o.z, o.w, hi.y, lo.y);
          reinterpret_cast<uint2*>(Ph + (size_t)row * H)[c] = hi;
          reinterpret_cast<uint2*>(Pl + (size_t)row * H)[c] = lo;
        }
      }
      // per-wave contributions to dgamma (dy * xhat) and dbeta (dy)
      reinterpret_cast<float4*>(s_g[wave])[c] = live ? make_float4(dv[i].x * xv[i].x, dv[i].y * xv[i].y, dv[i].z * xv[i].z, dv[i].w * xv[i].w)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(s_b[wave])[c] = live ? dv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    part[((size_t)blockIdx.x * 2 + 0) * H + c] = (s_g[0][c] + s_g[1][c]) + (s_g[2][c] + s_g[3][c]);
    part[((size_t)blockIdx.x * 2 + 1) * H + c] = (s_b[0][c] + s_b[1][c]) + (s_b[2][c] + s_b[3][c]);
  }
}

// dgamma[c] = sum_blocks part[blk][0][c], dbeta likewise.  Block = 32 columns x 32 row-lanes: lane r sums blocks
// r, r+32, ... (coalesced 128-B reads across the 32 columns), then a fixed-order LDS tree -> deterministic.
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ part,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               int nblk, int H) {
  __shared__ float s_g[32][33], s_b[32][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float g = 0.f, b = 0.f;
  if (c < H)
    for (int k = ry; k < nblk; k += 32) {
      g += part[((size_t)k * 2 + 0) * H + c];
      b += part[((size_t)k * 2 + 1) * H + c];
    }
  s_g[ry][cx] = g;
  s_b[ry][cx] = b;
  __syncthreads();
  if (ry == 0 && c < H) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) { sg += s_g[r][cx]; sb += s_b[r][cx]; }
    dgamma[c] = sg;
    dbeta[c] = sb;
  }
}

// mode 0: y = gelu_erf(x); mode 1: dx = dy * gelu'(x); mode 2: y = relu(x); mode 3: dx = dy * (x > 0)
__global__ void act_kernel(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ out,
                           size_t n4, int mode, unsigned short* __restrict__ Ph = nullptr,
                           unsigned short* __restrict__ Pl = nullptr) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(X)[i];
    float4 d = make_float4(1.f, 1.f, 1.f, 1.f);
    if (mode & 1) d = reinterpret_cast<const float4*>(dY)[i];
    const float xs[4] = {x.x, x.y, x.z, x.w}, ds[4] = {d.x, d.y, d.z, d.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = xs[e];
      if (mode == 0) o[e] = v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
      else if (mode == 1) {
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
        o[e] = ds[e] * (cdf + v * pdf);
      } else if (mode == 2) o[e] = fmaxf(v, 0.f);
      else o[e] = v > 0.f ? ds[e] : 0.f;
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (Ph) {                        // bf16 hi/lo planes of the result (same element order)
      uint2 hi, lo;
      split2_bf16(o[0], o[1], hi.x, lo.x);
      split2_bf16(o[2], o[3], hi.y, lo.y);
      reinterpret_cast<uint2*>(Ph)[i] = hi;
      reinterpret_cast<uint2*>(Pl)[i] = lo;
    }
  }
}

}  // namespace

extern "C" int gridmm_transpose_split(const float* X, int ldx, void* T_hi, void* T_lo, float* colsum,
                                      float* colsum_ws, void* R_hi, void* R_lo, int ldp, int M, int C, int Mp,
                                      gridmm_stream_t stream) {
  if (M <= 0 || C <= 0 || Mp < M || Mp % 32) return GRIDMM_EINVAL;
  if (R_hi && (!R_lo || ldp < C || ldp % 8)) return GRIDMM_EINVAL;
  if (colsum && !colsum_ws) return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  dim3 grid((C + 63) / 64, (Mp + 64 * TS_TILES - 1) / (64 * TS_TILES)), block(256);
  GRIDMM_LAUNCH(transpose_split_kernel, grid, block, 0, st, X, ldx, (unsigned short*)T_hi, (unsigned short*)T_lo,
                colsum_ws, (unsigned short*)R_hi, (unsigned short*)R_lo, ldp, M, C, Mp, M);
  GRIDMM_CHECK_LAUNCH();
  if (colsum) {
    GRIDMM_LAUNCH(colsum_reduce_kernel, dim3((C + 255) / 256), block, 0, st, colsum_ws, colsum, (int)grid.y, C);
    GRIDMM_CHECK_LAUNCH();
  }
  return GRIDMM_OK;
}

// X fp32 [M][C] -> row-major bf16 hi/lo planes [Mp][ldp] with rows [M, Mp) ZERO (Mp % 32 == 0) [+ colsum]: the one pass an
// activation / a gradient needs when the weight gradient runs as gridmm_linear_planes_tn -- the same planes are the A
// operand of the forward / dX GEMM (first M rows) and an operand of dW = dY^T X (all Mp rows); no transposed copy.
extern "C" int gridmm_split_rows_pad(const float* X, int ldx, void* R_hi, void* R_lo, int ldp, float* colsum,
                                     float* colsum_ws, int M, int C, int Mp, gridmm_stream_t stream) {
  if (M <= 0 || C <= 0 || Mp < M || Mp % 32 || !R_hi || !R_lo || ldp < C || ldp % 8) return GRIDMM_EINVAL;
  if (colsum && !colsum_ws) return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  dim3 grid((C + 63) / 64, (Mp + 64 * TS_TILES - 1) / (64 * TS_TILES)), block(256);
  GRIDMM_LAUNCH(transpose_split_kernel, grid, block, 0, st, X, ldx, (unsigned short*)nullptr, (unsigned short*)nullptr,
                colsum_ws, (unsigned short*)R_hi, (unsigned short*)R_lo, ldp, M, C, Mp, Mp);
  GRIDMM_CHECK_LAUNCH();
  if (colsum) {
    GRIDMM_LAUNCH(colsum_reduce_kernel, dim3((C + 255) / 256), block, 0, st, colsum_ws, colsum, (int)grid.y, C);
    GRIDMM_CHECK_LAUNCH();
  }
  return GRIDMM_OK;
}

extern "C" int gridmm_layernorm_dropout_bwd_planes(const float* X, const float* R, int ldr, const float* gamma, float eps,
                                                   const float* dY, float* dX, void* dX_hi, void* dX_lo, float* dR,
                                                   float* dgamma, float* dbeta, float* workspace, float p,
                                                   unsigned long long seed, const unsigned long long* seed_dev, int M, int H,
                                                   gridmm_stream_t stream) {
  if (M <= 0 || H <= 0 || H % 4 || H > MAX_H_BWD || (R && ldr % 4) || !(p >= 0.f && p < 1.f) ||
      (size_t)M * H >= (1ull << 32) || (dX_hi && !dX_lo))
    return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  const int nblk = (M + 3) / 4;
  dim3 grid(nblk), block(256);
  const int nv = (H / 4 + 63) / 64;
#define GRIDMM_LNBD(NV)                                                                                          \
  GRIDMM_LAUNCH((layernorm_bwd_kernel<NV, true>), grid, block, 0, st, X, H, R, ldr, gamma, eps, dY, H, dX, H, workspace, \
                M, H, dR, H, p, seed, seed_dev, (unsigned short*)dX_hi, (unsigned short*)dX_lo)
  if (nv == 1) GRIDMM_LNBD(1); else if (nv == 2) GRIDMM_LNBD(2); else if (nv == 3) GRIDMM_LNBD(3); else GRIDMM_LNBD(4);
#undef GRIDMM_LNBD
  GRIDMM_CHECK_LAUNCH();
  GRIDMM_LAUNCH(ln_param_reduce_kernel, dim3((H + 31) / 32), dim3(1024), 0, st, workspace, dgamma, dbeta, nblk, H);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_layernorm_dropout_bwd(const float* X, const float* R, int ldr, const float* gamma, float eps,
                                            const float* dY, float* dX, float* dR, float* dgamma, float* dbeta,
                                            float* workspace, float p, unsigned long long seed,
                                            const unsigned long long* seed_dev, int M, int H, gridmm_stream_t stream) {
  return gridmm_layernorm_dropout_bwd_planes(X, R, ldr, gamma, eps, dY, dX, nullptr, nullptr, dR, dgamma, dbeta, workspace, p,
                                             seed, seed_dev, M, H, stream);
}

extern "C" int gridmm_layernorm_bwd_planes(const float* X, int ldx, const float* R, int ldr, const float* gamma, float eps,
                                           const float* dY, int ldy, float* dX, int lddx, void* dX_hi, void* dX_lo,
                                           float* dgamma, float* dbeta, float* workspace, int M, int H,
                                           gridmm_stream_t stream) {
  if (M <= 0 || H <= 0 || H % 4 || H > MAX_H_BWD || ldx % 4 || ldy % 4 || lddx % 4 || (R && ldr % 4) || (dX_hi && !dX_lo))
    return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  const int nblk = (M + 3) / 4;
  dim3 grid(nblk), block(256);
  const int nv = (H / 4 + 63) / 64;
#define GRIDMM_LNB(NV)                                                                                      \
  GRIDMM_LAUNCH((layernorm_bwd_kernel<NV>), grid, block, 0, st, X, ldx, R, ldr, gamma, eps, dY, ldy, dX, lddx, \
                workspace, M, H, (float*)nullptr, 0, 0.f, 0ull, (const unsigned long long*)nullptr, (unsigned short*)dX_hi,  \
                (unsigned short*)dX_lo)
  if (nv == 1) GRIDMM_LNB(1); else if (nv == 2) GRIDMM_LNB(2); else if (nv == 3) GRIDMM_LNB(3); else GRIDMM_LNB(4);
#undef GRIDMM_LNB
  GRIDMM_CHECK_LAUNCH();
  GRIDMM_LAUNCH(ln_param_reduce_kernel, dim3((H + 31) / 32), dim3(1024), 0, st, workspace, dgamma, dbeta, nblk, H);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_layernorm_bwd(const float* X, int ldx, const float* R, int ldr, const float* gamma, float eps,
                                    const float* dY, int ldy, float* dX, int lddx, float* dgamma, float* dbeta,
                                    float* workspace, int M, int H, gridmm_stream_t stream) {
  return gridmm_layernorm_bwd_planes(X, ldx, R, ldr, gamma, eps, dY, ldy, dX, lddx, nullptr, nullptr, dgamma, dbeta, workspace, M,
                                     H, stream);
}

// _planes: also the bf16 hi/lo planes of the result (forward: the next Linear's A operand; backward modes 1 / 3: the dY
// operand of the Linear in front of the activation).
extern "C" int gridmm_activation_planes(const float* X, const float* dY, float* out, void* out_hi, void* out_lo, int64_t n,
                                        int mode, gridmm_stream_t stream) {
  if (n <= 0 || n % 4 || mode < 0 || mode > 3 || ((mode & 1) && !dY) || (out_hi && !out_lo)) return GRIDMM_EINVAL;
  const size_t n4 = (size_t)n / 4;
  unsigned grid = (unsigned)((n4 + 255) / 256);
  if (grid > 16384) grid = 16384;
  GRIDMM_LAUNCH(act_kernel, dim3(grid), dim3(256), 0, as_stream(stream), X, dY, out, n4, mode, (unsigned short*)out_hi,
                (unsigned short*)out_lo);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_activation(const float* X, const float* dY, float* out, int64_t n, int mode,
                                 gridmm_stream_t stream) {
  return gridmm_activation_planes(X, dY, out, nullptr, nullptr, n, mode, stream);
}

// ================================================================================================
// Backward of the instruction-relevance aggregation (gridmm_grid_aggregate) w.r.t. text = text_proj(txt):
//   w_j = max_l <x_j, t_l>,  a_j = softmax_{j in cell}(w_j),  cells[c] = sum_j a_j x_j
//   da_j = <dcells[c(j)], x_j>;  dw_j = a_j (da_j - sum_{j' in c} a_j' da_j');  dt[argmax_l(j)] += dw_j x_j
// (the slab itself is an input, not a parameter).  Two passes over the fp16 slab:
//   agg_bwd_points_kernel: da_j and argmax_l per point (4 sorted points per wave share every text row load)
//   agg_bwd_cells_kernel : per (cell, episode) softmax statistics, dw_j, atomic accumulation into dt
// ================================================================================================
namespace {

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
constexpr int AGG_P = 4;       // points per wave
constexpr int AGG_MAXV = 6;    // D <= 768: D/128 half2 per lane

template <int NV>
__global__ __launch_bounds__(256) void agg_bwd_points_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const float* __restrict__ text, const float* __restrict__ dcells, float* __restrict__ da,
    int32_t* __restrict__ amax, int cap, int D, int L) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  const int valid = cs[GRIDMM_CELLS];
  const int p0 = (blockIdx.x * 4 + wave) * AGG_P;
  if (p0 >= valid) return;
  const int32_t* perm_b = perm + (size_t)b * cap;
  float2 x[AGG_P][NV];
  int slot[AGG_P];
  float dav[AGG_P];
#pragma unroll
  for (int p = 0; p < AGG_P; ++p) {
    const int pos = min(p0 + p, valid - 1);
    slot[p] = perm_b[pos];
    // cell of sorted position pos: largest c with cs[c] <= pos (binary search over 197 boundaries)
    int lo = 0, hi = GRIDMM_CELLS;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cs[mid] <= pos) lo = mid; else hi = mid;
    }
    const f16x2_t* xr = reinterpret_cast<const f16x2_t*>(slab + ((size_t)b * cap + slot[p]) * D);
    const float2* dc = reinterpret_cast<const float2*>(dcells + ((size_t)b * GRIDMM_CELLS + lo) * D);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      x[p][i] = make_float2(0.f, 0.f);
      if (lane + 64 * i < D / 2) {
        const f16x2_t h = xr[lane + 64 * i];
        x[p][i] = make_float2((float)h[0], (float)h[1]);
        const float2 d = dc[lane + 64 * i];
        s += x[p][i].x * d.x + x[p][i].y * d.y;
      }
    }
    dav[p] = wave_sum(s);
  }
  float best[AGG_P];
  int arg[AGG_P];
#pragma unroll
  for (int p = 0; p < AGG_P; ++p) { best[p] = -3.0e38f; arg[p] = 0; }
  const float2* tb = reinterpret_cast<const float2*>(text + (size_t)b * L * D);
  for (int l = 0; l < L; ++l) {
    float2 t[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      t[i] = (lane + 64 * i < D / 2) ? tb[(size_t)l * (D / 2) + lane + 64 * i] : make_float2(0.f, 0.f);
#pragma unroll
    for (int p = 0; p < AGG_P; ++p) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) s += x[p][i].x * t[i].x + x[p][i].y * t[i].y;
      s = wave_sum(s);
      if (s > best[p]) { best[p] = s; arg[p] = l; }   // first maximum wins (torch.max returns the first index)
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int p = 0; p < AGG_P; ++p)
      if (p0 + p < valid) {
        da[(size_t)b * cap + slot[p]] = dav[p];
        amax[(size_t)b * cap + slot[p]] = arg[p];
      }
  }
}

__device__ __forceinline__ float block_reduce(float v, float* s_red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = s_red[0];
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, s_red[w]) : r + s_red[w];
  return r;
}

template <int NV>
__global__ __launch_bounds__(256) void agg_bwd_cells_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const float* __restrict__ relevance, const float* __restrict__ da, const int32_t* __restrict__ amax,
    float* __restrict__ dtext, int cap, int D, int L) {
  __shared__ float s_red[4];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  const int beg = cs[c], end = cs[c + 1];
  if (end <= beg) return;
  const int32_t* perm_b = perm + (size_t)b * cap;
  const float* w = relevance + (size_t)b * cap;
  const float* dab = da + (size_t)b * cap;
  float m = -3.0e38f;
  for (int p = beg + tid; p < end; p += 256) m = fmaxf(m, w[p]);     // relevance is stored by sorted position
  m = block_reduce(m, s_red, true);
  float z = 0.f, sa = 0.f;
  for (int p = beg + tid; p < end; p += 256) {
    const int s = perm_b[p];
    const float e = expf(w[p] - m);
    z += e;
    sa += e * dab[s];
  }
  z = block_reduce(z, s_red, false);
  sa = block_reduce(sa, s_red, false) / z;
  for (int p = beg + wave; p < end; p += 4) {
    const int s = perm_b[p];
    const float a = expf(w[p] - m) / z;
    const float dw = a * (dab[s] - sa);
    const f16x2_t* xr = reinterpret_cast<const f16x2_t*>(slab + ((size_t)b * cap + s) * D);
    float* dt = dtext + ((size_t)b * L + amax[(size_t)b * cap + s]) * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lane + 64 * i < D / 2) {
        const f16x2_t h = xr[lane + 64 * i];
        atomicAdd(dt + 2 * (lane + 64 * i), dw * (float)h[0]);
        atomicAdd(dt + 2 * (lane + 64 * i) + 1, dw * (float)h[1]);
      }
    }
  }
}


// ---- routed variant: the forward already found the arg-max token of every point (amax, by sorted position), so the
// backward is three streaming passes with no search, no atomics and a deterministic result:
//   agg_bwd_da_kernel      da_p = <dcells[cell(p)], x_p>                                  (4 sorted points per wave)
//   agg_bwd_dw_kernel      per (cell, episode): softmax statistics -> dw_p = a_p (da_p - sum_cell a da)
//   agg_bwd_gather_kernel  per (token, episode): dt[l] = sum over the points routed to l of dw_p x_p
template <int NV>
__global__ __launch_bounds__(256) void agg_bwd_da_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const float* __restrict__ dcells, float* __restrict__ da, int cap, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  const int valid = cs[GRIDMM_CELLS];
  const int p0 = (blockIdx.x * 4 + wave) * AGG_P;
  if (p0 >= valid) return;
  const int32_t* perm_b = perm + (size_t)b * cap;
#pragma unroll
  for (int p = 0; p < AGG_P; ++p) {
    const int pos = min(p0 + p, valid - 1);
    int lo = 0, hi = GRIDMM_CELLS;              // cell of sorted position pos: largest c with cs[c] <= pos
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cs[mid] <= pos) lo = mid; else hi = mid;
    }
    const f16x2_t* xr = reinterpret_cast<const f16x2_t*>(slab + ((size_t)b * cap + perm_b[pos]) * D);
    const float2* dc = reinterpret_cast<const float2*>(dcells + ((size_t)b * GRIDMM_CELLS + lo) * D);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lane + 64 * i < D / 2) {
        const f16x2_t h = xr[lane + 64 * i];
        const float2 d = dc[lane + 64 * i];
        s += (float)h[0] * d.x + (float)h[1] * d.y;
      }
    }
    s = wave_sum(s);
    if (lane == 0 && p0 + p < valid) da[(size_t)b * cap + pos] = s;
  }
}

__global__ __launch_bounds__(256) void agg_bwd_dw_kernel(const int32_t* __restrict__ cell_start,
                                                         const float* __restrict__ relevance,
                                                         const float* __restrict__ da, float* __restrict__ dw, int cap) {
  __shared__ float s_red[4];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  const int beg = cs[c], end = cs[c + 1];
  if (end <= beg) return;
  const float* w = relevance + (size_t)b * cap;      // everything here is indexed by sorted position
  const float* dab = da + (size_t)b * cap;
  float m = -3.0e38f;
  for (int p = beg + tid; p < end; p += 256) m = fmaxf(m, w[p]);
  m = block_reduce(m, s_red, true);
  float z = 0.f, sa = 0.f;
  for (int p = beg + tid; p < end; p += 256) {
    const float e = expf(w[p] - m);
    z += e;
    sa += e * dab[p];
  }
  z = block_reduce(z, s_red, false);
  sa = block_reduce(sa, s_red, false) / z;
  for (int p = beg + tid; p < end; p += 256) dw[(size_t)b * cap + p] = expf(w[p] - m) / z * (dab[p] - sa);
}

template <int NV>
__global__ __launch_bounds__(256) void agg_bwd_gather_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const int32_t* __restrict__ amax, const float* __restrict__ dw, float* __restrict__ dtext, int cap, int D, int L) {
  __shared__ float s_acc[3][AGG_MAXV * 128];
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int valid = cell_start[(size_t)b * (GRIDMM_CELLS + 2) + GRIDMM_CELLS];
  const int32_t* am = amax + (size_t)b * cap;
  const int32_t* perm_b = perm + (size_t)b * cap;
  const float* dwb = dw + (size_t)b * cap;
  float2 acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = make_float2(0.f, 0.f);
  for (int p0 = wave * 64; p0 < valid; p0 += 256) {      // each wave scans its own 64-point groups, in point order
    const int p = p0 + lane;
    unsigned long long hit = __ballot(p < valid && am[p] == l);
    while (hit) {
      const int q = p0 + __builtin_ctzll(hit);
      hit &= hit - 1;
      const float g = dwb[q];
      const f16x2_t* xr = reinterpret_cast<const f16x2_t*>(slab + ((size_t)b * cap + perm_b[q]) * D);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < D / 2) {
          const f16x2_t h = xr[lane + 64 * i];
          acc[i].x += g * (float)h[0];
          acc[i].y += g * (float)h[1];
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < D / 2) reinterpret_cast<float2*>(s_acc[wave - 1])[lane + 64 * i] = acc[i];
  }
  __syncthreads();
  if (wave == 0) {                                         // fixed summation order: deterministic
    float2* out = reinterpret_cast<float2*>(dtext + ((size_t)b * L + l) * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lane + 64 * i < D / 2) {
        float2 r = acc[i];
        for (int w = 0; w < 3; ++w) {
          const float2 o = reinterpret_cast<const float2*>(s_acc[w])[lane + 64 * i];
          r.x += o.x; r.y += o.y;
        }
        out[lane + 64 * i] = r;
      }
    }
  }
}

// ---- round 5 forms of the two slab passes (D % 8 == 0).  The first versions above spent their time OUTSIDE the slab reads:
// the da pass did a 8-step binary search through L2 per point, and the gather pass gave every token its own workgroup that
// scanned all points -- the routing concentrates on a few tokens, so a handful of workgroups did most of the row reads one
// dependent load chain at a time (1.4 ms per call at the 36 x 196 x 512 shape, 13 ms of a 128 ms fine-tune iteration).
//   agg_bwd_da2_kernel     a wave owns 16 consecutive sorted points: ONE search for the first, then the cell advances with
//                          the position; 16-byte row loads; the dcells row stays in registers while the cell does not change
//   agg_bwd_gather2_kernel point-balanced: a workgroup owns a contiguous chunk of the sorted order and a 256-column half of
//                          the rows, each of its two waves a private [L][128] fp32 table in LDS that it updates in point
//                          order (no atomics, deterministic); the chunk tables go to a workspace
//   agg_bwd_gsum_kernel    dtext = sum of the chunk tables in chunk order
constexpr int DA2_P = 16;
constexpr int GCH = 32;         // chunks per episode of the gather pass, at most (the workspace is sized for it); 8 / 16 / 32 by capacity

template <int NV8>              // 16-byte row pieces per lane: D <= 512 * NV8
__global__ __launch_bounds__(256) void agg_bwd_da2_kernel(const _Float16* __restrict__ slab, const int32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ cell_start, const float* __restrict__ dcells,
                                                          float* __restrict__ da, int cap, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  const int valid = cs[GRIDMM_CELLS];
  const int p0 = (blockIdx.x * 4 + wave) * DA2_P;
  if (p0 >= valid) return;
  const int32_t* perm_b = perm + (size_t)b * cap;
  int lo = 0, hi = GRIDMM_CELLS;                // cell of sorted position p0: largest c with cs[c] <= p0 (wave-uniform)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cs[mid] <= p0) lo = mid; else hi = mid;
  }
  int next = cs[lo + 1], loaded = -1;
  float dc[NV8][8];
  const int n = min(DA2_P, valid - p0);
  const int myperm = lane < n ? perm_b[p0 + lane] : 0;      // the wave's row ids in one load
  float keep = 0.f;
  for (int p = 0; p < n; ++p) {
    const int pos = p0 + p;
    while (pos >= next && lo < GRIDMM_CELLS - 1) { ++lo; next = cs[lo + 1]; }
    if (lo != loaded) {                          // (wave-uniform) the gradient row of this cell
#pragma unroll
      for (int i = 0; i < NV8; ++i) {
        const int d0 = 8 * lane + 512 * i;
        if (d0 < D) {
          const float4 u = *reinterpret_cast<const float4*>(dcells + ((size_t)b * GRIDMM_CELLS + lo) * D + d0);
          const float4 v = *reinterpret_cast<const float4*>(dcells + ((size_t)b * GRIDMM_CELLS + lo) * D + d0 + 4);
          dc[i][0] = u.x; dc[i][1] = u.y; dc[i][2] = u.z; dc[i][3] = u.w; dc[i][4] = v.x; dc[i][5] = v.y; dc[i][6] = v.z; dc[i][7] = v.w;
        }
      }
      loaded = lo;
    }
    const int row = __builtin_amdgcn_readlane(myperm, p);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      const int d0 = 8 * lane + 512 * i;
      if (d0 < D) {
        const f16x8_t h = *reinterpret_cast<const f16x8_t*>(slab + ((size_t)b * cap + row) * D + d0);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)h[e] * dc[i][e];
      }
    }
    s = wave_sum(s);
    if (lane == p) keep = s;
  }
  if (lane < n) da[(size_t)b * cap + p0 + lane] = keep;
}

__global__ __launch_bounds__(128) void agg_bwd_gather2_kernel(const _Float16* __restrict__ slab, const int32_t* __restrict__ perm,
                                                              const int32_t* __restrict__ cell_start,
                                                              const int32_t* __restrict__ amax, const float* __restrict__ dw,
                                                              float* __restrict__ part, int cap, int D, int L, int gch) {
  extern __shared__ __attribute__((aligned(16))) float2 s_tab[];          // [2 waves][L][64 lanes] (two dims per lane)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int halves = (D + 255) / 256;
  const int chunk = blockIdx.x / halves, half = blockIdx.x % halves, b = blockIdx.y;
  const int valid = cell_start[(size_t)b * (GRIDMM_CELLS + 2) + GRIDMM_CELLS];
  const int per = ((valid + gch - 1) / gch + 63) / 64 * 64;               // points per chunk, whole 64-point groups
  const int beg = chunk * per, end = min(valid, beg + per);
  const int d0 = half * 256 + wave * 128 + 2 * lane;                       // this lane's two columns
  const bool col_ok = d0 < D;
  float2* tab = s_tab + (size_t)wave * L * 64;
  for (int l = 0; l < L; ++l) tab[l * 64 + lane] = make_float2(0.f, 0.f);
  const int32_t* am = amax + (size_t)b * cap;
  const int32_t* perm_b = perm + (size_t)b * cap;
  const float* dwb = dw + (size_t)b * cap;
  const _Float16* sb = slab + (size_t)b * cap * D + (col_ok ? d0 : 0);
  for (int g0 = beg; g0 < end; g0 += 64) {
    const int p = g0 + lane;
    const bool ok = p < end;
    const float gv = ok ? dwb[p] : 0.f;
    const int lv = ok ? am[p] : 0, rv = ok ? perm_b[p] : 0;
    const int cnt = min(64, end - g0);
    for (int i0 = 0; i0 < cnt; i0 += 4) {                                // four rows in flight, then their four updates in order
      f16x2_t x[4];
      float gq[4];
      int lq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u, 63);
        gq[u] = __shfl(gv, i, 64);
        lq[u] = __shfl(lv, i, 64);
        const int r = __shfl(rv, i, 64);
        x[u] = *reinterpret_cast<const f16x2_t*>(sb + (size_t)r * D);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i0 + u < cnt) {
          float2 t = tab[lq[u] * 64 + lane];
          t.x += gq[u] * (float)x[u][0];
          t.y += gq[u] * (float)x[u][1];
          tab[lq[u] * 64 + lane] = t;
        }
      }
    }
  }
  if (col_ok) {
    float* out = part + (((size_t)b * gch + chunk) * L) * D + d0;
    for (int l = 0; l < L; ++l) *reinterpret_cast<float2*>(out + (size_t)l * D) = tab[l * 64 + lane];
  }
}

__global__ void agg_bwd_gsum_kernel(const float* __restrict__ part, float* __restrict__ dtext, size_t per_b4, int gch) {
  const size_t n4 = per_b4 * gridDim.y;
  (void)n4;
  const int b = blockIdx.y;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_b4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<const float4*>(part)[((size_t)b * gch) * per_b4 + i];
    for (int c = 1; c < gch; ++c) {
      const float4 v = reinterpret_cast<const float4*>(part)[((size_t)b * gch + c) * per_b4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(dtext)[(size_t)b * per_b4 + i] = a;
  }
}

}  // namespace

extern "C" int gridmm_grid_aggregate_bwd(const void* slab, const int32_t* perm, const int32_t* cell_start,
                                         const float* relevance, const float* text, const float* dcells,
                                         float* dtext, float* da_ws, int32_t* amax_ws, int B, int cap, int D, int L,
                                         gridmm_stream_t stream) {
  if (B <= 0 || cap <= 0 || L <= 0 || D <= 0 || D % 2 || D > 128 * AGG_MAXV) return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  if (!gridmm_zero_f32(dtext, (size_t)B * L * D, st)) return GRIDMM_ELAUNCH;
  const int nv = (D + 127) / 128;
  dim3 gp((cap + 4 * AGG_P - 1) / (4 * AGG_P), B), gc(GRIDMM_CELLS, B), block(256);
#define GRIDMM_AGGB(NV)                                                                                          \
  do {                                                                                                           \
    GRIDMM_LAUNCH((agg_bwd_points_kernel<NV>), gp, block, 0, st, (const _Float16*)slab, perm, cell_start, text,  \
                  dcells, da_ws, amax_ws, cap, D, L);                                                            \
    GRIDMM_CHECK_LAUNCH();                                                                                       \
    GRIDMM_LAUNCH((agg_bwd_cells_kernel<NV>), gc, block, 0, st, (const _Float16*)slab, perm, cell_start,         \
                  relevance, da_ws, amax_ws, dtext, cap, D, L);                                                  \
    GRIDMM_CHECK_LAUNCH();                                                                                       \
  } while (0)
  switch (nv) {
    case 1: GRIDMM_AGGB(1); break;
    case 2: GRIDMM_AGGB(2); break;
    case 3: GRIDMM_AGGB(3); break;
    case 4: GRIDMM_AGGB(4); break;
    case 5: GRIDMM_AGGB(5); break;
    default: GRIDMM_AGGB(6); break;
  }
#undef GRIDMM_AGGB
  return GRIDMM_OK;
}

extern "C" size_t gridmm_grid_aggregate_bwd_workspace(int B, int D, int L) {
  return (size_t)B * GCH * L * D * sizeof(float);
}

extern "C" int gridmm_grid_aggregate_bwd_routed(const void* slab, const int32_t* perm, const int32_t* cell_start,
                                                const float* relevance, const int32_t* amax, const float* dcells,
                                                float* dtext, float* da_ws, float* dw_ws, float* part_ws, int B, int cap, int D,
                                                int L, gridmm_stream_t stream) {
  if (B <= 0 || cap <= 0 || L <= 0 || D <= 0 || D % 2 || D > 128 * AGG_MAXV) return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  const size_t tab_bytes = (size_t)2 * L * 64 * sizeof(float2);
  if (part_ws && D % 8 == 0 && (D * L) % 4 == 0 && tab_bytes <= 80 * 1024) {      // the point-balanced forms (see above)
    dim3 gp2((cap + 4 * DA2_P - 1) / (4 * DA2_P), B), gc2(GRIDMM_CELLS, B);
    if (D <= 512) GRIDMM_LAUNCH((agg_bwd_da2_kernel<1>), gp2, dim3(256), 0, st, (const _Float16*)slab, perm, cell_start, dcells, da_ws, cap, D);
    else GRIDMM_LAUNCH((agg_bwd_da2_kernel<2>), gp2, dim3(256), 0, st, (const _Float16*)slab, perm, cell_start, dcells, da_ws, cap, D);
    GRIDMM_CHECK_LAUNCH();
    GRIDMM_LAUNCH(agg_bwd_dw_kernel, gc2, dim3(256), 0, st, cell_start, relevance, da_ws, dw_ws, cap);
    GRIDMM_CHECK_LAUNCH();
    const int halves = (D + 255) / 256;
    if (tab_bytes > 48 * 1024) {      // more dynamic LDS than the default limit: raise it (idempotent; a property of the kernel)
      static bool raised = false;
      if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(agg_bwd_gather2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                80 * 1024) != hipSuccess)
          return GRIDMM_ELAUNCH;
        raised = true;
      }
    }
    // chunks per episode: the pass is a chain of dependent row loads + LDS updates per wave, so deep memories (fine-tune
    // rollouts: tens of thousands of points per episode) want more, shorter chunks; their tables cost a write + a read each
    const int gch = cap >= 32768 ? 32 : (cap >= 16384 ? 16 : 8);
    GRIDMM_LAUNCH(agg_bwd_gather2_kernel, dim3(gch * halves, B), dim3(128), tab_bytes, st, (const _Float16*)slab, perm, cell_start,
                  amax, dw_ws, part_ws, cap, D, L, gch);
    GRIDMM_CHECK_LAUNCH();
    const size_t per_b4 = (size_t)L * D / 4;
    GRIDMM_LAUNCH(agg_bwd_gsum_kernel, dim3((unsigned)((per_b4 + 255) / 256 > 64 ? 64 : (per_b4 + 255) / 256), B), dim3(256), 0, st,
                  (const float*)part_ws, dtext, per_b4, gch);
    GRIDMM_CHECK_LAUNCH();
    return GRIDMM_OK;
  }
  const int nv = (D + 127) / 128;
  dim3 gp((cap + 4 * AGG_P - 1) / (4 * AGG_P), B), gc(GRIDMM_CELLS, B), gl(L, B), block(256);
#define GRIDMM_AGGR(NV)                                                                                          \
  do {                                                                                                           \
    GRIDMM_LAUNCH((agg_bwd_da_kernel<NV>), gp, block, 0, st, (const _Float16*)slab, perm, cell_start, dcells,    \
                  da_ws, cap, D);                                                                                \
    GRIDMM_CHECK_LAUNCH();                                                                                       \
    GRIDMM_LAUNCH(agg_bwd_dw_kernel, gc, block, 0, st, cell_start, relevance, da_ws, dw_ws, cap);                \
    GRIDMM_CHECK_LAUNCH();                                                                                       \
    GRIDMM_LAUNCH((agg_bwd_gather_kernel<NV>), gl, block, 0, st, (const _Float16*)slab, perm, cell_start, amax,  \
                  dw_ws, dtext, cap, D, L);                                                                      \
    GRIDMM_CHECK_LAUNCH();                                                                                       \
  } while (0)
  switch (nv) {
    case 1: GRIDMM_AGGR(1); break;
    case 2: GRIDMM_AGGR(2); break;
    case 3: GRIDMM_AGGR(3); break;
    case 4: GRIDMM_AGGR(4); break;
    case 5: GRIDMM_AGGR(5); break;
    default: GRIDMM_AGGR(6); break;
  }
#undef GRIDMM_AGGR
  return GRIDMM_OK;
}
