// gridmm_linear: C = act(A W^T + bias) (+ residual) with the contraction on MFMA bf16
// 16x16x32 tiles as a 3-term split (hi*hi + lo*hi + hi*lo, fp32 accumulate).
//
// Why 3 terms: plain bf16 operands move the navigation logits by ~2e-2 and fp16 operands by
// ~2e-3 against the fp32 reference (measured with the oracle); the 3-term
// split stays at ~1e-5 while still running on the bf16 matrix pipe (833 TFLOP/s effective peak
// vs 157 TFLOP/s for f32 MFMA).
//
// Tiling (wave = 64 lanes): a BM x BN block tile per 256-thread workgroup (2x2 waves), BK = 32
// per LDS stage = one 16x16x32 MFMA k-step.  fp32 A is split into bf16 hi/lo planes on the way
// into LDS (global -> regs -> cvt -> ds_write); W arrives pre-split (gridmm_split_weight).
// Register prefetch of stage k+1 is issued before the MFMAs of stage k (async-stage split), one
// LDS buffer.  The LDS image is row-major [row][32 bf16] with the 16-B chunk index XORed by
// bit 3 of the row so every ds_read_b128 lane group touches 16 distinct 16-B bank slots.
#include "common.h"

namespace {

constexpr int BK = 32;

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ (((row >> 3) & 1) << 1); }

template <int BM, int BN, int ACT, bool VEC_A>
__global__ __launch_bounds__(256) void linear_kernel(
    const float* __restrict__ A, int lda, const unsigned short* __restrict__ Whi,
    const unsigned short* __restrict__ Wlo, int Kp, const float* __restrict__ bias,
    const float* __restrict__ R, int ldr, float* __restrict__ C, int ldc, int M, int N, int K) {
  constexpr int TM = BM / 32;  // 16x16 MFMA tiles per wave along M (2 waves along M)
  constexpr int TN = BN / 32;
  constexpr int A_F4 = BM * BK / 4 / 256;      // float4 loads of A per thread per stage
  constexpr int W_CH = 2 * BN * BK / 8 / 256;  // 16-B chunks of W (both planes) per thread
  static_assert(A_F4 >= 1 && W_CH >= 1, "tile too small for 256 threads");

  __shared__ __attribute__((aligned(16))) unsigned short sA[2][BM * BK];  // [hi|lo][row][32]
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][BN * BK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  float4 ra[A_F4];
  uint4 rw[W_CH];

  auto load_stage = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int f = tid + i * 256, row = f >> 3, c4 = f & 7;
      const int m = bm + row, k = k0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        const float* p = A + (size_t)m * lda + k;
        if (VEC_A) {
          if (k < K) v = *reinterpret_cast<const float4*>(p);  // K % 4 == 0 here
        } else {
          if (k + 0 < K) v.x = p[0];
          if (k + 1 < K) v.y = p[1];
          if (k + 2 < K) v.z = p[2];
          if (k + 3 < K) v.w = p[3];
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
      const int g = tid + i * 256;
      const int plane = g / (BN * 4), rem = g % (BN * 4), row = rem >> 2, c = rem & 3;
      const int n = bn + row;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (n < N) {
        const unsigned short* base = plane ? Wlo : Whi;
        v = *reinterpret_cast<const uint4*>(base + (size_t)n * Kp + k0 + c * 8);
      }
      rw[i] = v;
    }
  };

  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int f = tid + i * 256, row = f >> 3, c4 = f & 7;
      const float x[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      u16x4_t hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned short h = f32_to_bf16_rne(x[e]);
        hi[e] = h;
        lo[e] = f32_to_bf16_rne(x[e] - bf16_bits_to_f32(h));
      }
      const int off = row * BK + swz_chunk(row, c4 >> 1) * 8 + (c4 & 1) * 4;
      *reinterpret_cast<u16x4_t*>(&sA[0][off]) = hi;
      *reinterpret_cast<u16x4_t*>(&sA[1][off]) = lo;
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
      const int g = tid + i * 256;
      const int plane = g / (BN * 4), rem = g % (BN * 4), row = rem >> 2, c = rem & 3;
      *reinterpret_cast<uint4*>(&sB[plane][row * BK + swz_chunk(row, c) * 8]) = rw[i];
    }
  };

  const int nk = (K + BK - 1) / BK;
  load_stage(0);
  store_stage();
  __syncthreads();

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_stage((kt + 1) * BK);

    bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wr * (BM / 2) + i * 16 + frow;
      const int off = row * BK + swz_chunk(row, fchunk) * 8;
      ah[i] = *reinterpret_cast<const bf16x8_t*>(&sA[0][off]);
      al[i] = *reinterpret_cast<const bf16x8_t*>(&sA[1][off]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wc * (BN / 2) + j * 16 + frow;
      const int off = row * BK + swz_chunk(row, fchunk) * 8;
      bh[j] = *reinterpret_cast<const bf16x8_t*>(&sB[0][off]);
      bl[j] = *reinterpret_cast<const bf16x8_t*>(&sB[1][off]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    __syncthreads();
    if (kt + 1 < nk) {
      store_stage();
      __syncthreads();
    }
  }

  // epilogue: lane holds rows (lane>>4)*4 + r, column lane&15 of every 16x16 tile
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = bn + wc * (BN / 2) + j * 16 + (lane & 15);
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = bm + wr * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
        float v = acc[i][j][r] + bv;
        if (ACT == GRIDMM_ACT_GELU) v = v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
        if (ACT == GRIDMM_ACT_RELU) v = fmaxf(v, 0.f);
        if (R) v += R[(size_t)m * ldr + n];
        C[(size_t)m * ldc + n] = v;
      }
    }
  }
}

__global__ void split_weight_kernel(const float* __restrict__ W, unsigned short* __restrict__ hi,
                                    unsigned short* __restrict__ lo, int N, int K, int Kp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * Kp) return;
  const int n = (int)(i / Kp), k = (int)(i % Kp);
  const float x = (k < K) ? W[(size_t)n * K + k] : 0.f;
  const unsigned short h = f32_to_bf16_rne(x);
  hi[i] = h;
  lo[i] = f32_to_bf16_rne(x - bf16_bits_to_f32(h));
}

template <int BM, int BN>
int launch_linear(const float* A, int lda, const unsigned short* Whi, const unsigned short* Wlo,
                  int Kp, const float* bias, const float* R, int ldr, float* C, int ldc, int M,
                  int N, int K, int act, hipStream_t st) {
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM), block(256);
  const bool vec = (lda % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
#define GRIDMM_LAUNCH_LIN(ACT, VEC)                                                            \
  GRIDMM_LAUNCH((linear_kernel<BM, BN, ACT, VEC>), grid, block, 0, st, A, lda, Whi, Wlo, \
                     Kp, bias, R, ldr, C, ldc, M, N, K)
  if (vec) {
    if (act == GRIDMM_ACT_NONE) GRIDMM_LAUNCH_LIN(GRIDMM_ACT_NONE, true);
    else if (act == GRIDMM_ACT_GELU) GRIDMM_LAUNCH_LIN(GRIDMM_ACT_GELU, true);
    else GRIDMM_LAUNCH_LIN(GRIDMM_ACT_RELU, true);
  } else {
    if (act == GRIDMM_ACT_NONE) GRIDMM_LAUNCH_LIN(GRIDMM_ACT_NONE, false);
    else if (act == GRIDMM_ACT_GELU) GRIDMM_LAUNCH_LIN(GRIDMM_ACT_GELU, false);
    else GRIDMM_LAUNCH_LIN(GRIDMM_ACT_RELU, false);
  }
#undef GRIDMM_LAUNCH_LIN
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

}  // namespace

extern "C" int gridmm_split_weight(const float* W, void* hi, void* lo, int N, int K, int Kp,
                                   gridmm_stream_t stream) {
  if (N <= 0 || K <= 0 || Kp < K || Kp % 32) return GRIDMM_EINVAL;
  const size_t total = (size_t)N * Kp;
  GRIDMM_LAUNCH(split_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     as_stream(stream), W, (unsigned short*)hi, (unsigned short*)lo, N, K, Kp);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_linear(const float* A, int lda, const void* W_hi, const void* W_lo, int Kp,
                             const float* bias, const float* residual, int ldr, float* C, int ldc,
                             int M, int N, int K, int act, gridmm_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || Kp < K || Kp % 32 || act < 0 || act > 2) return GRIDMM_EINVAL;
  const unsigned short* hi = (const unsigned short*)W_hi;
  const unsigned short* lo = (const unsigned short*)W_lo;
  hipStream_t st = as_stream(stream);
  // 128x128 tiles when they still give >= ~2 waves of workgroups over the 256 CUs, else 64x64.
  const long wg128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (wg128 >= 384)
    return launch_linear<128, 128>(A, lda, hi, lo, Kp, bias, residual, ldr, C, ldc, M, N, K, act, st);
  return launch_linear<64, 64>(A, lda, hi, lo, Kp, bias, residual, ldr, C, ldc, M, N, K, act, st);
}
