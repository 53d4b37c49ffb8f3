// gridmm_linear_planes: the hot-path GEMM.  Same contraction as gridmm_linear (MFMA bf16 16x16x32,
// 3-term split, fp32 accumulate) but BOTH operands arrive as pre-split bf16 hi/lo planes, so the
// tile pipeline has no conversion VALU and no VGPR staging:
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction = 16 rows x 64 B),
//     double-buffered, one barrier per k-step; the XOR swizzle of the LDS image is applied on the
//     per-lane SOURCE address (the DMA destination is lane-linear), the same involution on the read;
//   * epilogue through LDS: the fp32 accumulators are transposed per wave so that bias / activation /
//     residual and the stores are row-wise 128-bit accesses; the epilogue can emit the result as fp32
//     and/or as bf16 hi/lo planes (the next GEMM's A operand).
// Producers of activations (LayerNorm, attention, GELU epilogue) write the planes directly, see
// rowops.hip / attention.hip.  K % 32 == 0 and 16-byte aligned rows are required here; everything
// else goes through gridmm_linear.
#include "common.h"

namespace {

constexpr int BK = 32;

__device__ __forceinline__ int swz(int row) { return ((row >> 3) & 1) << 1; }

__device__ __forceinline__ void dma16(const unsigned short* gsrc, unsigned short* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int ACT>
__global__ __launch_bounds__(256) void linear_planes_kernel(
    const unsigned short* __restrict__ Ahi, const unsigned short* __restrict__ Alo, int lda,
    const unsigned short* __restrict__ Whi, const unsigned short* __restrict__ Wlo, int Kp,
    const float* __restrict__ bias, const float* __restrict__ R, int ldr, float* __restrict__ C, int ldc,
    unsigned short* __restrict__ Chi, unsigned short* __restrict__ Clo, int ldp, int M, int N, int K) {
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int STAGE = (2 * BM + 2 * BN) * BK;          // u16 elements per stage: Ahi|Alo|Whi|Wlo
  constexpr int PIECES = (2 * BM + 2 * BN) / 16;         // 1-KiB DMA pieces per stage
  constexpr int PPW = PIECES / 4;                        // per wave
  static_assert(PIECES % 4 == 0, "tile must split evenly over 4 waves");
  constexpr int EPI = (BM / 2) * (BN / 2);               // floats per wave in the epilogue transpose
  constexpr int LDS_U16 = (2 * STAGE * 2 > 4 * EPI * 4 ? 2 * STAGE : 4 * EPI * 2);
  __shared__ __attribute__((aligned(16))) unsigned short smem[LDS_U16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;

  // DMA plan of this wave: piece p covers 16 rows of one plane
  const unsigned short* src[PPW];
  int dst[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave * PPW + i;
    int plane, r0;
    if (p < BM / 16) { plane = 0; r0 = p * 16; }
    else if (p < 2 * BM / 16) { plane = 1; r0 = (p - BM / 16) * 16; }
    else if (p < (2 * BM + BN) / 16) { plane = 2; r0 = (p - 2 * BM / 16) * 16; }
    else { plane = 3; r0 = (p - (2 * BM + BN) / 16) * 16; }
    const int row = r0 + (lane >> 2);
    const int chunk = (lane & 3) ^ swz(row);
    if (plane < 2) {
      const int m = min(bm + row, M - 1);
      src[i] = (plane == 0 ? Ahi : Alo) + (size_t)m * lda + chunk * 8;
      dst[i] = plane * BM * BK + r0 * BK;
    } else {
      const int n = min(bn + row, N - 1);
      src[i] = (plane == 2 ? Whi : Wlo) + (size_t)n * Kp + chunk * 8;
      dst[i] = 2 * BM * BK + (plane - 2) * BN * BK + r0 * BK;
    }
  }

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
#pragma unroll
  for (int i = 0; i < PPW; ++i) dma16(src[i], smem + dst[i]);
  __syncthreads();  // (waits vmcnt(0): stage 0 landed)

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned short* cur = smem + (kt & 1) * STAGE;
    if (kt + 1 < nk) {
      unsigned short* nxt = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
      for (int i = 0; i < PPW; ++i) dma16(src[i] + (kt + 1) * BK, nxt + dst[i]);
    }
    bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wr * (BM / 2) + i * 16 + frow;
      const int off = row * BK + (fchunk ^ swz(row)) * 8;
      ah[i] = *reinterpret_cast<const bf16x8_t*>(cur + off);
      al[i] = *reinterpret_cast<const bf16x8_t*>(cur + BM * BK + off);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wc * (BN / 2) + j * 16 + frow;
      const int off = 2 * BM * BK + row * BK + (fchunk ^ swz(row)) * 8;
      bh[j] = *reinterpret_cast<const bf16x8_t*>(cur + off);
      bl[j] = *reinterpret_cast<const bf16x8_t*>(cur + BN * BK + off);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    __syncthreads();  // next stage landed (vmcnt(0)) and everyone is done reading `cur`
  }

  // ---- epilogue: per-wave transpose through LDS, then row-wise 128-bit accesses
  constexpr int WM = BM / 2, WN = BN / 2;  // wave sub-tile
  float* ep = reinterpret_cast<float*>(smem) + wave * EPI;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(i * 16 + (lane >> 4) * 4 + r) * WN + j * 16 + (lane & 15)] = acc[i][j][r];
  __builtin_amdgcn_wave_barrier();
  constexpr int F4_PER_ROW = WN / 4;                // float4 per sub-tile row
  constexpr int ROWS_PER_IT = 64 / F4_PER_ROW;
  const int c4 = lane % F4_PER_ROW, rr = lane / F4_PER_ROW;
  const int n0 = bn + wc * WN + c4 * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias && n0 < N) bv = *reinterpret_cast<const float4*>(bias + n0);  // N % 4 == 0
#pragma unroll
  for (int it = 0; it < WM / ROWS_PER_IT; ++it) {
    const int row = it * ROWS_PER_IT + rr;
    const int m = bm + wr * WM + row;
    float4 v = *reinterpret_cast<const float4*>(ep + row * WN + c4 * 4);
    if (m < M && n0 < N) {
      float x[4] = {v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ACT == GRIDMM_ACT_GELU) x[e] = x[e] * 0.5f * (1.0f + erff(x[e] * 0.70710678118654752440f));
        if (ACT == GRIDMM_ACT_RELU) x[e] = fmaxf(x[e], 0.f);
      }
      if (R) {
        const float4 r4 = *reinterpret_cast<const float4*>(R + (size_t)m * ldr + n0);
        x[0] += r4.x; x[1] += r4.y; x[2] += r4.z; x[3] += r4.w;
      }
      if (C) *reinterpret_cast<float4*>(C + (size_t)m * ldc + n0) = make_float4(x[0], x[1], x[2], x[3]);
      if (Chi) {
        u16x4_t hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned short h = f32_to_bf16_rne(x[e]);
          hi[e] = h;
          lo[e] = f32_to_bf16_rne(x[e] - bf16_bits_to_f32(h));
        }
        *reinterpret_cast<u16x4_t*>(Chi + (size_t)m * ldp + n0) = hi;
        *reinterpret_cast<u16x4_t*>(Clo + (size_t)m * ldp + n0) = lo;
      }
    }
  }
}

// x (M,K) fp32 -> bf16 hi/lo planes (M,ldp), zero padded to ldp
__global__ void split_rows_kernel(const float* __restrict__ X, int ldx, unsigned short* __restrict__ hi,
                                  unsigned short* __restrict__ lo, int ldp, int M, int K) {
  const size_t nv = (size_t)M * (ldp / 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / (ldp / 4)), k = (int)(i % (ldp / 4)) * 4;
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = (k + e < K) ? X[(size_t)m * ldx + k + e] : 0.f;
    u16x4_t h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned short hh = f32_to_bf16_rne(x[e]);
      h[e] = hh;
      l[e] = f32_to_bf16_rne(x[e] - bf16_bits_to_f32(hh));
    }
    *reinterpret_cast<u16x4_t*>(hi + (size_t)m * ldp + k) = h;
    *reinterpret_cast<u16x4_t*>(lo + (size_t)m * ldp + k) = l;
  }
}

template <int BM, int BN>
int launch(const unsigned short* Ahi, const unsigned short* Alo, int lda, const unsigned short* Whi,
           const unsigned short* Wlo, int Kp, const float* bias, const float* R, int ldr, float* C, int ldc,
           unsigned short* Chi, unsigned short* Clo, int ldp, int M, int N, int K, int act, hipStream_t st) {
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM), block(256);
#define GRIDMM_LP(ACT)                                                                                   \
  GRIDMM_LAUNCH((linear_planes_kernel<BM, BN, ACT>), grid, block, 0, st, Ahi, Alo, lda, Whi, Wlo, Kp, bias, \
                R, ldr, C, ldc, Chi, Clo, ldp, M, N, K)
  if (act == GRIDMM_ACT_NONE) GRIDMM_LP(GRIDMM_ACT_NONE);
  else if (act == GRIDMM_ACT_GELU) GRIDMM_LP(GRIDMM_ACT_GELU);
  else GRIDMM_LP(GRIDMM_ACT_RELU);
#undef GRIDMM_LP
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

}  // namespace

extern "C" int gridmm_split_rows(const float* X, int ldx, void* hi, void* lo, int ldp, int M, int K,
                                 gridmm_stream_t stream) {
  if (M <= 0 || K <= 0 || ldp < K || ldp % 8) return GRIDMM_EINVAL;
  const size_t nv = (size_t)M * (ldp / 4);
  unsigned grid = (unsigned)((nv + 255) / 256);
  if (grid > 8192) grid = 8192;
  GRIDMM_LAUNCH(split_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), X, ldx, (unsigned short*)hi,
                (unsigned short*)lo, ldp, M, K);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_linear_planes(const void* A_hi, const void* A_lo, int lda, const void* W_hi,
                                    const void* W_lo, int Kp, const float* bias, const float* residual, int ldr,
                                    float* C, int ldc, void* C_hi, void* C_lo, int ldp, int M, int N, int K,
                                    int act, gridmm_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 || Kp < K || lda % 8 || N % 4 || act < 0 || act > 2)
    return GRIDMM_EINVAL;
  if ((C && ldc % 4) || (residual && ldr % 4) || (C_hi && (ldp % 4 || !C_lo)) || (!C && !C_hi)) return GRIDMM_EINVAL;
  const unsigned short *ah = (const unsigned short*)A_hi, *al = (const unsigned short*)A_lo;
  const unsigned short *wh = (const unsigned short*)W_hi, *wl = (const unsigned short*)W_lo;
  unsigned short *ch = (unsigned short*)C_hi, *cl = (unsigned short*)C_lo;
  hipStream_t st = as_stream(stream);
  const long wg128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (wg128 >= 200)
    return launch<128, 128>(ah, al, lda, wh, wl, Kp, bias, residual, ldr, C, ldc, ch, cl, ldp, M, N, K, act, st);
  return launch<64, 64>(ah, al, lda, wh, wl, Kp, bias, residual, ldr, C, ldc, ch, cl, ldp, M, N, K, act, st);
}
