// gridmm_linear_planes_grouped: several SMALL plane GEMMs (C = act(A W^T + b)) in ONE launch.
//
// The tail of forward('navigation') (map_nav_src/models/vilmodel.py:859-877) is four ClsPrediction heads over 32..1824
// rows each: as separate launches they cost ~16-25 us apiece for a few us of matrix work (launch, first-operand latency,
// epilogue drain), and the fuse head additionally needed a torch.cat + split to build its (B, 2H) input.  Here every
// head is one entry of a problem table; workgroup -> (problem, 64x64 tile) by a prefix sum over the table.  Same tile
// pipeline as linear_planes_kernel<64, 64, 32, 32, 2, 64, ., 0, 0, 1> (LDS-DMA ring, bf16 hi/lo 3-term MFMA, direct
// epilogue from C^T accumulators); A rows go through the batched row map of gridmm_linear_planes_map, so a head reads
// "row 0 of every episode" or "the map-node rows of [cells | nodes]" in place.
#include "common.h"

namespace {

constexpr int GBM = 64, GBN = 64, GBK = 64, GNS = 2, GNW = 4;
constexpr int GSTAGE = (2 * GBM + 2 * GBN) * GBK;   // u16 per stage
constexpr int GPPW = ((2 * GBM + 2 * GBN) / 8) / GNW;   // 1-KiB DMA pieces (8 rows x 128 B) per wave and stage

struct GroupedArgs {
  gridmm_gemm_problem_t p[GRIDMM_MAX_GROUPED];
  int tile0[GRIDMM_MAX_GROUPED + 1];
  int n;
};

__device__ __forceinline__ int gswz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void gdma16(const unsigned short* gsrc, unsigned short* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(256) void linear_planes_grouped_kernel(const GroupedArgs args) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[GNS * GSTAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GRIDMM_MAX_GROUPED; ++i)
    if (i < args.n && (int)blockIdx.x >= args.tile0[i]) pi = i;
  const gridmm_gemm_problem_t& P = args.p[pi];
  const int M = P.M, N = P.N, K = P.K, lda = P.lda, Kp = P.Kp;
  const int t = blockIdx.x - args.tile0[pi];
  const int tn = (N + GBN - 1) / GBN;
  const int bm = (t / tn) * GBM, bn = (t % tn) * GBN;
  const unsigned short *Ahi = (const unsigned short*)P.A_hi, *Alo = (const unsigned short*)P.A_lo;
  const unsigned short *Whi = (const unsigned short*)P.W_hi, *Wlo = (const unsigned short*)P.W_lo;

  const unsigned short* src[GPPW];
  int dst[GPPW];
#pragma unroll
  for (int i = 0; i < GPPW; ++i) {
    const int p = wave * GPPW + i;          // 32 pieces: 8 per plane (64 rows / 8 rows per piece)
    const int plane = p >> 3, r0 = (p & 7) * 8;
    const int row = r0 + lane / 8;
    const int chunk = (lane % 8) ^ gswz(row);
    if (plane < 2) {
      const int m = min(bm + row, M - 1);
      size_t aoff = (size_t)m * lda;
      if (P.a_rpb > 0) { const int eb = m / P.a_rpb; aoff = (size_t)eb * P.a_bs + (size_t)(m - eb * P.a_rpb) * lda; }
      src[i] = (plane == 0 ? Ahi : Alo) + aoff + chunk * 8;
      dst[i] = plane * GBM * GBK + r0 * GBK;
    } else {
      const int n = min(bn + row, N - 1);
      src[i] = (plane == 2 ? Whi : Wlo) + (size_t)n * Kp + chunk * 8;
      dst[i] = 2 * GBM * GBK + (plane - 2) * GBN * GBK + r0 * GBK;
    }
  }
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nk = K / GBK;
#pragma unroll
  for (int i = 0; i < GPPW; ++i) gdma16(src[i], smem + dst[i]);
  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) {
      unsigned short* nxt = smem + ((kt + 1) & 1) * GSTAGE;
#pragma unroll
      for (int i = 0; i < GPPW; ++i) gdma16(src[i] + (kt + 1) * GBK, nxt + dst[i]);
    }
    const unsigned short* cur = smem + (kt & 1) * GSTAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wr * 32 + i * 16 + frow;
        const int off = row * GBK + ((ks * 4 + fchunk) ^ gswz(row)) * 8;
        ah[i] = *reinterpret_cast<const bf16x8_t*>(cur + off);
        al[i] = *reinterpret_cast<const bf16x8_t*>(cur + GBM * GBK + off);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wc * 32 + j * 16 + frow;
        const int off = 2 * GBM * GBK + row * GBK + ((ks * 4 + fchunk) ^ gswz(row)) * 8;
        bh[j] = *reinterpret_cast<const bf16x8_t*>(cur + off);
        bl[j] = *reinterpret_cast<const bf16x8_t*>(cur + GBN * GBK + off);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // operands swapped: the tile is C^T, a lane ends with 4 consecutive columns of one row
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    }
  }
  const int mrow = lane & 15, g4 = (lane >> 4) * 4;
  const int act = P.act;
  float* C = P.C;
  unsigned short *Chi = (unsigned short*)P.C_hi, *Clo = (unsigned short*)P.C_lo;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n0 = bn + wc * 32 + j * 16 + g4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.bias && n0 < N) bv = *reinterpret_cast<const float4*>(P.bias + n0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = bm + wr * 32 + i * 16 + mrow;
      if (m < M && n0 < N) {
        float x[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (act == GRIDMM_ACT_GELU) x[e] = x[e] * 0.5f * (1.0f + erff(x[e] * 0.70710678118654752440f));
          else if (act == GRIDMM_ACT_RELU) x[e] = fmaxf(x[e], 0.f);
        }
        if (C) *reinterpret_cast<float4*>(C + (size_t)m * P.ldc + n0) = make_float4(x[0], x[1], x[2], x[3]);
        if (Chi) {
          uint2 hi, lo;
          split2_bf16(x[0], x[1], hi.x, lo.x);
          split2_bf16(x[2], x[3], hi.y, lo.y);
          *reinterpret_cast<uint2*>(Chi + (size_t)m * P.ldp + n0) = hi;
          *reinterpret_cast<uint2*>(Clo + (size_t)m * P.ldp + n0) = lo;
        }
      }
    }
  }
}

}  // namespace

extern "C" int gridmm_linear_planes_grouped(const gridmm_gemm_problem_t* problems, int n_problems,
                                            gridmm_stream_t stream) {
  if (!problems || n_problems <= 0 || n_problems > GRIDMM_MAX_GROUPED) return GRIDMM_EINVAL;
  GroupedArgs a;
  a.n = n_problems;
  int tiles = 0;
  for (int i = 0; i < n_problems; ++i) {
    const gridmm_gemm_problem_t& p = problems[i];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.K % 64 || p.Kp < p.K || p.lda % 8 || p.N % 4 || !p.A_hi || !p.A_lo ||
        !p.W_hi || !p.W_lo || (!p.C && !p.C_hi) || (p.C && p.ldc % 4) || (p.C_hi && (!p.C_lo || p.ldp % 4)) ||
        (p.a_rpb > 0 && p.a_bs % 8) || p.act < 0 || p.act > GRIDMM_ACT_RELU)
      return GRIDMM_EINVAL;
    a.p[i] = p;
    a.tile0[i] = tiles;
    tiles += ((p.M + GBM - 1) / GBM) * ((p.N + GBN - 1) / GBN);
  }
  for (int i = n_problems; i <= GRIDMM_MAX_GROUPED; ++i) a.tile0[i] = tiles;
  GRIDMM_LAUNCH(linear_planes_grouped_kernel, dim3(tiles), dim3(256), 0, as_stream(stream), a);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
