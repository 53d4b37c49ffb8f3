// gridmm_linear_planes_tn: C (N x K, fp32) = A^T B with BOTH operands row-major over the contraction,
//   A = dY planes [M][>= N],  B = X planes [M][>= K]   (bf16 hi / lo; any M: the last 32-row step re-reads row M-1 for the
//   missing rows and zeroes A's copies of them in LDS),
// i.e. the weight gradient dW = dY^T X of a Linear (pretrain_src/train_r2r.py:262 / map_nav_src/r2r/agent_base.py:199
// run it as torch's autograd of nn.Linear) WITHOUT transposed copies of dY and X: the tile pipeline stages row-major
// [32 contraction rows][64 columns] panels in LDS by LDS-DMA and reads the MFMA fragments through the hardware transpose
// read (ds_read_b64_tr_b16), exactly as attention_rows reads V^T out of its row-major V image:
//   * panel image: 32 rows x 128 B, 16-byte slot `s` of row r holds chunk s ^ (r & 6) (swizzle applied on the DMA's
//     SOURCE side; the destination is lane-linear);
//   * lane (j, g) of a fragment read gets column 16 nb + j of rows 4g .. 4g+3 (first read) and 16+4g .. 16+4g+3 (second):
//     contraction slot 8g + e <-> row 4g + e (e < 4) / 16 + 4g + (e - 4) -- the SAME permutation for both operands, so
//     the products pair up row by row;
//   * 3-term bf16 split as in linear_planes (lo*hi + hi*lo + hi*hi, fp32 accumulate), operands swapped so that a lane
//     ends with 4 consecutive output columns (direct 128-bit stores, no LDS epilogue).
// Split-K over blockIdx.y: partial tiles to a workspace, summed in a fixed order (deterministic).
#include "common.h"

namespace {

__device__ __forceinline__ void dma16(const unsigned short* gsrc, unsigned short* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ uint2 lds_tr_b64(unsigned addr) {       // no wait: see tr_fence
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint2 lds_tr_b64_2k(unsigned addr) {    // + 16 rows (2048 B)
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void tr_fence(uint2 (&a)[N]) {          // s_waitcnt the uses of a[] cannot be moved above
  static_assert(N == 8, "eight pairs per fence");
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               :
               : "memory");
}

constexpr int PANEL = 32 * 64;   // u16 per panel image (32 rows x 64 columns)

// One output tile (`tile`) of one contraction range (`split` of `nsplits`) of one problem: the body of the plain kernel
// (tile = blockIdx.x, split = blockIdx.y) and of the grouped one (several weight gradients per launch).
template <int BM, int BN, int WM, int WN, int NS>
__device__ __forceinline__ void linear_planes_tn_tile(
    const unsigned short* __restrict__ Ahi, const unsigned short* __restrict__ Alo, int lda,
    const unsigned short* __restrict__ Bhi, const unsigned short* __restrict__ Blo, int ldb, float* __restrict__ C, int ldc,
    int M, int N, int K, const int tile, const int split, const int nsplits, float* __restrict__ dbw = nullptr) {
  constexpr int WAVES_N = BN / WN, NW = (BM / WM) * WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int PA = BM / 64, PB = BN / 64;                 // panels per plane
  constexpr int STAGE = 2 * (PA + PB) * PANEL;              // u16 per stage: A hi | A lo | B hi | B lo
  constexpr int PIECES = 2 * (PA + PB) * 4;                 // 1-KiB pieces (8 rows x 128 B) per stage
  static_assert(BM % 64 == 0 && BN % 64 == 0 && PIECES % NW == 0 && TM == 2 && TN == 2, "tile shape");
  constexpr int PPW = PIECES / NW;
  __shared__ __attribute__((aligned(16))) unsigned short smem[NS * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WAVES_N, wc = wave % WAVES_N;
  const int tn = (K + BN - 1) / BN;
  const int ty = tile / tn, tx = tile % tn;
  const int bm = ty * BM, bn = tx * BN;                     // first output row (column of A) / column (column of B)

  // DMA plan of this wave: piece p = (plane, panel, row group of 8)
  const int lrow = lane >> 3;
  const unsigned short* src[PPW];    // row (rg * 8 + lrow) of k-step 0
  const unsigned short* col0[PPW];   // row 0 of the same column (the last, partial k-step clamps its rows to M - 1)
  int dst[PPW], prow[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave * PPW + i;
    const int rg = p & 3, pp = p >> 2;                      // pp: A hi panels, A lo panels, B hi panels, B lo panels
    const bool isA = pp < 2 * PA;
    const int q = isA ? pp : pp - 2 * PA, P = isA ? PA : PB;
    const int lo = q / P, panel = q % P;
    const int ld = isA ? lda : ldb;
    int col = (isA ? bm : bn) + panel * 64 + (((lane & 7) ^ (lrow & 6)) << 3);
    col = min(col, ld - 8);                                 // tiles past the matrix edge re-read its last chunk (never stored)
    const unsigned short* base = isA ? (lo ? Alo : Ahi) : (lo ? Blo : Bhi);
    prow[i] = rg * 8 + lrow;
    col0[i] = base + col;
    src[i] = base + (size_t)prow[i] * ld + col;
    dst[i] = pp * PANEL + rg * 8 * 64;
  }
  const size_t stepA = (size_t)32 * lda, stepB = (size_t)32 * ldb;
  const int firstB = (2 * PA * 4 - wave * PPW);             // pieces [firstB, PPW) of this wave belong to B
  const int nk_all = (M + 31) >> 5, rem = M & 31;           // rem > 0: the last k-step holds only `rem` rows

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // dbw != NULL: also the column sums of A over this contraction range (the bias gradient of the Linear whose weight gradient
  // this is: db[n] = sum_m dY[m][n]) -- by the waves of the FIRST tile column that own distinct A columns, as two more MFMAs per
  // A tile and k-step with an all-ones first operand: every row of that product is sum_m A[m][c]; from the hi and lo planes
  // (their sum is dY to 2^-17 relative), fp32 accumulate.  One partial row per contraction range, summed by the summing pass.
  const bool do_db = dbw != nullptr && tx == 0 && wc == 0;
  f32x4_t accd[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) accd[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));
  int nk = nk_all, k0 = 0;
  if (nsplits > 1) {
    const int per = (nk + nsplits - 1) / nsplits;
    k0 = split * per;
    nk = max(0, min(per, nk - k0));
  }
  auto issue = [&](int kt, int slot) {
    if (rem && k0 + kt == nk_all - 1) {                       // partial step: rows past M - 1 re-read row M - 1 (finite data;
#pragma unroll                                                // A's copies are zeroed in LDS before the fragments are read)
      for (int i = 0; i < PPW; ++i) {
        const int r = min((k0 + kt) * 32 + prow[i], M - 1);
        dma16(col0[i] + (size_t)r * (i < firstB ? lda : ldb), smem + slot * STAGE + dst[i]);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      dma16(src[i] + (size_t)(k0 + kt) * (i < firstB ? stepA : stepB), smem + slot * STAGE + dst[i]);
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(s, s);

  // fragment addresses (bytes inside a panel): lane (j, g): row 4g + (j >> 2), 16-column block nb, 4 columns (j & 3) * 4 ..
  const int j = lane & 15, g = lane >> 4;
  const int frow = 4 * g + (j >> 2), s2 = (frow >> 1) & 3;
  unsigned aoff[TM], boff[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int c = wr * WM + i * 16, panel = c >> 6, nb = (c & 63) >> 4;
    aoff[i] = (unsigned)(panel * PANEL * 2 + frow * 128 + ((nb ^ s2) << 5) + ((j & 3) << 3));
  }
#pragma unroll
  for (int jj = 0; jj < TN; ++jj) {
    const int c = wc * WN + jj * 16, panel = c >> 6, nb = (c & 63) >> 4;
    boff[jj] = (unsigned)((2 * PA + panel) * PANEL * 2 + frow * 128 + ((nb ^ s2) << 5) + ((j & 3) << 3));
  }
  const unsigned lds0 = (unsigned)(size_t)smem;

  for (int kt = 0; kt < nk; ++kt) {
    if (NS >= 3 && kt + NS - 2 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // stage kt visible to all waves; buffer (kt-1) % NS is free
    if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
    if (rem && k0 + kt == nk_all - 1) {                       // zero A's rows [rem, 32) of this stage (both planes, all panels)
      unsigned short* a0 = smem + (kt % NS) * STAGE;
      const int chunks = 2 * PA * (32 - rem) * 8;             // 16-byte chunks
      for (int c = tid; c < chunks; c += NW * 64) {
        const int panel = c / ((32 - rem) * 8), rc = c % ((32 - rem) * 8);
        *reinterpret_cast<uint4*>(a0 + panel * PANEL + (rem + rc / 8) * 64 + (rc % 8) * 8) = make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
    }
    const unsigned cur = lds0 + (unsigned)((kt % NS) * STAGE * 2);
    uint2 ra[8], rb[8];             // [tile][hi/lo][row half]
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ra[i * 4 + 0] = lds_tr_b64(cur + aoff[i]);
      ra[i * 4 + 1] = lds_tr_b64_2k(cur + aoff[i]);
      ra[i * 4 + 2] = lds_tr_b64(cur + aoff[i] + PA * PANEL * 2);
      ra[i * 4 + 3] = lds_tr_b64_2k(cur + aoff[i] + PA * PANEL * 2);
    }
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
      rb[jj * 4 + 0] = lds_tr_b64(cur + boff[jj]);
      rb[jj * 4 + 1] = lds_tr_b64_2k(cur + boff[jj]);
      rb[jj * 4 + 2] = lds_tr_b64(cur + boff[jj] + PB * PANEL * 2);
      rb[jj * 4 + 3] = lds_tr_b64_2k(cur + boff[jj] + PB * PANEL * 2);
    }
    tr_fence(ra);
    tr_fence(rb);
    bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = __builtin_bit_cast(bf16x8_t, make_uint4(ra[i * 4].x, ra[i * 4].y, ra[i * 4 + 1].x, ra[i * 4 + 1].y));
      al[i] = __builtin_bit_cast(bf16x8_t, make_uint4(ra[i * 4 + 2].x, ra[i * 4 + 2].y, ra[i * 4 + 3].x, ra[i * 4 + 3].y));
    }
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
      bh[jj] = __builtin_bit_cast(bf16x8_t, make_uint4(rb[jj * 4].x, rb[jj * 4].y, rb[jj * 4 + 1].x, rb[jj * 4 + 1].y));
      bl[jj] = __builtin_bit_cast(bf16x8_t, make_uint4(rb[jj * 4 + 2].x, rb[jj * 4 + 2].y, rb[jj * 4 + 3].x, rb[jj * 4 + 3].y));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj) {   // C^T tiles: a lane ends with 4 consecutive COLUMNS (B columns) of one row (A column)
        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[jj], al[i], acc[i][jj], 0, 0, 0);
        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[jj], ah[i], acc[i][jj], 0, 0, 0);
        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[jj], ah[i], acc[i][jj], 0, 0, 0);
      }
    if (do_db) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        accd[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, al[i], accd[i], 0, 0, 0);
        accd[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ah[i], accd[i], 0, 0, 0);
      }
    }
  }
  if (do_db && (lane >> 4) == 0) {            // lane c of the first 16 holds (row 0 of) the column sum of A column 16 i + c
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = bm + wr * WM + i * 16 + (lane & 15);
      if (m < N) dbw[(size_t)split * N + m] = accd[i][0];
    }
  }
  // ---- epilogue straight from the accumulators: lane (m = lane & 15, g) holds C[row m][cols 4g .. 4g+3] of every tile
  const int mrow = lane & 15, g4 = (lane >> 4) * 4;
  float* out = C + (nsplits > 1 ? (size_t)split * N * ldc : 0);
#pragma unroll
  for (int jj = 0; jj < TN; ++jj) {
    const int n0 = bn + wc * WN + jj * 16 + g4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = bm + wr * WM + i * 16 + mrow;
      if (m < N && n0 < K)
        *reinterpret_cast<float4*>(out + (size_t)m * ldc + n0) =
            make_float4(acc[i][jj][0], acc[i][jj][1], acc[i][jj][2], acc[i][jj][3]);
    }
  }
}

template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void linear_planes_tn_kernel(
    const unsigned short* __restrict__ Ahi, const unsigned short* __restrict__ Alo, int lda,
    const unsigned short* __restrict__ Bhi, const unsigned short* __restrict__ Blo, int ldb, float* __restrict__ C, int ldc,
    int M, int N, int K, float* __restrict__ dbw) {
  linear_planes_tn_tile<BM, BN, WM, WN, NS>(Ahi, Alo, lda, Bhi, Blo, ldb, C, ldc, M, N, K, (int)blockIdx.x, (int)blockIdx.y,
                                            (int)gridDim.y, dbw);
}

// Several weight gradients in ONE launch (the six of a cross-modal layer's backward, the four of a BertLayer's): problem p
// owns the workgroups [first[p], first[p + 1]) = its tiles x its contraction ranges.  Same tile body, same tile / range
// numbering per problem as the plain launch: bit-identical results.
constexpr int TN_GROUP_MAX = 8;
struct TnProb {
  const unsigned short *Ahi, *Alo, *Bhi, *Blo;
  float* out;                    // C, or the split-K workspace
  float* dbw;                    // bias-gradient partials (nsplits x N), or NULL
  int lda, ldb, M, N, K, nsplits, tiles;
};
struct TnGroup {
  TnProb p[TN_GROUP_MAX];
  int first[TN_GROUP_MAX + 1];
  int n;
};
template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void linear_planes_tn_grouped_kernel(const TnGroup g) {
  int q = 0;
  while (q + 1 < g.n && (int)blockIdx.x >= g.first[q + 1]) ++q;
  const TnProb& pr = g.p[q];
  const int local = (int)blockIdx.x - g.first[q];
  linear_planes_tn_tile<BM, BN, WM, WN, NS>(pr.Ahi, pr.Alo, pr.lda, pr.Bhi, pr.Blo, pr.ldb, pr.out, pr.K, pr.M, pr.N, pr.K,
                                            local % pr.tiles, local / pr.tiles, pr.nsplits, pr.dbw);
}

// Second stages of a group (see sum_splits_tn_kernel): problem p owns the blocks [first[p], first[p + 1]).
struct SumProb {
  const float* ws; float* out; size_t n4; const float* colpart; float* db;
  int splits, sum_blocks, n_part, C;
};
struct SumGroup {
  SumProb p[TN_GROUP_MAX];
  int first[TN_GROUP_MAX + 1];
  int n;
};
__global__ void sum_splits_tn_grouped_kernel(const SumGroup g) {
  int q = 0;
  while (q + 1 < g.n && (int)blockIdx.x >= g.first[q + 1]) ++q;
  const SumProb& pr = g.p[q];
  const int blk = (int)blockIdx.x - g.first[q];
  if (blk >= pr.sum_blocks) {
    const int c = (blk - pr.sum_blocks) * blockDim.x + threadIdx.x;
    if (c < pr.C) {
      float s = 0.f;
      for (int r = 0; r < pr.n_part; ++r) s += pr.colpart[(size_t)r * pr.C + c];
      pr.db[c] = s;
    }
    return;
  }
  for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < pr.n4; i += (size_t)pr.sum_blocks * blockDim.x) {
    float4 a = reinterpret_cast<const float4*>(pr.ws)[i];
    for (int s = 1; s < pr.splits; ++s) {
      const float4 b = reinterpret_cast<const float4*>(pr.ws)[(size_t)s * pr.n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(pr.out)[i] = a;
  }
}

// Second stage of a weight gradient: out = the `splits` partial products summed in order; the workgroups past `sum_blocks`
// reduce, likewise in order, the bias-gradient partials that the GEMM wrote (one row of column sums of A per contraction range,
// see linear_planes_tn_tile): db[c] = sum_r colpart[r][c] -- the bias gradient without a launch of its own.
__global__ void sum_splits_tn_kernel(const float* __restrict__ ws, float* __restrict__ out, size_t n4, int splits,
                                     int sum_blocks, const float* __restrict__ colpart, float* __restrict__ db, int n_part,
                                     int C) {
  if ((int)blockIdx.x >= sum_blocks) {
    const int c = ((int)blockIdx.x - sum_blocks) * blockDim.x + threadIdx.x;
    if (c < C) {
      float s = 0.f;
      for (int r = 0; r < n_part; ++r) s += colpart[(size_t)r * C + c];
      db[c] = s;
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)sum_blocks * blockDim.x) {
    float4 a = reinterpret_cast<const float4*>(ws)[i];
    for (int s = 1; s < splits; ++s) {
      const float4 b = reinterpret_cast<const float4*>(ws)[(size_t)s * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

template <int BM, int BN, int NS>
int launch_tn(const unsigned short* ah, const unsigned short* al, int lda, const unsigned short* bh, const unsigned short* bl,
              int ldb, float* C, int M, int N, int K, int splits, hipStream_t st, float* dbw = nullptr) {
  dim3 grid(((N + BM - 1) / BM) * ((K + BN - 1) / BN), splits), block((BM / 32) * (BN / 32) * 64);
  GRIDMM_LAUNCH((linear_planes_tn_kernel<BM, BN, 32, 32, NS>), grid, block, 0, st, ah, al, lda, bh, bl, ldb, C, K, M, N, K, dbw);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

}  // namespace

// How many contraction ranges to cut an (N x K) <- M-row weight gradient into: enough 128x128 workgroups to fill the chip
// (~384: two per CU with slack for the tail), at least 256 rows per range, at most 8 (the partial tiles are fp32 N x K each).
extern "C" int gridmm_linear_planes_tn_splits(int M, int N, int K) {
  const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128);
  long s = (384 + tiles / 2) / tiles;
  if (s > 8) s = 8;
  if (s > M / 256) s = M / 256;
  return s < 1 ? 1 : (int)s;
}

// C (N x K fp32, contiguous) = A^T B over the M rows of A (M x >= N) and B (M x >= K); splits > 1: the contraction is
// cut into `splits` ranges whose partial results go to `workspace` (splits x N x K floats) and are summed in order.
// db != NULL: also db (N floats) = the column sums of A (the bias gradient of the Linear whose weight gradient this is),
// computed by the GEMM itself from the planes (one partial row per range in db_ws, splits x N floats) and summed, in range
// order, by the summing pass.
static int linear_planes_tn_impl(const void* A_hi, const void* A_lo, int lda, const void* B_hi, const void* B_lo, int ldb,
                                 float* C, float* workspace, int M, int N, int K, int splits, float* db_ws, float* db,
                                 gridmm_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 4 || lda % 8 || ldb % 8 || lda < 8 || ldb < 8 || !C || splits < 1 ||
      splits > 64 || (splits > 1 && (!workspace || (M + 31) / 32 < splits)) || (db && !db_ws))
    return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  const unsigned short *ah = (const unsigned short*)A_hi, *al = (const unsigned short*)A_lo;
  const unsigned short *bh = (const unsigned short*)B_hi, *bl = (const unsigned short*)B_lo;
  float* out = splits > 1 ? workspace : C;
  float* dbw = db ? db_ws : nullptr;
  const long t128 = (long)((N + 127) / 128) * ((K + 127) / 128) * splits;
  int rc;
  if (t128 >= 100 && N >= 128 && K >= 128) rc = launch_tn<128, 128, 2>(ah, al, lda, bh, bl, ldb, out, M, N, K, splits, st, dbw);
  else rc = launch_tn<64, 64, 3>(ah, al, lda, bh, bl, ldb, out, M, N, K, splits, st, dbw);
  if (rc != GRIDMM_OK) return rc;
  if (splits > 1 || db) {
    const size_t n4 = (size_t)N * K / 4;
    const int sum_blocks = splits > 1 ? (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256) : 0;
    const int db_blocks = db ? (N + 255) / 256 : 0;
    GRIDMM_LAUNCH(sum_splits_tn_kernel, dim3((unsigned)(sum_blocks + db_blocks)), dim3(256), 0, st, workspace, C, n4, splits,
                  sum_blocks, dbw, db, splits, N);
    GRIDMM_CHECK_LAUNCH();
  }
  return GRIDMM_OK;
}

extern "C" int gridmm_linear_planes_tn(const void* A_hi, const void* A_lo, int lda, const void* B_hi, const void* B_lo,
                                       int ldb, float* C, float* workspace, int M, int N, int K, int splits,
                                       gridmm_stream_t stream) {
  return linear_planes_tn_impl(A_hi, A_lo, lda, B_hi, B_lo, ldb, C, workspace, M, N, K, splits, nullptr, nullptr, stream);
}

extern "C" int gridmm_linear_planes_tn_db(const void* A_hi, const void* A_lo, int lda, const void* B_hi, const void* B_lo,
                                          int ldb, float* C, float* workspace, int M, int N, int K, int splits,
                                          float* db_ws, float* db, gridmm_stream_t stream) {
  return linear_planes_tn_impl(A_hi, A_lo, lda, B_hi, B_lo, ldb, C, workspace, M, N, K, splits, db_ws, db, stream);
}

// n <= 8 weight gradients (+ their bias gradients) as at most two GEMM launches (one per tile class) and one summing launch:
// exactly the results of n calls of gridmm_linear_planes_tn_db, problem by problem.
extern "C" int gridmm_linear_planes_tn_grouped(const gridmm_tn_problem_t* probs, int n, gridmm_stream_t stream) {
  if (!probs || n < 1 || n > TN_GROUP_MAX) return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  TnGroup big, small;
  SumGroup sums;
  big.n = small.n = sums.n = 0;
  big.first[0] = small.first[0] = sums.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    const gridmm_tn_problem_t& q = probs[i];
    if (q.M <= 0 || q.N <= 0 || q.K <= 0 || q.K % 4 || q.lda % 8 || q.ldb % 8 || q.lda < 8 || q.ldb < 8 || !q.C || q.splits < 1 ||
        q.splits > 64 || (q.splits > 1 && (!q.workspace || (q.M + 31) / 32 < q.splits)) || (q.db && !q.db_ws))
      return GRIDMM_EINVAL;
    const long t128 = (long)((q.N + 127) / 128) * ((q.K + 127) / 128) * q.splits;
    const bool use128 = t128 >= 100 && q.N >= 128 && q.K >= 128;              // the plain entry point's rule
    const int BT = use128 ? 128 : 64;
    TnGroup& g = use128 ? big : small;
    TnProb& t = g.p[g.n];
    t.Ahi = (const unsigned short*)q.A_hi; t.Alo = (const unsigned short*)q.A_lo;
    t.Bhi = (const unsigned short*)q.B_hi; t.Blo = (const unsigned short*)q.B_lo;
    t.out = q.splits > 1 ? q.workspace : q.C;
    t.dbw = q.db ? q.db_ws : nullptr;
    t.lda = q.lda; t.ldb = q.ldb; t.M = q.M; t.N = q.N; t.K = q.K; t.nsplits = q.splits;
    t.tiles = ((q.N + BT - 1) / BT) * ((q.K + BT - 1) / BT);
    g.first[g.n + 1] = g.first[g.n] + t.tiles * q.splits;
    ++g.n;
    if (q.splits > 1 || q.db) {
      SumProb& sp = sums.p[sums.n];
      sp.n4 = (size_t)q.N * q.K / 4;
      sp.ws = q.workspace; sp.out = q.C; sp.splits = q.splits;
      sp.sum_blocks = q.splits > 1 ? (int)((sp.n4 + 255) / 256 > 2048 ? 2048 : (sp.n4 + 255) / 256) : 0;
      sp.colpart = q.db ? q.db_ws : nullptr; sp.db = q.db; sp.n_part = q.splits; sp.C = q.N;
      sums.first[sums.n + 1] = sums.first[sums.n] + sp.sum_blocks + (q.db ? (q.N + 255) / 256 : 0);
      ++sums.n;
    }
  }
  if (big.n) {
    GRIDMM_LAUNCH((linear_planes_tn_grouped_kernel<128, 128, 32, 32, 2>), dim3((unsigned)big.first[big.n]), dim3(16 * 64), 0, st, big);
    GRIDMM_CHECK_LAUNCH();
  }
  if (small.n) {
    GRIDMM_LAUNCH((linear_planes_tn_grouped_kernel<64, 64, 32, 32, 3>), dim3((unsigned)small.first[small.n]), dim3(4 * 64), 0, st, small);
    GRIDMM_CHECK_LAUNCH();
  }
  if (sums.n) {
    GRIDMM_LAUNCH(sum_splits_tn_grouped_kernel, dim3((unsigned)sums.first[sums.n]), dim3(256), 0, st, sums);
    GRIDMM_CHECK_LAUNCH();
  }
  return GRIDMM_OK;
}
