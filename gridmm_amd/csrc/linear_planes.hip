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
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

// LDS image of a plane tile: row-major [row][BK] bf16; the 16-B chunk index is XOR-swizzled so that every
// ds_read_b128 lane group (16 rows x one k-chunk) hits 16 distinct bank slots:
//   BK = 32 (64-B rows, 4 chunks):  chunk ^= ((row >> 3) & 1) << 1
//   BK = 64 (128-B rows, 8 chunks): chunk ^= (row >> 1) & 7      (full 128-B lines per DMA row)
template <int BK>
__device__ __forceinline__ int swz(int row) {
  return BK == 32 ? (((row >> 3) & 1) << 1) : ((row >> 1) & 7);
}

__device__ __forceinline__ void dma16(const unsigned short* gsrc, unsigned short* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Per-episode shift of the PLANE output only (training: the K / V projections of the differentiable path hand the attention
// kernels planes of K - K[row 0 of the episode], V - V[row 0] -- softmax(Q K^T) is invariant to a shift of K and
// P V = P (V - v0) + (sum_k P_k) v0 -- so that the bf16 products of the attention carry no large common component; the fp32
// output C stays the true projection).  Columns >= c0 of row m get tab[(m / rpb) * N + n] subtracted before the hi / lo split.
struct PShift { const float* tab; int rpb, c0; int pre_c; };   // pre_c: C receives the PRE-activation (GRIDMM_ACT_GELU_PLANES)

// BM x BN block tile, WM x WN per wave (16x16x32 MFMA tiles), NS-stage LDS ring filled by LDS-DMA.
// One raw s_barrier per k-step; the DMA of stage kt+NS-1 is issued right after the barrier that retires
// stage kt-1, and only a COUNTED s_waitcnt vmcnt keeps the younger stages in flight across barriers.
//
// PP = 1 ("ping-pong", 8 waves, NS = 2, BK = 32): the k-step is cut into 4 barrier-separated intervals
//   R1 (ds_read: all W fragments + A fragments of the upper half of the wave tile) | M1 (its MFMAs)
//   R2 (ds_read: A fragments of the lower half)                                     | M2 (its MFMAs)
// and the upper half of the waves runs ONE interval behind the lower half.  A SIMD hosts waves of both groups, so in every interval
// exactly one of its two waves is inside an MFMA cluster while the other reads LDS and then parks on the barrier:
// the matrix pipe never waits for a ds_read or for a barrier bubble (lockstep waves all read, then all contend).
// The DMA of stage s+1 is issued by every wave at the first interval of step s (the buffer's last readers, the
// lagging group's R2 of step s-1, finished one barrier earlier) and retired (vmcnt(0)) by every wave before the
// barrier that ends the 4th interval -- one barrier before the leading group's first read of it, two before the
// lagging group's -- so a full k-step of MFMA time covers the global->LDS latency.
// WT = 1: the W planes arrive TILED as [N / RPP][Kp / BK][RPP rows][BK k] blocks (used with BK = 32: 16 x 32) -- every 1-KiB DMA
// piece is one contiguous KiB of memory instead of 16 HALF cache lines a row pitch apart (tools/l2_to_lds_bw.hip: contiguous
// pieces stream at 17.8-23.7 TB/s out of the Infinity Cache, strided rows at 12.5).
// RP = 1 ("register-pipelined", BK = 64, round 6): the fragments of sub-step t + 1 (32 of k) are read from LDS into a SECOND
// register set under the MFMAs of sub-step t, so the LDS array works while the matrix pipe does (the plain loop runs them one
// after the other: a k-step costs LDS time + MFMA time once a CU holds a single workgroup).  The barrier of k-step kt sits in
// front of its second sub-step: by then every fragment of stage kt is in registers, so the buffer is FREE a k-step earlier than in
// the plain loop and the ring keeps all NS stages in flight -- the DMA of stage kt + NS is issued right behind that barrier,
// piece by piece between the MFMAs.
template <int BM, int BN, int WM, int WN, int NS, int BK, int ACT, int ABLATE = 0, int PP = 0, int TR = 0, int WT = 0, int RP = 0>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, (BM == 192 && NS == 2 && BK == 32 && WM == 48 && WN == 32) ? 8 : 1) void linear_planes_kernel(
    const unsigned short* __restrict__ Ahi, const unsigned short* __restrict__ Alo, int lda,
    const unsigned short* __restrict__ Whi, const unsigned short* __restrict__ Wlo, int Kp,
    const float* __restrict__ bias, const float* __restrict__ R, int ldr, float* __restrict__ C, int ldc,
    unsigned short* __restrict__ Chi, unsigned short* __restrict__ Clo, int ldp, int M, int N, int K,
    int a_rpb, long a_bs, PShift ps) {
  constexpr int WAVES_N = BN / WN, NW = (BM / WM) * WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int STAGE = (2 * BM + 2 * BN) * BK;          // u16 elements per stage: Ahi|Alo|Whi|Wlo
  constexpr int RPP = 512 / BK;                          // rows per 1-KiB DMA piece
  constexpr int CPR = BK / 8;                            // 16-B chunks per row
  constexpr int PIECES = (2 * BM + 2 * BN) / RPP;        // 1-KiB DMA pieces per stage
  // pieces per wave: an even split takes consecutive pieces; a tile whose piece count is no multiple of the wave count
  // (192 x 128 with 16 waves: 40 pieces) deals them round-robin, the first NFULL waves carry PPW, the rest PPW - 1 -- every
  // wave counts its OWN queue, so the counted vmcnt below is per wave class
  constexpr bool UNEVEN = PIECES % NW != 0;
  constexpr int PPW = (PIECES + NW - 1) / NW;
  constexpr int NFULL = UNEVEN ? PIECES - (PPW - 1) * NW : NW;
  static_assert(!UNEVEN || (!PP && PPW >= 2), "uneven piece split: plain main loop only");
  // rows per epilogue pass (LDS budget); the two-per-CU 192x128 form (80-KB ring) takes 16-row passes so that the epilogue fits the ring
  constexpr int ER = (BM == 192 && NS == 2 && BK == 32 && WM == 48 && WN == 32) ? 16 : (NW > 8 && WM * WN >= 4096) ? 32 : (WM < 64 ? WM : (WM % 64 ? 32 : 64));
  constexpr int EPI = ER * WN;                           // floats per wave in the epilogue transpose
  constexpr int LDS_U16 = (TR || NS * STAGE * 2 > NW * EPI * 4) ? NS * STAGE : NW * EPI * 2;
  __shared__ __attribute__((aligned(16))) unsigned short smem[LDS_U16];
  static_assert(!WT || (!PP && !TR), "tiled planes: the plain main loop");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WAVES_N, wc = wave % WAVES_N;
  // XCD-aware tile order.  Workgroup b runs on XCD b % 8 (each XCD has its own 4 MB L2): give every XCD a
  // CONTIGUOUS chunk of a strip-major tile list (strips of 8 column tiles, row-major inside), so the
  // workgroups resident on one XCD at a time cover a compact ~8x8 block of tiles and share their A / W
  // rows through that L2 instead of re-fetching them from the Infinity Cache (measured on this chip:
  // L2-resident LDS-DMA streams at ~27 TB/s, L2-missing strided rows at ~12 TB/s; tools/l2_to_lds_bw.hip).
  int ty, tx;
  {
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN, T = tm * tn;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, q = T >> 3, rem = T & 7;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;   // bijective for any T
    constexpr int W = 8;
    const int nfs = tn / W, full = nfs * tm * W;
    if (t < full) {
      const int strip = t / (tm * W), r = t - strip * tm * W;
      ty = r / W;
      tx = strip * W + r % W;
    } else {
      const int w = tn - nfs * W, r = t - full;
      ty = r / w;
      tx = nfs * W + r % w;
    }
  }
  if (ABLATE == 4) { ty = 0; tx = 0; }          // every workgroup streams the SAME tile (all L2 hits): the L2 -> LDS path alone
  if (ABLATE == 5) { ty = ty & 7; tx = 0; }     // 8 distinct row tiles, one column tile
  const int bm = ty * BM, bn = tx * BN;

  // DMA plan of this wave: piece p covers 16 rows of one plane
  const unsigned short* src[PPW];
  [[maybe_unused]] const unsigned short* sbase[PPW];   // RP: wave-uniform plane base (SGPRs) + a 32-bit per-lane element offset --
  [[maybe_unused]] unsigned voff[PPW];                 // half the address registers of the 64-bit per-lane pointers
  int dst[PPW];
  [[maybe_unused]] int kstep[PPW];          // elements between consecutive k-steps of a piece (WT: a W block is 512 apart)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    kstep[i] = BK;
    const int p = UNEVEN ? min(i * NW + wave, PIECES - 1) : wave * PPW + i;
    int plane, r0;
    if (p < BM / RPP) { plane = 0; r0 = p * RPP; }
    else if (p < 2 * BM / RPP) { plane = 1; r0 = (p - BM / RPP) * RPP; }
    else if (p < (2 * BM + BN) / RPP) { plane = 2; r0 = (p - 2 * BM / RPP) * RPP; }
    else { plane = 3; r0 = (p - (2 * BM + BN) / RPP) * RPP; }
    const int row = r0 + lane / CPR;
    const int chunk = (lane % CPR) ^ swz<BK>(row);
    if (plane < 2) {
      const int m = min(bm + row, M - 1);
      // batched row map (a_rpb > 0): row m = (episode m / a_rpb, token m % a_rpb) of a [B][.][lda] buffer with batch
      // stride a_bs -- a sequence that lives inside a longer one ([map | txt] contexts) is read in place
      size_t aoff = (size_t)m * lda;
      if (a_rpb > 0) { const int eb = m / a_rpb; aoff = (size_t)eb * a_bs + (size_t)(m - eb * a_rpb) * lda; }
      src[i] = (plane == 0 ? Ahi : Alo) + aoff + chunk * 8;
      sbase[i] = plane == 0 ? Ahi : Alo;
      voff[i] = (unsigned)(aoff + chunk * 8);
      dst[i] = plane * BM * BK + r0 * BK;
    } else {
      const int n = min(bn + row, N - 1);
      if constexpr (WT) {
        src[i] = (plane == 2 ? Whi : Wlo) + (size_t)(n / RPP) * (Kp / BK) * 512 + (n % RPP) * BK + chunk * 8;
        kstep[i] = 512;
      } else {
        src[i] = (plane == 2 ? Whi : Wlo) + (size_t)n * Kp + chunk * 8;
      }
      sbase[i] = plane == 2 ? Whi : Wlo;
      voff[i] = (unsigned)((size_t)n * Kp + chunk * 8);
      dst[i] = 2 * BM * BK + (plane - 2) * BN * BK + r0 * BK;
    }
  }

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // split-K (gridDim.y > 1, TR kernels only): workgroup (tile, blockIdx.y) contracts k-steps [k0, k0 + nk) and writes
  // its partial tile to slice blockIdx.y of a workspace (summed by sum_splits_kernel) -- for the weight-gradient
  // GEMMs, whose output is small (N x K of a Linear) and whose contraction is the whole batch (thousands of rows):
  // without it only N*K/(BM*BN) workgroups exist, each walking hundreds of k-steps.
  int nk = K / BK;
  if (gridDim.y > 1) {
    const int per = (nk + gridDim.y - 1) / gridDim.y, k0 = blockIdx.y * per;
    nk = max(0, min(per, nk - k0));
#pragma unroll
    for (int i = 0; i < PPW; ++i) { src[i] += (size_t)k0 * BK; voff[i] += (unsigned)(k0 * BK); }
  }
#pragma unroll
  for (int s = 0; s < (RP ? NS : NS - 1); ++s)
    if (s < nk) {
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        if (!UNEVEN || i < PPW - 1 || wave < NFULL) {
          if constexpr (RP) dma16(sbase[i] + (voff[i] + (unsigned)(s * BK)), smem + s * STAGE + dst[i]);
          else dma16(src[i] + s * (WT ? kstep[i] : BK), smem + s * STAGE + dst[i]);
        }
    }

  const int frow = lane & 15, fchunk = lane >> 4;
  if constexpr (RP) {
    static_assert(BK == 64 && !PP && !UNEVEN && !ABLATE && !WT, "register-pipelined loop: BK = 64, even piece split, row-major W");
    bf16x8_t ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    // fragment addresses: the swizzle term (row >> 1) & 7 does not depend on the 16-row block (16 i and wr * WM are multiples of
    // 16), so ONE lane offset per (operand, sub-step) + compile-time block offsets (the ds_read offset field) address everything
    static_assert(WM % 16 == 0 && WN % 16 == 0, "");
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ra = wr * WM + frow, rb = wc * WN + frow;
      a_off[ks] = ra * BK + ((ks * 4 + fchunk) ^ swz<BK>(ra)) * 8;
      b_off[ks] = 2 * BM * BK + rb * BK + ((ks * 4 + fchunk) ^ swz<BK>(rb)) * 8;
    }
    auto rd = [&](auto setc, const unsigned short* cur, auto ksc) {
      constexpr int set = decltype(setc)::value, ks = decltype(ksc)::value;
      const unsigned short* pa = cur + a_off[ks];
      const unsigned short* pb = cur + b_off[ks];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[set][i] = *reinterpret_cast<const bf16x8_t*>(pa + i * 16 * BK);
        al[set][i] = *reinterpret_cast<const bf16x8_t*>(pa + BM * BK + i * 16 * BK);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[set][j] = *reinterpret_cast<const bf16x8_t*>(pb + j * 16 * BK);
        bl[set][j] = *reinterpret_cast<const bf16x8_t*>(pb + BN * BK + j * 16 * BK);
      }
    };
    // the MFMAs of one sub-step; `stage` >= 0: the DMA pieces of that stage go out between them (one per accumulator tile)
    auto mma = [&](auto setc, int stage) {
      constexpr int set = decltype(setc)::value;
      unsigned short* nxt = smem + (stage >= 0 ? stage % NS : 0) * STAGE;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (TR) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[set][j], al[set][i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[set][j], ah[set][i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[set][j], ah[set][i], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[set][i], bh[set][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][i], bl[set][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][i], bh[set][j], acc[i][j], 0, 0, 0);
          }
          if (i * TN + j < PPW && stage >= 0)
            dma16(sbase[i * TN + j] + (voff[i * TN + j] + (unsigned)(stage * BK)), nxt + dst[i * TN + j]);
        }
      if constexpr (PPW > TM * TN) {
        if (stage >= 0) {
#pragma unroll
          for (int q = TM * TN; q < PPW; ++q) dma16(sbase[q] + (voff[q] + (unsigned)(stage * BK)), nxt + dst[q]);
        }
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // stage 0 landed (the younger NS - 1 stay in flight) and visible
    if (nk >= NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rd(S0{}, smem, S0{});
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned short* cur = smem + (kt % NS) * STAGE;
      rd(S1{}, cur, S1{});
      mma(S0{}, -1);
      int refill = -1;
      if (kt + 1 < nk) {
        // stage kt + 1 landed: of the stages issued so far (up to kt + NS - 1) the NS - 2 youngest may stay in flight
        if (NS >= 3 && kt + NS - 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's last reads of stage kt are in registers
        __builtin_amdgcn_s_barrier();                        // stage kt + 1 visible to all; buffer kt % NS is free
        rd(S0{}, smem + ((kt + 1) % NS) * STAGE, S0{});
        if (kt + NS < nk) refill = kt + NS;
      }
      mma(S1{}, refill);
    }
  } else
  if constexpr (PP) {
    static_assert(!PP || (NS == 2 && BK == 32 && (NW == 8 || NW == 16) && TM % 2 == 0), "ping-pong schedule: 8 / 16 waves, 2 stages, BK 32");
    constexpr int HM = TM / 2;
    bf16x8_t ah[HM], al[HM], bh[TN], bl[TN];
    auto issue = [&](int stage) {
      unsigned short* nxt = smem + (stage & 1) * STAGE;
#pragma unroll
      for (int i = 0; i < PPW; ++i) dma16(src[i] + stage * BK, nxt + dst[i]);
    };
    auto read_a = [&](const unsigned short* cur, int half) {
#pragma unroll
      for (int i = 0; i < HM; ++i) {
        const int row = wr * WM + (half * HM + i) * 16 + frow;
        const int off = row * BK + (fchunk ^ swz<BK>(row)) * 8;
        ah[i] = *reinterpret_cast<const bf16x8_t*>(cur + off);
        al[i] = *reinterpret_cast<const bf16x8_t*>(cur + BM * BK + off);
      }
    };
    auto read_b = [&](const unsigned short* cur) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wc * WN + j * 16 + frow;
        const int off = 2 * BM * BK + row * BK + (fchunk ^ swz<BK>(row)) * 8;
        bh[j] = *reinterpret_cast<const bf16x8_t*>(cur + off);
        bl[j] = *reinterpret_cast<const bf16x8_t*>(cur + BN * BK + off);
      }
    };
    auto mma_half = [&](int half) {   // 3 passes over the half tile: every accumulator's MFMAs are HM*TN apart
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < HM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[half * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[half * HM + i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < HM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[half * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[half * HM + i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < HM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[half * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[half * HM + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    // interval boundary: nothing may be scheduled across it
#define GRIDMM_IVAL_END()                          \
  do {                                             \
    __builtin_amdgcn_sched_barrier(0);             \
    __builtin_amdgcn_s_barrier();                  \
    __builtin_amdgcn_sched_barrier(0);             \
  } while (0)
#define GRIDMM_READS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define GRIDMM_DMA_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
    GRIDMM_DMA_DONE();            // stage 0 (issued above)
    GRIDMM_IVAL_END();            // ... visible to everyone
    if (wave < NW / 2) {          // leading group: intervals 4s .. 4s+3 of step s
      for (int s = 0; s < nk; ++s) {
        const unsigned short* cur = smem + (s & 1) * STAGE;
        if (s + 1 < nk) issue(s + 1);
        read_b(cur); read_a(cur, 0); GRIDMM_READS_DONE(); GRIDMM_IVAL_END();
        mma_half(0); GRIDMM_IVAL_END();
        read_a(cur, 1); GRIDMM_READS_DONE(); GRIDMM_IVAL_END();
        mma_half(1); GRIDMM_DMA_DONE(); GRIDMM_IVAL_END();
      }
      GRIDMM_IVAL_END();          // the lagging group's last interval
    } else {                      // lagging group: one interval behind
      if (1 < nk) issue(1);
      GRIDMM_IVAL_END();          // interval 0 (idle)
      for (int s = 0; s < nk; ++s) {
        const unsigned short* cur = smem + (s & 1) * STAGE;
        read_b(cur); read_a(cur, 0); GRIDMM_READS_DONE(); GRIDMM_IVAL_END();
        mma_half(0); GRIDMM_IVAL_END();
        read_a(cur, 1); GRIDMM_READS_DONE(); GRIDMM_DMA_DONE(); GRIDMM_IVAL_END();
        if (s + 2 < nk) issue(s + 2);
        mma_half(1); GRIDMM_IVAL_END();
      }
    }
#undef GRIDMM_IVAL_END
#undef GRIDMM_READS_DONE
#undef GRIDMM_DMA_DONE
  } else
  for (int kt = 0; kt < nk; ++kt) {
    // retire stage kt: everything except the (NS-2) younger stages must have landed
    if (NS >= 3 && kt + NS - 2 < nk) {
      if (UNEVEN && wave >= NFULL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (PPW - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // stage kt visible to all waves; buffer (kt-1)%NS is free
    if (ABLATE != 2 && kt + NS - 1 < nk) {
      unsigned short* nxt = smem + ((kt + NS - 1) % NS) * STAGE;
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        if (!UNEVEN || i < PPW - 1 || wave < NFULL) dma16(src[i] + (kt + NS - 1) * (WT ? kstep[i] : BK), nxt + dst[i]);
    }
    const unsigned short* cur = smem + (kt % NS) * STAGE;
    if (ABLATE >= 3) continue;      // the operand stream alone: DMA + waits + barriers
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wr * WM + i * 16 + frow;
        const int off = row * BK + ((ks * 4 + fchunk) ^ swz<BK>(row)) * 8;
        ah[i] = *reinterpret_cast<const bf16x8_t*>(cur + off);
        al[i] = *reinterpret_cast<const bf16x8_t*>(cur + BM * BK + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wc * WN + j * 16 + frow;
        const int off = 2 * BM * BK + row * BK + ((ks * 4 + fchunk) ^ swz<BK>(row)) * 8;
        bh[j] = *reinterpret_cast<const bf16x8_t*>(cur + off);
        bl[j] = *reinterpret_cast<const bf16x8_t*>(cur + BN * BK + off);
      }
      if (ABLATE == 1) {   // keep the fragment reads alive without the matrix work
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (TR) {   // C^T tiles: a lane ends with 4 consecutive COLUMNS of one row (direct row-wise stores)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
        }
    }
  }
  if constexpr (TR) {
    // ---- epilogue straight from the accumulators (no LDS pass, no barrier): with the operands swapped the tile is
    // C^T, i.e. lane (m = lane & 15, g = lane >> 4) holds C[m][4g .. 4g+3] of every 16x16 tile.
    const int mrow = lane & 15, g4 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n0 = bn + wc * WN + j * 16 + g4;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias && n0 < N) bv = *reinterpret_cast<const float4*>(bias + n0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = bm + wr * WM + i * 16 + mrow;
        if (gridDim.y > 1) {            // split-K partial -> its own slice of the workspace (plain stores)
          if (m < M && n0 < N)
            *reinterpret_cast<float4*>(C + (size_t)blockIdx.y * M * ldc + (size_t)m * ldc + n0) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          continue;
        }
        if (m < M && n0 < N) {
          float x[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
          const float4 pre = make_float4(x[0], x[1], x[2], x[3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (ACT == GRIDMM_ACT_GELU) x[e] = x[e] * 0.5f * (1.0f + erff(x[e] * 0.70710678118654752440f));
            if (ACT == GRIDMM_ACT_RELU) x[e] = fmaxf(x[e], 0.f);
            if (ACT == GRIDMM_ACT_QUICKGELU) x[e] = x[e] / (1.0f + __expf(-1.702f * x[e]));
          }
          if (R) {
            const float4 r4 = *reinterpret_cast<const float4*>(R + (size_t)m * ldr + n0);
            x[0] += r4.x; x[1] += r4.y; x[2] += r4.z; x[3] += r4.w;
          }
          if (C) *reinterpret_cast<float4*>(C + (size_t)m * ldc + n0) = (ACT != GRIDMM_ACT_NONE && ps.pre_c) ? pre : make_float4(x[0], x[1], x[2], x[3]);
          if (Chi) {
            if (ps.tab && n0 >= ps.c0) {
              const float4 s4 = *reinterpret_cast<const float4*>(ps.tab + (size_t)(m / ps.rpb) * N + n0);
              x[0] -= s4.x; x[1] -= s4.y; x[2] -= s4.z; x[3] -= s4.w;
            }
            uint2 hi, lo;
            split2_bf16(x[0], x[1], hi.x, lo.x);
            split2_bf16(x[2], x[3], hi.y, lo.y);
            *reinterpret_cast<uint2*>(Chi + (size_t)m * ldp + n0) = hi;
            *reinterpret_cast<uint2*>(Clo + (size_t)m * ldp + n0) = lo;
          }
        }
      }
    }
    return;
  }
  __syncthreads();  // all waves are done with the last stage before LDS is reused by the epilogue

  // ---- epilogue: per-wave transpose through LDS, then row-wise 128-bit accesses
  float* ep = reinterpret_cast<float*>(smem) + wave * EPI;
  constexpr int F4_PER_ROW = WN / 4;                // float4 per sub-tile row
  constexpr int ROWS_PER_IT = 64 / F4_PER_ROW;
  const int c4 = lane % F4_PER_ROW, rr = lane / F4_PER_ROW;
  const int n0 = bn + wc * WN + c4 * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias && n0 < N) bv = *reinterpret_cast<const float4*>(bias + n0);  // N % 4 == 0
#pragma unroll
  for (int h = 0; h < WM / ER; ++h) {
    if (h) __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < ER / 16; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ep[(i * 16 + (lane >> 4) * 4 + r) * WN + j * 16 + (lane & 15)] = acc[h * (ER / 16) + i][j][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < ER / ROWS_PER_IT; ++it) {
      const int row = it * ROWS_PER_IT + rr;
      const int m = bm + wr * WM + h * ER + row;
      float4 v = *reinterpret_cast<const float4*>(ep + row * WN + c4 * 4);
      if (m < M && n0 < N) {
        float x[4] = {v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w};
        const float4 pre = make_float4(x[0], x[1], x[2], x[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ACT == GRIDMM_ACT_GELU) x[e] = x[e] * 0.5f * (1.0f + erff(x[e] * 0.70710678118654752440f));
          if (ACT == GRIDMM_ACT_RELU) x[e] = fmaxf(x[e], 0.f);
          if (ACT == GRIDMM_ACT_QUICKGELU) x[e] = x[e] / (1.0f + __expf(-1.702f * x[e]));
        }
        if (R) {
          const float4 r4 = *reinterpret_cast<const float4*>(R + (size_t)m * ldr + n0);
          x[0] += r4.x; x[1] += r4.y; x[2] += r4.z; x[3] += r4.w;
        }
        if (C) *reinterpret_cast<float4*>(C + (size_t)m * ldc + n0) = (ACT != GRIDMM_ACT_NONE && ps.pre_c) ? pre : make_float4(x[0], x[1], x[2], x[3]);
        if (Chi) {
          if (ps.tab && n0 >= ps.c0) {
            const float4 s4 = *reinterpret_cast<const float4*>(ps.tab + (size_t)(m / ps.rpb) * N + n0);
            x[0] -= s4.x; x[1] -= s4.y; x[2] -= s4.z; x[3] -= s4.w;
          }
          u16x4_t hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned short hh = f32_to_bf16_rne(x[e]);
            hi[e] = hh;
            lo[e] = f32_to_bf16_rne(x[e] - bf16_bits_to_f32(hh));
          }
          *reinterpret_cast<u16x4_t*>(Chi + (size_t)m * ldp + n0) = hi;
          *reinterpret_cast<u16x4_t*>(Clo + (size_t)m * ldp + n0) = lo;
        }
      }
    }
  }
}

// x (M,K) fp32 -> bf16 hi/lo planes (M,ldp), zero padded to ldp
__global__ void split_rows_kernel(const float* __restrict__ X, int ldx, unsigned short* __restrict__ hi,
                                  unsigned short* __restrict__ lo, int ldp, int M, int K, int p_rpb, long p_bs) {
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
    size_t poff = (size_t)m * ldp;
    if (p_rpb > 0) { const int eb = m / p_rpb; poff = (size_t)eb * p_bs + (size_t)(m - eb * p_rpb) * ldp; }
    *reinterpret_cast<u16x4_t*>(hi + poff + k) = h;
    *reinterpret_cast<u16x4_t*>(lo + poff + k) = l;
  }
}

// QG: also instantiate the QuickGELU epilogue (only the configurations pick_cfg can choose carry it)
template <int BM, int BN, int WM, int WN, int NS, int BK, int ABLATE = 0, int PP = 0, int TR = 0, bool QG = false, int WT = 0, int RP = 0>
int launch(const unsigned short* Ahi, const unsigned short* Alo, int lda, const unsigned short* Whi,
           const unsigned short* Wlo, int Kp, const float* bias, const float* R, int ldr, float* C, int ldc,
           unsigned short* Chi, unsigned short* Clo, int ldp, int M, int N, int K, int act, hipStream_t st,
           int ksplit = 1, int a_rpb = 0, long a_bs = 0, PShift ps = PShift{nullptr, 1, 0}) {
  dim3 grid(((N + BN - 1) / BN) * ((M + BM - 1) / BM), ksplit), block((BM / WM) * (BN / WN) * 64);
#define GRIDMM_LP(ACT)                                                                                        \
  GRIDMM_LAUNCH((linear_planes_kernel<BM, BN, WM, WN, NS, BK, ACT, ABLATE, PP, TR, WT, RP>), grid, block, 0, st, Ahi, Alo, lda, Whi, Wlo, \
                Kp, bias, R, ldr, C, ldc, Chi, Clo, ldp, M, N, K, a_rpb, a_bs, ps)
  if (act == GRIDMM_ACT_NONE) GRIDMM_LP(GRIDMM_ACT_NONE);
  else if (act == GRIDMM_ACT_GELU) GRIDMM_LP(GRIDMM_ACT_GELU);
  else if (act == GRIDMM_ACT_RELU) GRIDMM_LP(GRIDMM_ACT_RELU);
  else {
    if constexpr (QG) GRIDMM_LP(GRIDMM_ACT_QUICKGELU);
    else return GRIDMM_EINVAL;
  }
#undef GRIDMM_LP
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

}  // namespace

namespace {
__global__ void sum_splits_kernel(const float* __restrict__ ws, float* __restrict__ out, size_t n4, int splits) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<const float4*>(ws)[i];
    for (int s = 1; s < splits; ++s) {
      const float4 b = reinterpret_cast<const float4*>(ws)[(size_t)s * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}
}  // namespace

// C (fp32, M x N, contiguous) = A W^T with the contraction split over `splits` workgroups per tile; the partial
// tiles go to `workspace` (splits x M x N floats) and are summed in a fixed order (deterministic).
// For GEMMs with a small output and a long contraction: the weight gradients dW = dY^T X of training.
extern "C" int gridmm_linear_planes_splitk(const void* A_hi, const void* A_lo, int lda, const void* W_hi,
                                           const void* W_lo, int Kp, float* C, float* workspace, int M, int N, int K,
                                           int splits, gridmm_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 || Kp < K || lda % 8 || N % 4 || !C || !workspace || splits < 2 || splits > 64)
    return GRIDMM_EINVAL;
  if ((K / 32) < splits) return GRIDMM_EINVAL;   // every split needs at least one k-step
  hipStream_t st = as_stream(stream);
  const unsigned short *ah = (const unsigned short*)A_hi, *al = (const unsigned short*)A_lo;
  const unsigned short *wh = (const unsigned short*)W_hi, *wl = (const unsigned short*)W_lo;
  // 128x128 (16 waves) when the tiles x splits still fill the chip, else 64x64
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * splits;
  int rc;
  if (t128 >= 200)
    rc = launch<128, 128, 32, 32, 2, 32, 0, 0, 1>(ah, al, lda, wh, wl, Kp, nullptr, nullptr, 0, workspace, N, nullptr, nullptr,
                                                  0, M, N, K, 0, st, splits);
  else
    rc = launch<64, 64, 32, 32, 2, 32, 0, 0, 1>(ah, al, lda, wh, wl, Kp, nullptr, nullptr, 0, workspace, N, nullptr, nullptr, 0,
                                                M, N, K, 0, st, splits);
  if (rc != GRIDMM_OK) return rc;
  const size_t n4 = (size_t)M * N / 4;
  GRIDMM_LAUNCH(sum_splits_kernel, dim3((unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256)), dim3(256), 0, st,
                workspace, C, n4, splits);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_split_rows_map(const float* X, int ldx, void* hi, void* lo, int ldp, int p_rpb, int64_t p_bs,
                                     int M, int K, gridmm_stream_t stream) {
  if (M <= 0 || K <= 0 || ldp < K || ldp % 8 || (p_rpb > 0 && p_bs % 8)) return GRIDMM_EINVAL;
  const size_t nv = (size_t)M * (ldp / 4);
  unsigned grid = (unsigned)((nv + 255) / 256);
  if (grid > 8192) grid = 8192;
  GRIDMM_LAUNCH(split_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), X, ldx, (unsigned short*)hi,
                (unsigned short*)lo, ldp, M, K, p_rpb, (long)p_bs);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_split_rows(const float* X, int ldx, void* hi, void* lo, int ldp, int M, int K,
                                 gridmm_stream_t stream) {
  return gridmm_split_rows_map(X, ldx, hi, lo, ldp, 0, 0, M, K, stream);
}

// Tile selection (cfg 0): estimated time = ceil(workgroups / (256 CUs * resident workgroups per CU)) rounds,
// each costing BM*BN*occ / quality, with the relative per-tile throughputs measured on MI355X by
// tools/bench_gemm.py (profiles/gemm_tiles_r1.txt): larger tiles re-use more of each LDS-DMA'd byte, small
// ones fill the 256 CUs when M*N is small.
#ifdef GRIDMM_DEBUG_HOOKS
// Tuning hooks -- ONLY in the development build (make debug -> libgridmm_hip_dbg.so, -DGRIDMM_DEBUG_HOOKS); the shipping
// library has no process-global state.  gridmm_debug_gemm_cfg_override forces the tile configuration of one (M, N, K)
// problem shape inside a running process, so that tools/sweep_gemm_cfg_step.py can time candidate tiles IN the captured
// step (operands arrive as the step leaves them: A just written by the previous launch, weights cold) instead of in an
// isolated loop; gridmm_debug_gemm_shapes lists the problem shapes the heuristic has been asked about.
static int g_cfg_override[32][4];
static int g_cfg_overrides = 0;
extern "C" int gridmm_debug_gemm_cfg_override(int M, int N, int K, int cfg) {
  for (int i = 0; i < g_cfg_overrides; ++i)
    if (g_cfg_override[i][0] == M && g_cfg_override[i][1] == N && g_cfg_override[i][2] == K) {
      g_cfg_override[i][3] = cfg;
      return GRIDMM_OK;
    }
  if (g_cfg_overrides == 32) return GRIDMM_EINVAL;
  int* e = g_cfg_override[g_cfg_overrides++];
  e[0] = M; e[1] = N; e[2] = K; e[3] = cfg;
  return GRIDMM_OK;
}
static int g_shape_log[128][5];
static int g_shape_logged = 0, g_shape_logging = 0;
extern "C" int gridmm_debug_gemm_shapes(int* out, int max_rows, int log) {
  int n = 0;
  if (out)
    for (; n < g_shape_logged && n < max_rows; ++n)
      for (int j = 0; j < 5; ++j) out[n * 5 + j] = g_shape_log[n][j];
  if (log >= 0) { g_shape_logging = log; if (log == 1) g_shape_logged = 0; }
  return n;
}
#endif
static int pick_cfg_impl(int M, int N, int K);
static int pick_cfg(int M, int N, int K) {
#ifdef GRIDMM_DEBUG_HOOKS
  for (int i = 0; i < g_cfg_overrides; ++i)
    if (g_cfg_override[i][0] == M && g_cfg_override[i][1] == N && g_cfg_override[i][2] == K && g_cfg_override[i][3] > 0)
      return g_cfg_override[i][3];
  const int c = pick_cfg_impl(M, N, K);
  if (g_shape_logging) {
    int i = 0;
    for (; i < g_shape_logged; ++i)
      if (g_shape_log[i][0] == M && g_shape_log[i][1] == N && g_shape_log[i][2] == K) break;
    if (i == g_shape_logged && g_shape_logged < 128) { int* e = g_shape_log[g_shape_logged++]; e[0] = M; e[1] = N; e[2] = K; e[3] = c; e[4] = 0; }
    if (i < 128) ++g_shape_log[i][4];
  }
  return c;
#else
  return pick_cfg_impl(M, N, K);
#endif
}

static int pick_cfg_impl(int M, int N, int K) {
  struct Cand { int cfg, bm, bn, occ; float q; bool k64; };
  static const Cand cands[] = {
      {36, 256, 256, 1, 1.35f, false}, // 16 waves, 64x64 per wave (4 waves / SIMD): 3-8 % over the 8-wave 128x64 form
      {16, 256, 128, 1, 1.00f, false}, // 16 waves
      {15, 128, 128, 2, 1.10f, false}, // 16 waves x 2 workgroups = full 32-wave occupancy
      {2, 128, 128, 1, 0.87f, true},   // 8 waves, BK = 64
      {43, 64, 64, 2, 0.78f, true},    // small M*N: fills the 256 CUs; epilogue straight from C^T accumulators
      {4, 64, 64, 5, 0.60f, false}};
  int best = 4;
  float best_t = 1e30f;
  for (const Cand& c : cands) {
    if (c.k64 && (K % 64)) continue;
    const long wgs = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
    const long rounds = (wgs + 256L * c.occ - 1) / (256L * c.occ);
    const float t = (float)rounds * c.occ * c.bm * c.bn / c.q;
    if (t < best_t) { best_t = t; best = c.cfg; }
  }
  // Thin problems whose 128x64 tiles make ONE round over at least half the CUs: 128x64, 8 waves, 3-stage BK = 64 ring.
  // Measured INSIDE the captured step (tools/sweep_gemm_cfg_step.py, profiles/r4_gemm_cfg_sweep_in_step.txt): the
  // 1824x768x768 launches -1.6 us each, 1824x768x3072 -10 us each against the 64x64 tiles the isolated micro-benchmark
  // prefers -- half the tile rows' re-reads of W with the ring deep enough to cover operands that arrive cold.
  if (best == 43 && K % 64 == 0) {
    const long t13 = (long)((M + 127) / 128) * ((N + 63) / 64);
    // round 6: the same tile with the register-pipelined main loop (RP: fragments of sub-step t + 1 read under the MFMAs of t,
    // the whole ring in flight); long contractions also take the direct C^T epilogue (in-step A/B: 1824x768x3072 -20 us per
    // step over 4 launches, 1824x768x768 -4 us over 12; profiles/r6_gemm_cfg_sweep_in_step.txt)
    if (t13 >= 128 && t13 <= 256) best = K >= 1536 ? 75 : 71;
  }
  // Mid-size problems whose 128x128 tiling leaves the second slot of most CUs empty (6912 x 768: 324 tiles on 512 slots): ONE
  // 16-wave workgroup per CU with a 192x128 tile and 80-KB stages of full 128-B lines (BK = 64) moves 17 % fewer operand bytes
  // per flop and keeps one whole stage in flight per CU.  In-step A/B: 6912x768x3072 -43 us per step (2 launches), 6912x768x768
  // -8..-21 us (5 launches).  The operand stream of these launches runs at ~12 TB/s whatever the ring depth
  // (profiles/r6_gemm_ablation.txt: DMA-only 71 us of a 93-us launch), so bytes per flop is the lever.
  if (best == 15 && K % 64 == 0) {
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128), t192 = (long)((M + 191) / 192) * ((N + 127) / 128);
    if (t128 > 256 && t192 >= 160 && t192 <= 256) best = 66;
  }
  // Multi-round problems of the 128x128 class (FFN1 of the grid layers: 1 296 tiles = 2.5 rounds at two per CU): the same
  // 192x128 tile as TWO 16-wave workgroups per CU -- BK 32, 2 stages = 80 KB, 61 VGPRs, 16-row epilogue passes so that the
  // epilogue fits the ring, tiled weight planes -- keeps the epilogue of one workgroup under the main loop of the other (the
  // one-per-CU form loses 25 us per launch here) with 17 % fewer operand bytes per flop: 864 tiles = 1.7 rounds.  In-step A/B:
  // -9 .. -40 us per step over the two launches (four alternations; same-tile noise +-1), isolated 101.5 vs 121 us.
  if (best == 15) {
    const long t192 = (long)((M + 191) / 192) * ((N + 127) / 128);
    if (t192 > 512) best = 76;
  }
  return best;
}

// cfg: 0 = the heuristic above; the shipping library instantiates the tiles the heuristic can choose (2, 4, 13, 15, 16, 36,
// 43, 57), the development build (-DGRIDMM_DEBUG_HOOKS) the whole experiment table of tools/bench_gemm.py and the sweeps.
// w_layout: GRIDMM_W_ROWMAJOR ([N][Kp] planes) or GRIDMM_W_TILED ([ceil(N/16)][Kp/32][16][32] blocks, gridmm_linear_t.wt_hi):
// tiled planes are read by the BK = 32 tiles; a shape whose tile is not one of them answers GRIDMM_EUNSUPPORTED and the
// caller passes the row-major planes.
static int linear_planes_dispatch(const void* A_hi, const void* A_lo, int lda, const void* W_hi,
                                        const void* W_lo, int Kp, int w_layout, const float* bias, const float* residual,
                                        int ldr, float* C, int ldc, void* C_hi, void* C_lo, int ldp, int M, int N,
                                        int K, int act, int cfg, int a_rpb, long a_bs, gridmm_stream_t stream,
                                        PShift ps = PShift{nullptr, 1, 0}) {
  if (ps.tab && (ps.rpb <= 0 || ps.c0 < 0 || ps.c0 % 4 || !C_hi)) return GRIDMM_EINVAL;
  if (w_layout != GRIDMM_W_ROWMAJOR && w_layout != GRIDMM_W_TILED) return GRIDMM_EINVAL;
  const bool wt = w_layout == GRIDMM_W_TILED;
  if (act == GRIDMM_ACT_GELU_PLANES) {          // C = x W^T + b (what the GELU backward needs), planes = those of gelu(C)
    if (!C || !C_hi || residual) return GRIDMM_EINVAL;
    ps.pre_c = 1;
    act = GRIDMM_ACT_GELU;
  }
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 || Kp < K || lda <= 0 || lda % 8 || N % 4 || act < 0 || act > 3)
    return GRIDMM_EINVAL;
  if ((C && ldc % 4) || (residual && ldr % 4) || (C_hi && (ldp % 4 || !C_lo)) || (!C && !C_hi)) return GRIDMM_EINVAL;
  const unsigned short *ah = (const unsigned short*)A_hi, *al = (const unsigned short*)A_lo;
  const unsigned short *wh = (const unsigned short*)W_hi, *wl = (const unsigned short*)W_lo;
  unsigned short *ch = (unsigned short*)C_hi, *cl = (unsigned short*)C_lo;
  hipStream_t st = as_stream(stream);
  if (cfg == 0) cfg = pick_cfg(M, N, K);
#define GRIDMM_ARGS ah, al, lda, wh, wl, Kp, bias, residual, ldr, C, ldc, ch, cl, ldp, M, N, K, act, st, 1, a_rpb, a_bs, ps
  if (wt) {     // tiled planes (16 x 32 blocks): the BK = 32 tiles of the heuristic.  An 8 x 64 copy for the BK = 64 tiles was
                // measured too (their pieces are 8 full 128-B lines already): +-2 us per step, not kept.
    if (cfg == 15) return launch<128, 128, 32, 32, 2, 32, 0, 0, 0, true, 1>(GRIDMM_ARGS);
    if (cfg == 36) return launch<256, 256, 64, 64, 2, 32, 0, 0, 0, true, 1>(GRIDMM_ARGS);
    if (cfg == 76) return launch<192, 128, 48, 32, 2, 32, 0, 0, 0, true, 1>(GRIDMM_ARGS);   // two 16-wave workgroups per CU (80 KB, 61 VGPRs)
#ifdef GRIDMM_DEBUG_HOOKS
    if (cfg == 16) return launch<256, 128, 64, 32, 2, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
    if (cfg == 14) return launch<128, 128, 64, 32, 2, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
    if (cfg == 12) return launch<128, 128, 64, 32, 3, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
    if (cfg == 48) return launch<128, 128, 32, 32, 3, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
    if (cfg == 3) return launch<256, 128, 64, 64, 2, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
    if (cfg == 60) return launch<192, 128, 48, 64, 2, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);   // 192-row tiles (round 5 sweep)
    if (cfg == 61) return launch<192, 128, 96, 32, 2, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
    if (cfg == 62) return launch<256, 128, 64, 64, 3, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
    if (cfg == 63) return launch<128, 128, 64, 64, 2, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
    if (cfg == 64) return launch<192, 256, 96, 64, 2, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
    if (cfg == 65) return launch<192, 128, 48, 32, 3, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);   // round 6: ONE 16-wave workgroup per CU, 80 KB in flight
    if (cfg == 67) return launch<192, 128, 48, 32, 4, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);   //          4-stage ring (160 KB)
    if (cfg == 68) return launch<256, 128, 64, 32, 3, 32, 0, 0, 0, false, 1>(GRIDMM_ARGS);
#endif
    return GRIDMM_EUNSUPPORTED;
  }
  if (K % 64 && (cfg == 2 || cfg == 13 || cfg == 43 || cfg == 57 || cfg == 66 || cfg == 71 || cfg == 75)) return GRIDMM_EINVAL;
  switch (cfg) {
    case 2: return launch<128, 128, 64, 32, 2, 64, 0, 0, 0, true>(GRIDMM_ARGS);
    case 4: return launch<64, 64, 32, 32, 2, 32, 0, 0, 0, true>(GRIDMM_ARGS);
    case 13: return launch<128, 64, 32, 32, 3, 64, 0, 0, 0, true>(GRIDMM_ARGS);
    case 15: return launch<128, 128, 32, 32, 2, 32, 0, 0, 0, true>(GRIDMM_ARGS);
    case 16: return launch<256, 128, 64, 32, 2, 32, 0, 0, 0, true>(GRIDMM_ARGS);
    case 36: return launch<256, 256, 64, 64, 2, 32, 0, 0, 0, true>(GRIDMM_ARGS);   // 16 waves, lockstep
    case 43: return launch<64, 64, 32, 32, 2, 64, 0, 0, 1, true>(GRIDMM_ARGS);     // direct epilogue from C^T accumulators
    case 57: return launch<128, 64, 32, 32, 3, 64, 0, 0, 0, true>(GRIDMM_ARGS);    // = 13
    case 66: return launch<192, 128, 48, 32, 2, 64, 0, 0, 0, true>(GRIDMM_ARGS);   // one 16-wave workgroup per CU, BK = 64
    case 76: return launch<192, 128, 48, 32, 2, 32, 0, 0, 0, true>(GRIDMM_ARGS);   // two 16-wave workgroups per CU, BK = 32
    case 71: return launch<128, 64, 32, 32, 3, 64, 0, 0, 0, true, 0, 1>(GRIDMM_ARGS);   // cfg 13, register-pipelined
    case 75: return launch<128, 64, 32, 32, 3, 64, 0, 0, 1, true, 0, 1>(GRIDMM_ARGS);   // ... with the direct C^T epilogue
    default: break;
  }
#ifdef GRIDMM_DEBUG_HOOKS
  if (K % 64 && (cfg == 66 || (cfg >= 70 && cfg <= 79) || cfg % 100 == 66 || cfg % 100 == 13)) return GRIDMM_EINVAL;
  if (cfg >= 400 && cfg < 600) { /* same-tile ablations write garbage: fine */ }
  if (K % 64 && (cfg == 5 || cfg == 8 || cfg == 9 || cfg == 10 || cfg == 11 || cfg == 108 || cfg == 208 || cfg == 45 || cfg == 49 ||
                 cfg == 50 || cfg == 51 || cfg == 53 || cfg == 54 || cfg == 56))
    return GRIDMM_EINVAL;
  switch (cfg) {   // the experiment table (tools/bench_gemm.py, tools/sweep_gemm_cfg_step.py; profiles/gemm_tiles_r1.txt, r4_gemm_cfg_sweep_*)
    case 1: return launch<128, 128, 64, 64, 2, 32>(GRIDMM_ARGS);
    case 3: return launch<256, 128, 64, 64, 2, 32>(GRIDMM_ARGS);
    case 5: return launch<128, 128, 64, 64, 2, 64>(GRIDMM_ARGS);
    case 6: return launch<128, 64, 64, 32, 2, 32>(GRIDMM_ARGS);
    case 7: return launch<256, 256, 128, 64, 2, 32>(GRIDMM_ARGS);
    case 8: return launch<64, 64, 32, 32, 2, 64>(GRIDMM_ARGS);
    case 9: return launch<128, 64, 32, 32, 2, 64>(GRIDMM_ARGS);
    case 10: return launch<64, 64, 32, 32, 4, 64>(GRIDMM_ARGS);
    case 11: return launch<64, 64, 32, 32, 3, 64>(GRIDMM_ARGS);
    case 12: return launch<128, 128, 64, 32, 3, 32>(GRIDMM_ARGS);
    case 14: return launch<128, 128, 64, 32, 2, 32>(GRIDMM_ARGS);
    case 17: return launch<64, 64, 32, 32, 4, 32>(GRIDMM_ARGS);
    case 18: return launch<64, 64, 32, 32, 3, 32>(GRIDMM_ARGS);
    case 21: return launch<128, 64, 32, 32, 4, 32>(GRIDMM_ARGS);
    case 22: return launch<256, 128, 64, 64, 3, 32>(GRIDMM_ARGS);
    case 23: return launch<256, 128, 64, 32, 3, 32>(GRIDMM_ARGS);
    case 24: return launch<128, 256, 64, 64, 3, 32>(GRIDMM_ARGS);
    case 30: return launch<256, 256, 128, 64, 2, 32, 0, 1>(GRIDMM_ARGS);   // ping-pong schedules
    case 33: return launch<128, 128, 64, 32, 2, 32, 0, 1>(GRIDMM_ARGS);
    case 34: return launch<256, 256, 64, 64, 2, 32, 0, 1>(GRIDMM_ARGS);    // 16 waves: 2 + 2 per SIMD
    case 40: return launch<256, 256, 64, 64, 2, 32, 0, 0, 1>(GRIDMM_ARGS);  // TR = direct epilogue from C^T accumulators
    case 42: return launch<128, 128, 32, 32, 2, 32, 0, 0, 1>(GRIDMM_ARGS);
    case 44: return launch<256, 128, 64, 32, 2, 32, 0, 0, 1>(GRIDMM_ARGS);
    case 45: return launch<64, 64, 32, 32, 3, 64, 0, 0, 1, true>(GRIDMM_ARGS);    // deeper rings (cold operands in the step)
    case 48: return launch<128, 128, 32, 32, 3, 32, 0, 0, 0, true>(GRIDMM_ARGS);
    case 50: return launch<128, 64, 32, 32, 3, 64, 0, 0, 1>(GRIDMM_ARGS);          // 128x64 tiles, direct epilogue (thin M, in-step sweep)
    case 51: return launch<128, 64, 32, 32, 2, 64, 0, 0, 1>(GRIDMM_ARGS);
    case 52: return launch<128, 64, 32, 32, 4, 32, 0, 0, 1>(GRIDMM_ARGS);
    case 53: return launch<64, 128, 32, 32, 3, 64, 0, 0, 1>(GRIDMM_ARGS);
    case 54: return launch<128, 64, 64, 32, 3, 64, 0, 0, 1>(GRIDMM_ARGS);
    case 55: return launch<128, 64, 32, 32, 3, 32, 0, 0, 1>(GRIDMM_ARGS);
    case 56: return launch<96, 64, 48, 32, 3, 64>(GRIDMM_ARGS);
    case 65: return launch<192, 128, 48, 32, 3, 32>(GRIDMM_ARGS);
    case 67: return launch<192, 128, 48, 32, 4, 32>(GRIDMM_ARGS);
    case 68: return launch<256, 128, 64, 32, 3, 32>(GRIDMM_ARGS);
    case 69: return launch<96, 64, 48, 32, 3, 64, 0, 0, 1>(GRIDMM_ARGS);
    // register-pipelined main loops (RP = 1)
    case 70: return launch<192, 128, 48, 32, 2, 64, 0, 0, 0, false, 0, 1>(GRIDMM_ARGS);
    case 72: return launch<128, 128, 32, 32, 2, 64, 0, 0, 0, false, 0, 1>(GRIDMM_ARGS);    // 16 waves, one workgroup per CU
   // 96 x 64 thin tiles (228 workgroups at M = 1824), direct epilogue
    // ablations (tools/bench_gemm.py): 1xx = no MFMA (DMA + LDS reads only), 2xx = no DMA after the prologue
    case 108: return launch<64, 64, 32, 32, 2, 64, 1>(GRIDMM_ARGS);
    case 208: return launch<64, 64, 32, 32, 2, 64, 2>(GRIDMM_ARGS);
    case 466: return launch<192, 128, 48, 32, 2, 64, 4>(GRIDMM_ARGS);
    case 566: return launch<192, 128, 48, 32, 2, 64, 5>(GRIDMM_ARGS);
    case 467: return launch<192, 128, 48, 32, 4, 32, 4>(GRIDMM_ARGS);
    case 365: return launch<192, 128, 48, 32, 3, 32, 3>(GRIDMM_ARGS);
    case 367: return launch<192, 128, 48, 32, 4, 32, 3>(GRIDMM_ARGS);
    case 165: return launch<192, 128, 48, 32, 3, 32, 1>(GRIDMM_ARGS);
    case 167: return launch<192, 128, 48, 32, 4, 32, 1>(GRIDMM_ARGS);
    case 265: return launch<192, 128, 48, 32, 3, 32, 2>(GRIDMM_ARGS);
    case 166: return launch<192, 128, 48, 32, 2, 64, 1>(GRIDMM_ARGS);
    case 266: return launch<192, 128, 48, 32, 2, 64, 2>(GRIDMM_ARGS);
    case 366: return launch<192, 128, 48, 32, 2, 64, 3>(GRIDMM_ARGS);
    case 315: return launch<128, 128, 32, 32, 2, 32, 3>(GRIDMM_ARGS);
    case 336: return launch<256, 256, 64, 64, 2, 32, 3>(GRIDMM_ARGS);
    case 136: return launch<256, 256, 64, 64, 2, 32, 1>(GRIDMM_ARGS);
    case 236: return launch<256, 256, 64, 64, 2, 32, 2>(GRIDMM_ARGS);
    case 313: return launch<128, 64, 32, 32, 3, 64, 3>(GRIDMM_ARGS);
    case 113: return launch<128, 64, 32, 32, 3, 64, 1>(GRIDMM_ARGS);
    case 213: return launch<128, 64, 32, 32, 3, 64, 2>(GRIDMM_ARGS);
    case 115: return launch<128, 128, 32, 32, 2, 32, 1>(GRIDMM_ARGS);
    case 215: return launch<128, 128, 32, 32, 2, 32, 2>(GRIDMM_ARGS);
    case 107: return launch<256, 256, 128, 64, 2, 32, 1>(GRIDMM_ARGS);
    case 207: return launch<256, 256, 128, 64, 2, 32, 2>(GRIDMM_ARGS);
    default: break;
  }
#endif
  return GRIDMM_EINVAL;
#undef GRIDMM_ARGS
}

extern "C" int gridmm_linear_planes_cfg(const void* A_hi, const void* A_lo, int lda, const void* W_hi,
                                        const void* W_lo, int Kp, const float* bias, const float* residual,
                                        int ldr, float* C, int ldc, void* C_hi, void* C_lo, int ldp, int M, int N,
                                        int K, int act, int cfg, gridmm_stream_t stream) {
  return linear_planes_dispatch(A_hi, A_lo, lda, W_hi, W_lo, Kp, GRIDMM_W_ROWMAJOR, bias, residual, ldr, C, ldc, C_hi, C_lo, ldp,
                                M, N, K, act, cfg, 0, 0, stream);
}

// The general form: A rows through a batched row map -- row m of the GEMM = row (m % a_rpb) of episode (m / a_rpb) in a
// buffer whose episodes are a_bs elements apart (a_rpb <= 0: plain rows); lets a GEMM read a sub-sequence of a longer padded
// sequence in place (the instruction rows inside the local encoder's [map | txt] context, the map nodes inside
// [cells | nodes]) -- and the W planes in either layout (w_layout).
extern "C" int gridmm_linear_planes_map(const void* A_hi, const void* A_lo, int lda, int a_rpb, int64_t a_bs,
                                        const void* W_hi, const void* W_lo, int Kp, int w_layout, const float* bias,
                                        const float* residual, int ldr, float* C, int ldc, void* C_hi, void* C_lo,
                                        int ldp, int M, int N, int K, int act, gridmm_stream_t stream) {
  if (a_rpb > 0 && (a_bs % 8)) return GRIDMM_EINVAL;
  return linear_planes_dispatch(A_hi, A_lo, lda, W_hi, W_lo, Kp, w_layout, bias, residual, ldr, C, ldc, C_hi, C_lo, ldp, M, N, K,
                                act, 0, a_rpb, (long)a_bs, stream);
}

// gridmm_linear_planes whose PLANE output is shifted per episode (PShift above): planes[m][n] = split(x[m][n] -
// shift[(m / shift_rpb) * N + n]) for the columns n >= shift_c0, C = x as always.  The K / V projections of the differentiable
// path (the attention kernels on the bf16 matrix pipe read the shifted planes, gridmm_attention_rows_train).
extern "C" int gridmm_linear_planes_shift(const void* A_hi, const void* A_lo, int lda, const void* W_hi, const void* W_lo, int Kp,
                                          const float* bias, const float* residual, int ldr, float* C, int ldc, void* C_hi,
                                          void* C_lo, int ldp, const float* shift, int shift_rpb, int shift_c0, int M, int N, int K,
                                          int act, gridmm_stream_t stream) {
  return linear_planes_dispatch(A_hi, A_lo, lda, W_hi, W_lo, Kp, GRIDMM_W_ROWMAJOR, bias, residual, ldr, C, ldc, C_hi, C_lo, ldp, M,
                                N, K, act, 0, 0, 0, stream, PShift{shift, shift_rpb, shift_c0});
}

#ifdef GRIDMM_DEBUG_HOOKS
// (development build: the general form with an explicit tile configuration -- tools/bench_gemm_tiled.py)
extern "C" int gridmm_debug_linear_planes_map_cfg(const void* A_hi, const void* A_lo, int lda, int a_rpb, int64_t a_bs,
                                                  const void* W_hi, const void* W_lo, int Kp, int w_layout, const float* bias,
                                                  const float* residual, int ldr, float* C, int ldc, void* C_hi, void* C_lo,
                                                  int ldp, int M, int N, int K, int act, int cfg, gridmm_stream_t stream) {
  return linear_planes_dispatch(A_hi, A_lo, lda, W_hi, W_lo, Kp, w_layout, bias, residual, ldr, C, ldc, C_hi, C_lo, ldp, M, N, K,
                                act, cfg, a_rpb, (long)a_bs, stream);
}
#endif

extern "C" int gridmm_linear_planes(const void* A_hi, const void* A_lo, int lda, const void* W_hi,
                                    const void* W_lo, int Kp, const float* bias, const float* residual, int ldr,
                                    float* C, int ldc, void* C_hi, void* C_lo, int ldp, int M, int N, int K,
                                    int act, gridmm_stream_t stream) {
  return gridmm_linear_planes_cfg(A_hi, A_lo, lda, W_hi, W_lo, Kp, bias, residual, ldr, C, ldc, C_hi, C_lo, ldp, M,
                                  N, K, act, 0, stream);
}
