// Relevance pass of the two-pass aggregation for LONG instructions: w_p = max_{l < L} <x_p, text_l> for every point of the
// cell-sorted order, any L <= 256 (the reference pads to max_instr_len = 200, map_nav_src/scripts/run_r2r.sh:38; RxR
// instructions are longer than R2R's 80), D = 256 / 512 / 768.  Reference: map_nav_src/models/vilmodel.py:797-798.
//
// aggregate_pipe.hip keeps the text fragments of a 16-token tile resident in the registers of one wave, which bounds it
// to L <= 96; past that the relevance product is no longer memory-bound anyway (2 N D L flops on f16 hi + lo against
// N D 2 bytes: 400 flop / byte at L = 200, above the ridge), so it is laid out as a GEMM here:
//   S^T [token][point] = T [token][k] . X^T [k][point],   tile = all Lt token tiles x 256 points, k-steps of 32
//   * text operand: the fragment planes gridmm_text_fragments wrote ([hi|lo][token tile][k-step][64 lanes][8]) -- one
//     1-KiB LDS-DMA per (plane, token tile, k-step), read back lane-linearly as MFMA A fragments (no swizzle needed);
//     it is re-streamed from L2 once per 256 points (416 KB at L = 200, D = 512; the episodes of an XCD share it);
//   * point operand: gathered slab rows (perm), 64 B of every row per k-step, LDS image + source-side XOR swizzle as in
//     linear_planes.hip (BK = 32);
//   * 8 MFMA waves x 32 points (two per SIMD), accumulators [LTMAX][2] f32x4, the text fragments of token tile c + 1 read
//     while tile c multiplies (in-kernel stamps of the first version: 52 cycles per MFMA with the LDS latency exposed);
//     NS-stage ring over the flat (tile, k-step) sequence, one barrier per k-step, counted vmcnt (the MFMA waves issue
//     nothing but DMA);
//   * epilogue per tile: max (and first arg-max) over tokens in registers + two cross-row steps, stored by the wave
//     itself; reads and writes retire out of order with respect to each other, so the step after a store waits with
//     vmcnt(0) instead of a counted value (one drain per tile of KS steps).
// The second pass is grid_aggregate_pipe_kernel<., ., ., PREW> (aggregate_pipe.hip).
#include "agg_accum.h"

namespace {

using namespace gridmm_agg;

#ifdef GRIDMM_RELG_PROF
__device__ long long g_relg_prof[8][8];
#define RG_T() ((long long)__builtin_readcyclecounter())
#define RG(k) { const long long t_ = RG_T(); rgp[k] += t_ - rgt; rgt = t_; }
#else
#define RG(k)
#endif

constexpr int GPT = 256;      // points per tile
constexpr int GMW = 8;        // MFMA waves (32 points each; two per SIMD: one multiplies while the other reads / issues DMA)

__device__ __forceinline__ int gswz32(int row) { return ((row >> 3) & 1) << 1; }

__device__ __forceinline__ void gdma(const _Float16* gsrc, _Float16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void wait_vm_upto(int n) {     // s_waitcnt vmcnt(n), n wave-uniform, 0..23
  switch (n) {
#define GRIDMM_WV(N) case N: wait_vm<N>(); break;
    GRIDMM_WV(23) GRIDMM_WV(22) GRIDMM_WV(21) GRIDMM_WV(20) GRIDMM_WV(19) GRIDMM_WV(18) GRIDMM_WV(17) GRIDMM_WV(16)
    GRIDMM_WV(15) GRIDMM_WV(14) GRIDMM_WV(13) GRIDMM_WV(12) GRIDMM_WV(11) GRIDMM_WV(10) GRIDMM_WV(9) GRIDMM_WV(8)
    GRIDMM_WV(7) GRIDMM_WV(6) GRIDMM_WV(5) GRIDMM_WV(4) GRIDMM_WV(3) GRIDMM_WV(2) GRIDMM_WV(1)
#undef GRIDMM_WV
    default: wait_vm<0>(); break;
  }
}

template <int KS, int LTMAX, int GNS, int BKS>   // BKS: 32-wide k-steps per ring stage (1 or 2)
__global__ __launch_bounds__(GMW * 64) void grid_relevance_gemm_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const _Float16* __restrict__ text_frag, float* __restrict__ relevance, int32_t* __restrict__ amax, int cap, int L,
    int Lt_all, int tile0, int Lt, int n_chunks) {
  // Lt_all: token tiles of the whole instruction (plane stride of text_frag); this launch takes the Lt <= LTMAX tiles from
  // tile0 on.  tile0 > 0 (instructions of more than 256 tokens run as two launches over token groups): the result is
  // combined with what the earlier group stored -- a later token replaces the stored maximum only if strictly larger.
  constexpr int D = 32 * KS;
  constexpr int STAGE = BKS * (2 * LTMAX * 512 + GPT * 32);   // halfs per ring stage: text blocks | point rows
  constexpr int KSTEPS = KS / BKS;                             // ring stages per tile
  static_assert(KS % BKS == 0 && (BKS == 1 || BKS == 2), "one or two k-steps per stage");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* s_ring = reinterpret_cast<_Float16*>(smem);                          // [GNS][STAGE]
  int* s_ids = reinterpret_cast<int*>(smem + (size_t)GNS * STAGE * 2);           // [2][GPT] slab rows of a tile's points

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b, k;
  {   // XCD-aware (episode, chunk) order: the chunks of an episode share one L2 (their text operand lives there)
    const int T = gridDim.x * gridDim.y, lid = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lid & 7, j = lid >> 3, q = T >> 3, rem = T & 7;
    const int f = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
    b = f / n_chunks;
    k = f - b * n_chunks;
  }
  const int n_valid = cell_start[(size_t)b * (GRIDMM_CELLS + 2) + GRIDMM_CELLS];
  const int per = ((n_valid + n_chunks - 1) / n_chunks + GPT - 1) / GPT * GPT;
  const int p_lo = k * per, p_hi = min(n_valid, p_lo + per);
  if (p_lo >= p_hi) return;
  const int ntiles = (p_hi - p_lo + GPT - 1) / GPT;
  const int nsteps = ntiles * KSTEPS;
  const _Float16* slab_b = slab + (size_t)b * cap * D;
  const int32_t* perm_b = perm + (size_t)b * cap;
  float* rel_b = relevance + (size_t)b * cap;

  const int n_text = BKS * 2 * Lt;                         // text pieces per stage: (k-step, plane, token tile)
  const int tpw = (n_text + GMW - 1) / GMW;                // per wave (the last ones repeat a valid piece)
  const int ppw = tpw + 2 * BKS;                           // + the point pieces: DMA instructions per wave and stage
  const size_t plane = (size_t)Lt_all * KS * 512;
  const _Float16* tf_b = text_frag + (size_t)b * 2 * plane + (size_t)tile0 * KS * 512 + (size_t)lane * 8;
  const int prow = BKS == 1 ? lane >> 2 : lane >> 3;       // row inside a 1-KiB point piece (16 rows x 64 B / 8 rows x 128 B)
  // Row ids travel by LDS-DMA as well (a register load would have to be waited for with a vmcnt the compiler picks --
  // it drained the ring every iteration): every wave fetches 64 of a tile's 256 ids (wave w and w + 4 the same quarter:
  // duplicates write the same words), ONE instruction per wave and tile, so the queues stay uniform.
  auto load_ids = [&](int t) {
    const int p = min(p_lo + t * GPT + (wave & 3) * 64 + lane, p_hi - 1);     // short tiles repeat the last row
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(perm_b + p),
                                     (__attribute__((address_space(3))) void*)(s_ids + (t & 1) * GPT + (wave & 3) * 64),
                                     4, 0, 0);
  };
  auto issue = [&](int s) {
    const int t = s / KSTEPS, kk = s - t * KSTEPS;
    _Float16* st = s_ring + (size_t)(s % GNS) * STAGE;
    int id[2 * BKS];
    {   // (asm: a compiler-visible ds_read behind pending LDS-DMA gets an s_waitcnt vmcnt(0) in front)
      const unsigned a = (unsigned)(size_t)(s_ids + (t & 1) * GPT + (BKS == 1 ? 16 : 8) * wave + prow);
      if constexpr (BKS == 1)
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(id[0]), "=&v"(id[1]) : "v"(a), "n"(16 * GMW * 4) : "memory");
      else
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:%5\n\tds_read_b32 %2, %4 offset:%6\n\t"
                     "ds_read_b32 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(id[0]), "=&v"(id[1]), "=&v"(id[2]), "=&v"(id[3])
                     : "v"(a), "n"(8 * GMW * 4), "n"(2 * 8 * GMW * 4), "n"(3 * 8 * GMW * 4) : "memory");
    }
    for (int i = 0; i < tpw; ++i) {
      int p = wave + GMW * i;
      if (p >= n_text) p = n_text - 1;                     // padding: the same data to the same place
      const int sub = p >= 2 * Lt ? 1 : 0, pp = p - sub * 2 * Lt;
      const int hl = pp >= Lt ? 1 : 0, ct = pp - hl * Lt;
      gdma(tf_b + hl * plane + ((size_t)ct * KS + kk * BKS + sub) * 512, st + (size_t)(sub * 2 * LTMAX + pp) * 512);
    }
#pragma unroll
    for (int j = 0; j < 2 * BKS; ++j) {
      const int q = wave + GMW * j;
      if constexpr (BKS == 1) {
        const int row = 16 * q + prow, chunk = (lane & 3) ^ gswz32(row);
        gdma(slab_b + (size_t)id[j] * D + kk * 32 + chunk * 8, st + 2 * LTMAX * 512 + q * 512);
      } else {
        const int row = 8 * q + prow, chunk = (lane & 7) ^ ((row >> 1) & 7);
        gdma(slab_b + (size_t)id[j] * D + kk * 64 + chunk * 8, st + BKS * 2 * LTMAX * 512 + q * 512);
      }
    }
  };

  f32x4_t acc[LTMAX][2];
#pragma unroll
  for (int c = 0; c < LTMAX; ++c) { acc[c][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[c][1] = acc[c][0]; }

  load_ids(0);
  if (ntiles > 1) load_ids(1);
  wait_vm<0>();
  __syncthreads();
  for (int s = 0; s < GNS - 1 && s < nsteps; ++s) issue(s);
  const int frow = lane & 15, g = lane >> 4;
  int ids_young = 0;                         // an id DMA issued after the youngest stage's DMA
  bool stored = false;                       // a global store may be outstanding: counted waits are inexact
#ifdef GRIDMM_RELG_PROF
  long long rgp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rgt = RG_T();
  const long long rg0 = rgt;
#endif
  for (int s = 0; s < nsteps; ++s) {
    // stage s landed: everything but the younger stages' DMA (+ a younger id fetch) may still fly (in-order queue)
    const int younger = min(GNS - 2, nsteps - 1 - s);
    if (stored) wait_vm<0>(); else wait_vm_upto(min(23, younger * ppw + ids_young));
    stored = false;
    RG(0)
    __builtin_amdgcn_s_barrier();            // stage s visible; the slot of stage s - 1 is free
    RG(1)
    const int t = s / KSTEPS, ks = s - t * KSTEPS;
    if (s + GNS - 1 < nsteps) issue(s + GNS - 1);
    RG(2)
    ids_young = 0;
    // ids of tile t + 1 into the slot of tile t - 1 (last read when tile t's first stages were issued, GNS - 1 steps
    // before tile t began); they are needed GNS - 1 steps before tile t + 1 begins: fetched at step 1 (KSTEPS >= 4)
    if (ks == 1 && t >= 1 && t + 1 < ntiles) { load_ids(t + 1); ids_young = 1; }
    const _Float16* stg = s_ring + (size_t)(s % GNS) * STAGE;
    const _Float16* pts = stg + BKS * 2 * LTMAX * 512;
#pragma unroll
    for (int sub = 0; sub < BKS; ++sub) {
      const _Float16* st = stg + (size_t)sub * 2 * LTMAX * 512;
      f16x8_t x[2];
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int row = wave * 32 + nf * 16 + frow;
        if constexpr (BKS == 1) x[nf] = *reinterpret_cast<const f16x8_t*>(pts + row * 32 + ((g ^ gswz32(row)) * 8));
        else x[nf] = *reinterpret_cast<const f16x8_t*>(pts + row * 64 + (((sub * 4 + g) ^ ((row >> 1) & 7)) * 8));
      }
      // token tiles beyond Lt (LTMAX is the instantiation's bound) multiply whatever the stage holds there: their
      // columns are masked in the epilogue (tok >= L), and no branch splits the MFMA stream
      f16x8_t ahi[2], alo[2];
      ahi[0] = *reinterpret_cast<const f16x8_t*>(st + lane * 8);
      alo[0] = *reinterpret_cast<const f16x8_t*>(st + (size_t)Lt * 512 + lane * 8);
#pragma unroll
      for (int c = 0; c < LTMAX; ++c) {
        if (c + 1 < LTMAX) {
          ahi[(c + 1) & 1] = *reinterpret_cast<const f16x8_t*>(st + (size_t)(c + 1) * 512 + lane * 8);
          alo[(c + 1) & 1] = *reinterpret_cast<const f16x8_t*>(st + (size_t)(Lt + c + 1) * 512 + lane * 8);
        }
        acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[c & 1], x[0], acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[c & 1], x[1], acc[c][1], 0, 0, 0);
        acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[c & 1], x[0], acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[c & 1], x[1], acc[c][1], 0, 0, 0);
      }
    }
#ifdef GRIDMM_RELG_PROF
    asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[LTMAX - 1][1]));
    RG(3)
#endif
    if (ks == KSTEPS - 1) {      // tile t finished: lane holds tokens 16 c + 4 g + r of point (wave 32 + nf 16 + frow)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        float best = NEG_BIG;
        int arg = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < LTMAX; ++c) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int tok = (tile0 + c) * 16 + 4 * g + r;
            const float v = tok < L ? acc[c][nf][r] : NEG_BIG;
            if (v > best) { best = v; arg = tok; }               // ascending tokens: the first maximum stays
          }
          acc[c][nf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
          const float y = __shfl_xor(best, m, 64);
          const int j = __shfl_xor(arg, m, 64);
          if (y > best || (y == best && j < arg)) { best = y; arg = j; }
        }
        const int p = p_lo + t * GPT + wave * 32 + nf * 16 + frow;
        if (g == 0 && p < p_hi) {
          bool keep = true;
          if (tile0 > 0) keep = best > rel_b[p];     // (the load drains the DMA queue once per tile: stored = true below)
          if (keep) {
            rel_b[p] = best;
            if (amax) amax[(size_t)b * cap + p] = arg;
          }
        }
      }
      stored = true;
    }
  }
#ifdef GRIDMM_RELG_PROF
  if (blockIdx.x == 1 && blockIdx.y == 3 && lane == 0) {
    long long* o = g_relg_prof[wave];
    o[0] = RG_T() - rg0; o[1] = rgp[0]; o[2] = rgp[1]; o[3] = rgp[2]; o[4] = rgp[3]; o[5] = nsteps;
  }
#endif
}

}  // namespace

#ifdef GRIDMM_RELG_PROF
extern "C" int gridmm_debug_relg_prof(long long* out) {      // development aid (-DGRIDMM_RELG_PROF builds only)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_relg_prof), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

// Returns GRIDMM_EINVAL when the shape is outside this kernel's range.
int gridmm_grid_relevance_gemm(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* relevance, int32_t* amax, int B, int cap, int D, int L, int n_chunks,
                               hipStream_t st) {
  const int Lt_all = (L + 15) / 16;
  if (!relevance || Lt_all < 1 || Lt_all > 32 || (D != 256 && D != 512 && D != 768)) return GRIDMM_EINVAL;
  dim3 grid(n_chunks, B), block(GMW * 64);
  // more than 16 token tiles (L > 256; rxr_pretrain.json: 300): two launches over token groups, the second one combines
  // two k-steps per ring stage where the LDS holds two such stages (Lt <= 10: the per-stage costs -- DMA issue, the wait
  // for the gathered rows, the barrier -- are ~2000 cycles whatever the stage carries), else one k-step, three stages
#define GRIDMM_RELG(KS, LTM, NS, BKS)                                                                              \
  do {                                                                                                             \
    auto kern = grid_relevance_gemm_kernel<KS, LTM, NS, BKS>;                                                      \
    const size_t lds = (size_t)NS * BKS * (2 * LTM * 512 + GPT * 32) * 2 + 2 * GPT * sizeof(int);                  \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                            (int)lds) != hipSuccess)                                                               \
      return GRIDMM_EINVAL;                                                                                        \
    GRIDMM_LAUNCH(kern, grid, block, lds, st, (const _Float16*)slab, perm, cell_start, (const _Float16*)text_frag, \
                  relevance, amax, cap, L, Lt_all, tile0, Lt, n_chunks);                                           \
  } while (0)
#define GRIDMM_RELG_D(KS)                                  \
  do {                                                     \
    if (Lt <= 4) GRIDMM_RELG(KS, 4, 2, 2);                 \
    else if (Lt <= 6) GRIDMM_RELG(KS, 6, 2, 2);            \
    else if (Lt <= 8) GRIDMM_RELG(KS, 8, 2, 2);            \
    else if (Lt <= 10) GRIDMM_RELG(KS, 10, 2, 2);          \
    else if (Lt <= 13) GRIDMM_RELG(KS, 13, 3, 1);          \
    else GRIDMM_RELG(KS, 16, 3, 1);                        \
  } while (0)
  for (int tile0 = 0; tile0 < Lt_all; tile0 += 16) {
    const int Lt = Lt_all - tile0 < 16 ? Lt_all - tile0 : 16;
    if (D == 256) GRIDMM_RELG_D(8);
    else if (D == 512) GRIDMM_RELG_D(16);
    else GRIDMM_RELG_D(24);
  }
#undef GRIDMM_RELG_D
#undef GRIDMM_RELG
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
