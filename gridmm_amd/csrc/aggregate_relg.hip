// Relevance pass of the two-pass aggregation for LONG instructions: w_p = max_{l < L} <x_p, text_l> for every point of the
// cell-sorted order, any L <= 256 (the reference pads to max_instr_len = 200, map_nav_src/scripts/run_r2r.sh:38; RxR
// instructions are longer than R2R's 80), D = 256 / 512 / 768.  Reference: map_nav_src/models/vilmodel.py:797-798.
//
// aggregate_pipe.hip keeps the text fragments of a 16-token tile resident in the registers of one wave, which bounds it
// to L <= 96; past that the relevance product is no longer memory-bound anyway (2 N D L flops on f16 hi + lo against
// N D 2 bytes: 400 flop / byte at L = 200, above the ridge), so it is laid out as a GEMM here:
//   S^T [token][point] = T [token][k] . X^T [k][point],   tile = all Lt token tiles x 128 points, k-steps of 32
//   * text operand: the fragment planes gridmm_text_fragments wrote ([hi|lo][token tile][k-step][64 lanes][8]) -- one
//     1-KiB LDS-DMA per (plane, token tile, k-step), read back lane-linearly as MFMA A fragments (no swizzle needed);
//     it is re-streamed from L2 once per 128 points (416 KB at L = 200, D = 512; the episodes of an XCD share it);
//   * point operand: gathered slab rows (perm), 64 B of every row per k-step, LDS image + source-side XOR swizzle as in
//     linear_planes.hip (BK = 32);
//   * 4 MFMA waves x 32 points, accumulators [Lt][2] f32x4; NS-stage ring over the flat (tile, k-step) sequence, one
//     barrier per k-step, counted vmcnt (the MFMA waves issue nothing but DMA and the row-id loads);
//   * epilogue per tile: max (and first arg-max) over tokens in registers + two cross-row steps -> LDS -> a fifth wave
//     stores them (a store in an MFMA wave would make its counted vmcnt waits inexact: reads and writes retire out of
//     order with respect to each other).
// The second pass is grid_aggregate_pipe_kernel<., ., ., PREW> (aggregate_pipe.hip).
#include "agg_accum.h"

namespace {

using namespace gridmm_agg;

constexpr int GPT = 128;      // points per tile
constexpr int GMW = 4;        // MFMA waves (32 points each)
constexpr int GNS = 3;        // ring stages

__device__ __forceinline__ int gswz32(int row) { return ((row >> 3) & 1) << 1; }

__device__ __forceinline__ void gdma(const _Float16* gsrc, _Float16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void wait_vm_upto(int n) {     // s_waitcnt vmcnt(n), n wave-uniform, 0..12
  switch (n) {
    case 12: wait_vm<12>(); break;
    case 11: wait_vm<11>(); break;
    case 10: wait_vm<10>(); break;
    case 9: wait_vm<9>(); break;
    case 8: wait_vm<8>(); break;
    case 7: wait_vm<7>(); break;
    case 6: wait_vm<6>(); break;
    case 5: wait_vm<5>(); break;
    case 4: wait_vm<4>(); break;
    case 3: wait_vm<3>(); break;
    case 2: wait_vm<2>(); break;
    case 1: wait_vm<1>(); break;
    default: wait_vm<0>(); break;
  }
}

template <int KS, int LTMAX>
__global__ __launch_bounds__((GMW + 1) * 64) void grid_relevance_gemm_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const _Float16* __restrict__ text_frag, float* __restrict__ relevance, int32_t* __restrict__ amax, int cap, int L,
    int Lt, int n_chunks) {
  constexpr int D = 32 * KS;
  constexpr int STAGE = 2 * LTMAX * 512 + GPT * 32;      // halfs per ring stage: text blocks | point rows
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* s_ring = reinterpret_cast<_Float16*>(smem);                          // [GNS][STAGE]
  float* s_w = reinterpret_cast<float*>(smem + (size_t)GNS * STAGE * 2);         // [2][GPT] w of the finished tile
  int* s_a = reinterpret_cast<int*>(s_w + 2 * GPT);                              // [2][GPT] its arg-max token
  int* s_ids = s_a + 2 * GPT;                                                    // [2][GPT] slab rows of a tile's points

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
  const int nsteps = ntiles * KS;
  const _Float16* slab_b = slab + (size_t)b * cap * D;
  const int32_t* perm_b = perm + (size_t)b * cap;
  float* rel_b = relevance + (size_t)b * cap;

  if (wave == GMW) {            // ---------------- the storing wave: barriers + the finished tiles' w / arg-max
    __syncthreads();
    for (int s = 0; s <= nsteps; ++s) {
      __builtin_amdgcn_s_barrier();
      if (s > 0 && s % KS == 0) {
        const int t = s / KS - 1, p0 = p_lo + t * GPT;
#pragma unroll
        for (int h = 0; h < GPT / 64; ++h) {
          const int i = h * 64 + lane;
          if (p0 + i < p_hi) {
            rel_b[p0 + i] = s_w[(t & 1) * GPT + i];
            if (amax) amax[(size_t)b * cap + p0 + i] = s_a[(t & 1) * GPT + i];
          }
        }
      }
    }
    return;
  }

  // ---------------- MFMA waves
  const int n_text = 2 * Lt;                               // text pieces per stage: (plane, token tile)
  const int tpw = (n_text + GMW - 1) / GMW;                // per wave (the last ones repeat a valid piece)
  const int ppw = tpw + 2;                                 // + 2 point pieces: DMA instructions per wave and stage
  const size_t plane = (size_t)Lt * KS * 512;
  const _Float16* tf_b = text_frag + (size_t)b * 2 * plane + (size_t)lane * 8;
  const int prow = lane >> 2;                              // row inside a 16-row point piece
  // Row ids travel by LDS-DMA as well (a register load would have to be waited for with a vmcnt the compiler picks --
  // it drained the ring every iteration): every wave fetches 64 of a tile's 128 ids (waves 0 / 2 the first half, 1 / 3
  // the second: duplicates write the same words), ONE instruction per wave and tile, so the queues stay uniform.
  auto load_ids = [&](int t) {
    const int p = min(p_lo + t * GPT + (wave & 1) * 64 + lane, p_hi - 1);     // short tiles repeat the last row
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(perm_b + p),
                                     (__attribute__((address_space(3))) void*)(s_ids + (t & 1) * GPT + (wave & 1) * 64),
                                     4, 0, 0);
  };
  auto issue = [&](int s) {
    const int t = s / KS, ks = s - t * KS;
    _Float16* st = s_ring + (size_t)(s % GNS) * STAGE;
    int id0, id1;
    {   // (asm: a compiler-visible ds_read behind pending LDS-DMA gets an s_waitcnt vmcnt(0) in front)
      const unsigned a = (unsigned)(size_t)(s_ids + (t & 1) * GPT + 16 * wave + prow);
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(id0), "=&v"(id1) : "v"(a), "n"(16 * GMW * 4) : "memory");
    }
    for (int i = 0; i < tpw; ++i) {
      int p = wave + GMW * i;
      if (p >= n_text) p = n_text - 1;                     // padding: the same data to the same place
      const int hl = p >= Lt ? 1 : 0, ct = p - hl * Lt;
      gdma(tf_b + hl * plane + ((size_t)ct * KS + ks) * 512, st + (size_t)p * 512);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = wave + GMW * j, row = 16 * q + prow;
      const int chunk = (lane & 3) ^ gswz32(row);
      gdma(slab_b + (size_t)(j ? id1 : id0) * D + ks * 32 + chunk * 8, st + 2 * LTMAX * 512 + q * 512);
    }
  };

  f32x4_t acc[LTMAX][2];
#pragma unroll
  for (int c = 0; c < LTMAX; ++c) { acc[c][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[c][1] = acc[c][0]; }

  load_ids(0);
  if (ntiles > 1) load_ids(1);
  wait_vm<0>();
  __syncthreads();                           // (the storing wave answers with its first barrier)
  for (int s = 0; s < GNS - 1 && s < nsteps; ++s) issue(s);
  const int frow = lane & 15, g = lane >> 4;
  int ids_young = 0;                         // an id DMA issued after the youngest stage's DMA
  for (int s = 0; s < nsteps; ++s) {
    // stage s landed: everything but the younger stages' DMA (+ a younger id fetch) may still fly (in-order queue)
    const int younger = min(GNS - 2, nsteps - 1 - s);
    wait_vm_upto(younger * ppw + ids_young);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the finished tile's s_w / s_a writes, before the raw barrier
    __builtin_amdgcn_s_barrier();            // stage s visible; the slot of stage s - 1 is free
    const int t = s / KS, ks = s - t * KS;
    if (s + GNS - 1 < nsteps) issue(s + GNS - 1);
    ids_young = 0;
    // ids of tile t + 1 into the slot of tile t - 1 (last read when tile t's first stages were issued, GNS - 1 steps
    // before tile t began); they are needed GNS - 1 steps before tile t + 1 begins: fetched at ks = 1 (KS >= 8)
    if (ks == 1 && t >= 1 && t + 1 < ntiles) { load_ids(t + 1); ids_young = 1; }
    const _Float16* st = s_ring + (size_t)(s % GNS) * STAGE;
    const _Float16* pts = st + 2 * LTMAX * 512;
    f16x8_t x[2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      const int row = wave * 32 + nf * 16 + frow;
      x[nf] = *reinterpret_cast<const f16x8_t*>(pts + row * 32 + ((g ^ gswz32(row)) * 8));
    }
#pragma unroll
    for (int c = 0; c < LTMAX; ++c) {
      if (c < Lt) {
        const f16x8_t ahi = *reinterpret_cast<const f16x8_t*>(st + (size_t)c * 512 + lane * 8);
        const f16x8_t alo = *reinterpret_cast<const f16x8_t*>(st + (size_t)(Lt + c) * 512 + lane * 8);
        acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, x[0], acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, x[1], acc[c][1], 0, 0, 0);
        acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, x[0], acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, x[1], acc[c][1], 0, 0, 0);
      }
    }
    if (ks == KS - 1) {          // tile t finished: lane holds tokens 16 c + 4 g + r of point (wave 32 + nf 16 + frow)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        float best = NEG_BIG;
        int arg = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < LTMAX; ++c) {
          if (c < Lt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int tok = c * 16 + 4 * g + r;
              const float v = tok < L ? acc[c][nf][r] : NEG_BIG;
              if (v > best) { best = v; arg = tok; }             // ascending tokens: the first maximum stays
            }
          }
          acc[c][nf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
          const float y = __shfl_xor(best, m, 64);
          const int j = __shfl_xor(arg, m, 64);
          if (y > best || (y == best && j < arg)) { best = y; arg = j; }
        }
        if (g == 0) {
          s_w[(t & 1) * GPT + wave * 32 + nf * 16 + frow] = best;
          s_a[(t & 1) * GPT + wave * 32 + nf * 16 + frow] = arg;
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();              // the last tile's w for the storing wave
}

}  // namespace

// Returns GRIDMM_EINVAL when the shape is outside this kernel's range.
int gridmm_grid_relevance_gemm(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* relevance, int32_t* amax, int B, int cap, int D, int L, int n_chunks,
                               hipStream_t st) {
  const int Lt = (L + 15) / 16;
  if (!relevance || Lt < 1 || Lt > 16 || (D != 256 && D != 512 && D != 768)) return GRIDMM_EINVAL;
  dim3 grid(n_chunks, B), block((GMW + 1) * 64);
#define GRIDMM_RELG(KS, LTM)                                                                                       \
  do {                                                                                                             \
    auto kern = grid_relevance_gemm_kernel<KS, LTM>;                                                               \
    const size_t lds = (size_t)GNS * (2 * LTM * 512 + GPT * 32) * 2 + 2 * GPT * (sizeof(float) + 2 * sizeof(int));     \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                            (int)lds) != hipSuccess)                                                               \
      return GRIDMM_EINVAL;                                                                                        \
    GRIDMM_LAUNCH(kern, grid, block, lds, st, (const _Float16*)slab, perm, cell_start, (const _Float16*)text_frag, \
                  relevance, amax, cap, L, Lt, n_chunks);                                                          \
  } while (0)
#define GRIDMM_RELG_D(KS)                                  \
  do {                                                     \
    if (Lt <= 8) GRIDMM_RELG(KS, 8);                       \
    else if (Lt <= 13) GRIDMM_RELG(KS, 13);                \
    else GRIDMM_RELG(KS, 16);                              \
  } while (0)
  if (D == 256) GRIDMM_RELG_D(8);
  else if (D == 512) GRIDMM_RELG_D(16);
  else GRIDMM_RELG_D(24);
#undef GRIDMM_RELG_D
#undef GRIDMM_RELG
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
