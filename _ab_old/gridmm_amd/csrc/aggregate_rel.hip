// First pass of the D = 768 aggregation (the reference's own feature width: 12 views x 49 CLIP ViT-B/32 patch tokens
// of 768 dims per observation), L <= 80: relevance w_p = max_l <x_p, text_l> of every point, by sorted position.  The
// second pass (grid_aggregate_pipe_kernel<24, 3, 10, PREW>) turns w into the per-cell softmax sums.
//
// aggregate_pipe.hip keeps the text fragments of one 16-token tile resident in ONE wave: at D = 768 that is 192 VGPRs
// and does not fit beside the MFMA working set.  Here the 240 fragment units (5 token tiles x 24 k-steps x hi/lo) are
// spread over all 8 waves, 30 units = 120 VGPRs each: a wave holds a K-slice of one or two token tiles and the relevance
// of a tile is a sum of per-wave partial products in s_part[point][token]: a token tile has up to three contributing
// waves, which store / add in turn (two barriers; ds_add_f32 atomics measured ~1000 cycles per instruction).  (Doing
// the accumulation in the same kernel needs another ~100 VGPRs per wave: measured 91 spills -- hence two passes; the
// slab of a typical episode is still in the Infinity Cache for the second one.)
//   phase R    every wave: its K-slice of S = T . X^T on the matrix pipe (text fragment = A operand) -> s_part
//   barrier B  partial sums complete, tile i + 1 landed;  wave 7: max over tokens (lane = point, the lane halves split the tokens) -> w;
//              waves 0..3: feed the ring (tile i + 2 by LDS-DMA, 8 rows each; no global stores in these waves, so
//              their counted vmcnt waits are exact)
// Ring: 3 x 48 KB.  Row ids travel by LDS-DMA two iterations ahead (see aggregate_pipe.hip).
#include "agg_accum.h"

namespace {
#ifdef GRIDMM_AGG_PROF
__device__ long long g_relprof[8][8];
#define RP_T() ((long long)__builtin_readcyclecounter())
#define RP(k) { const long long t_ = RP_T(); rp[k] += t_ - rpt; rpt = t_; }
#else
#define RP(k)
#endif

using namespace gridmm_agg;

// LDS accesses of the ring-feeding waves as asm: a compiler-visible LDS store behind an LDS-DMA gets an s_waitcnt
// vmcnt(0) in front (the DMA destination might alias), which would drain the ring every iteration.
__device__ __forceinline__ void lds_store4(const float* p, const f32x4_t& v) {
  asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)p), "v"(v) : "memory");
}
// Cross-row reductions (max over the four 16-lane rows of a wave, lane & 15 stays) use ds_bpermute through asm: a
// compiler-visible one is an "LDS load" for the waitcnt pass (see above).
// (value, index) variant: the larger value wins, ties go to the smaller index (torch.max returns the first maximum)
__device__ __forceinline__ void argmax_over_rows(float& x, int& idx) {
  const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
  for (unsigned m = 16; m <= 32; m <<= 1) {
    float y;
    int j;
    asm volatile("ds_bpermute_b32 %0, %2, %3\n\tds_bpermute_b32 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(y), "=&v"(j) : "v"((lane ^ m) << 2), "v"(x), "v"(idx) : "memory");
    if (y > x || (y == x && j < idx)) { x = y; idx = j; }
  }
}
__device__ __forceinline__ void lds_load4x2(const float* p0, const float* p1, f32x4_t& o0, f32x4_t& o1) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(o0), "=&v"(o1) : "v"((unsigned)(size_t)p0), "v"((unsigned)(size_t)p1) : "memory");
}
// two values at once: both permutes of a step are in flight together
__device__ __forceinline__ void max_over_rows2(float& x0, float& x1) {
  const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
  for (unsigned m = 16; m <= 32; m <<= 1) {
    float y0, y1;
    asm volatile("ds_bpermute_b32 %0, %2, %3\n\tds_bpermute_b32 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(y0), "=&v"(y1) : "v"((lane ^ m) << 2), "v"(x0), "v"(x1) : "memory");
    x0 = fmaxf(x0, y0);
    x1 = fmaxf(x1, y1);
  }
}

constexpr int KS = 24, D = 32 * KS;           // 768
constexpr int NCH = D / 8, IPR = 2;           // 16-B chunks per row; DMA instructions per row
constexpr int R = 3;                          // ring slots
constexpr int PPW = 15;                       // (token tile, k-step) pairs per wave, hi + lo fragment each
constexpr int PP = 84;                        // s_part pitch (floats per point): 80 tokens + pad, rows 16-B aligned
constexpr int NDMA = 4;                       // ring-feeding waves (0 .. NDMA - 1)
constexpr int RW = PT / NDMA;                 // rows fetched per DMA wave and tile

__global__ __launch_bounds__(512) void grid_relevance_wide_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const _Float16* __restrict__ text_frag, float* __restrict__ relevance, int32_t* __restrict__ amax, int cap, int L,
    int Lt, int n_chunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* s_tiles = reinterpret_cast<_Float16*>(smem);                        // [R][PT][D]
  float* s_part = reinterpret_cast<float*>(smem + (size_t)R * PT * D * 2);      // [PT][PP] relevance partial sums
  float* s_wmax = s_part + PT * PP;                                             // [PT][8] per-token-tile maxima of a point
  int* s_warg = reinterpret_cast<int*>(s_wmax + PT * 8);                        // [PT][8] arg-max token per token tile
  int* s_ids = s_warg + PT * 8;                        // [NDMA waves][4 tiles][RW] slab rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, k = blockIdx.x;
  // points (sorted positions) [p_lo, p_hi) of the episode's valid points: any even cut will do here
  const int n_valid = cell_start[(size_t)b * (GRIDMM_CELLS + 2) + GRIDMM_CELLS];
  const int per = ((n_valid + n_chunks - 1) / n_chunks + PT - 1) / PT * PT;
  const int p_lo = k * per, p_hi = min(n_valid, p_lo + per);
  if (p_lo >= p_hi) return;
  const int ntiles = (p_hi - p_lo + PT - 1) / PT;
  for (int i = tid; i < PT * PP; i += 512) s_part[i] = 0.f;
  for (int i = tid; i < PT * 8; i += 512) s_wmax[i] = NEG_BIG;          // columns of absent token tiles stay at -inf

  // ---- this wave's text fragments: pairs 15 wave .. 15 wave + 14 of the (token tile, k-step) list
  const size_t plane = (size_t)Lt * KS * 64 * 8;
  const _Float16* tf_b = text_frag + (size_t)b * 2 * plane + (size_t)lane * 8;
  const int gp0 = PPW * wave;
  const int ct_a = gp0 / KS;                                 // token tile of the first pair
  const int n_a = min(PPW, KS * (ct_a + 1) - gp0);          // pairs of token tile ct_a; the rest belong to ct_a + 1
  f16x8_t thi[PPW], tlo[PPW];
#pragma unroll
  for (int p = 0; p < PPW; ++p) {
    const int gp = gp0 + p, ct = gp / KS, ks = gp % KS;
    if (ct < Lt) {
      thi[p] = *reinterpret_cast<const f16x8_t*>(tf_b + ((size_t)ct * KS + ks) * 64 * 8);
      tlo[p] = *reinterpret_cast<const f16x8_t*>(tf_b + plane + ((size_t)ct * KS + ks) * 64 * 8);
    } else {
      thi[p] = tlo[p] = (f16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  const bool seg_a = ct_a < Lt, seg_b = n_a < PPW && ct_a + 1 < Lt;

  const _Float16* slab_b = slab + (size_t)b * cap * D;
  const int32_t* perm_b = perm + (size_t)b * cap;
  const bool is_dma = wave < NDMA;

  // ---- ring feed (waves 0 .. NDMA - 1): rows wave, wave + NDMA, ... of a tile
  auto load_ids = [&](int t) {                               // -> s_ids[wave][t & 3][j]
    if (lane < RW) {
      int p = p_lo + t * PT + wave + NDMA * lane;
      if (p >= p_hi) p = p_hi - 1;                           // short tiles repeat the last valid row
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(perm_b + p),
                                       (__attribute__((address_space(3))) void*)(s_ids + (wave * 4 + (t & 3)) * RW),
                                       4, 0, 0);
    }
  };
  int ids_s[RW];                                             // slab rows of the tile being fetched (wave-uniform)
  auto dma_prepare = [&](int t) {
    static_assert(RW == 8, "ids are read as two int4");
    int4 idv[2];
    {
      const unsigned a = (unsigned)(size_t)(s_ids + (wave * 4 + (t & 3)) * RW);           // uniform address: broadcast
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(idv[0]), "=&v"(idv[1]) : "v"(a) : "memory");
    }
    const int idl[RW] = {idv[0].x, idv[0].y, idv[0].z, idv[0].w, idv[1].x, idv[1].y, idv[1].z, idv[1].w};
#pragma unroll
    for (int j = 0; j < RW; ++j) ids_s[j] = __builtin_amdgcn_readfirstlane(idl[j]);
  };
  auto dma_rows = [&](int t, int j0, int j1) {               // position c of row r holds global chunk c ^ (r & 15)
    _Float16* dst = s_tiles + (size_t)(t % R) * PT * D;
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      int r = wave + NDMA * j;
      asm volatile("" : "+s"(r));                            // opaque: keeps hoisted lane-offset VGPRs from spilling
      const _Float16* row = slab_b + (size_t)ids_s[j] * D;
#pragma unroll
      for (int c0 = 0; c0 < NCH; c0 += 64) {
        const int c = c0 + lane;
        if (c < NCH)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(row + (size_t)(c ^ (r & 15)) * 8),
              (__attribute__((address_space(3))) void*)(dst + (size_t)r * D + (size_t)c0 * 8), 16, 0, 0);
      }
    }
  };

  __builtin_amdgcn_s_waitcnt(0);                // text fragments: retire ordinary loads before the loop
  if (is_dma) {
    for (int t = 0; t < 4 && t < ntiles; ++t) load_ids(t);
    wait_vm<0>();
  }
  __syncthreads();                              // s_part cleared
  if (is_dma) {
    for (int t = 0; t < R - 1 && t < ntiles; ++t) { dma_prepare(t); dma_rows(t, 0, RW); }
    if (ntiles > 1) wait_vm<RW * IPR>(); else wait_vm<0>();     // tile 0 landed (tile 1 may fly)
  }
  __syncthreads();

  const int pi = lane & 15, g = lane >> 4;
#ifdef GRIDMM_AGG_PROF
  long long rp[6] = {0, 0, 0, 0, 0, 0}, rpt = RP_T();
#endif
  for (int i = 0; i < ntiles; ++i) {
    RP(1)
    const _Float16* s_tile = s_tiles + (size_t)(i % R) * PT * D;

    // ---- phase R: this wave's K-slice of the relevance products of tile i
    {
      // four independent MFMA chains (lo / hi fragment x point halves 0 / 1); at the boundary between the wave's two
      // token tiles their sums are set aside and the chains restart
      const f32x4_t zero4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      f32x4_t c0 = zero4, c1 = zero4, c2 = zero4, c3 = zero4, a0 = zero4, a1 = zero4, b0 = zero4, b1 = zero4;
      const f16x8_t* row0 = reinterpret_cast<const f16x8_t*>(s_tile + (size_t)pi * D);
      const f16x8_t* row1 = reinterpret_cast<const f16x8_t*>(s_tile + (size_t)(16 + pi) * D);
      constexpr int GP = 3;                     // pairs per fragment group, two groups of registers in flight
      // the ring feed (tile i + 2 into the slot of tile i - 1, read for the last time before the
      // barriers of iteration i - 1) goes out in slices between the
      // MFMA groups
      const bool fetch = is_dma && i + 2 < ntiles;
      if (fetch) dma_prepare(i + 2);
      f16x8_t fa[2][GP], fb[2][GP];
#pragma unroll
      for (int u = 0; u < GP; ++u) {
        const int ks = (gp0 + u) % KS;
        fa[0][u] = row0[(ks * 4 + g) ^ pi];
        fb[0][u] = row1[(ks * 4 + g) ^ pi];
      }
#pragma unroll
      for (int q = 0; q < PPW / GP; ++q) {
        if (fetch && q < 4) dma_rows(i + 2, 2 * q, 2 * q + 2);
        if (q + 1 < PPW / GP) {
#pragma unroll
          for (int u = 0; u < GP; ++u) {
            const int ks = (gp0 + (q + 1) * GP + u) % KS;
            fa[(q + 1) & 1][u] = row0[(ks * 4 + g) ^ pi];
            fb[(q + 1) & 1][u] = row1[(ks * 4 + g) ^ pi];
          }
        }
#pragma unroll
        for (int u = 0; u < GP; ++u) {
          const int p = q * GP + u;
          if (p == n_a) {                        // wave-uniform: token tile ct_a is complete
            a0 = c0 + c2; a1 = c1 + c3;
            c0 = c1 = c2 = c3 = zero4;
          }
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(tlo[p], fa[q & 1][u], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(tlo[p], fb[q & 1][u], c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(thi[p], fa[q & 1][u], c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(thi[p], fb[q & 1][u], c3, 0, 0, 0);
        }
      }
      if (n_a >= PPW) { a0 = c0 + c2; a1 = c1 + c3; } else { b0 = c0 + c2; b1 = c1 + c3; }
      if (fetch && i + 4 < ntiles) load_ids(i + 4);              // consumed two iterations from now
      RP(5)
      // lane (point pi, g) holds tokens 4 g .. 4 g + 3 of the token tile
      // Partial sums of a token tile come from up to three waves (consecutive K-slices): the first one stores, the
      // others add in turn, one barrier apart (LDS float atomics were ~1000 cycles per instruction here).
      const int rank_a = wave - (KS * ct_a) / PPW;              // this wave's turn for token tile ct_a: 0, 1 or 2
      float* pa0 = s_part + pi * PP + ct_a * 16 + 4 * g;
      float* pa1 = s_part + (16 + pi) * PP + ct_a * 16 + 4 * g;
      if (seg_a && rank_a == 0) { lds_store4(pa0, a0); lds_store4(pa1, a1); }
      if (seg_b) { lds_store4(pa0 + 16, b0); lds_store4(pa1 + 16, b1); }  // first wave of the next token tile
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // the LAST contributor of a token tile keeps the complete sums in registers: it reduces them to the tile's
      // maximum per point (over the valid tokens) instead of writing them back
      const bool last_a = (KS * ct_a + KS - 1) / PPW == wave;
      auto add_turn = [&]() {
        f32x4_t o0, o1;
        lds_load4x2(pa0, pa1, o0, o1);
        o0 += a0;
        o1 += a1;
        if (!last_a) {
          lds_store4(pa0, o0);
          lds_store4(pa1, o1);
        } else {
          float x0 = NEG_BIG, x1 = NEG_BIG;
          if (amax) {   // training: also the arg-max token, for the backward's routing
            int i0 = 0x7fffffff, i1 = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int tok = ct_a * 16 + 4 * g + r;
              const float v0 = tok < L ? o0[r] : NEG_BIG, v1 = tok < L ? o1[r] : NEG_BIG;
              if (v0 > x0) { x0 = v0; i0 = tok; }
              if (v1 > x1) { x1 = v1; i1 = tok; }
            }
            argmax_over_rows(x0, i0);
            argmax_over_rows(x1, i1);
            if (g == 0) {
              const unsigned aa = (unsigned)(size_t)(s_warg + pi * 8 + ct_a);
              asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:512" :: "v"(aa), "v"(i0), "v"(i1) : "memory");
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool colv = ct_a * 16 + 4 * g + r < L;
              x0 = fmaxf(x0, colv ? o0[r] : NEG_BIG);
              x1 = fmaxf(x1, colv ? o1[r] : NEG_BIG);
            }
            max_over_rows2(x0, x1);
          }
          if (g == 0) {
            const unsigned aw = (unsigned)(size_t)(s_wmax + pi * 8 + ct_a);
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:512" :: "v"(aw), "v"(x0), "v"(x1) : "memory");
          }
        }
      };
      if (seg_a && rank_a == 1) add_turn();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (seg_a && rank_a == 2) add_turn();
    }
    RP(2)
    // ---- barrier B: the partial sums of tile i are complete AND tile i + 1 has landed.  Queue of a DMA wave (in order):
    // iteration i issued DMA(i + 2) then IDS(i + 4) above; it needs DMA(i + 1) and IDS(i + 3) (issued at i - 1) and
    // leaves this iteration's two in flight.
    if (is_dma) {
      if (i + 4 < ntiles) wait_vm<RW * IPR + 1>();
      else if (i + 2 < ntiles) wait_vm<RW * IPR>();
      else wait_vm<0>();
    }
    RP(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    RP(3)
    if (wave == 7) {
      // relevance = max over the L instruction tokens (vilmodel.py:798) = max over the token tiles' maxima; lane = point
      const int p0 = p_lo + i * PT;
      const int npt = min(PT, p_hi - p0);
      const int lp = lane & (PT - 1);
      const unsigned a_w = (unsigned)(size_t)(s_wmax + lp * 8);
      float4 w0, w1;                            // (asm: see agg_accum.h -- this wave has global stores in flight)
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(w0), "=&v"(w1) : "v"(a_w) : "memory");
      const float w = fmaxf(fmaxf(fmaxf(w0.x, w0.y), fmaxf(w0.z, w0.w)), fmaxf(fmaxf(w1.x, w1.y), fmaxf(w1.z, w1.w)));
      if (lane < npt) relevance[(size_t)b * cap + p0 + lane] = w;   // by sorted position
      if (amax) {                               // first token tile that attains w
        int4 g0, g1;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(g0), "=&v"(g1) : "v"((unsigned)(size_t)(s_warg + lp * 8)) : "memory");
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const int wa[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        int arg = wa[0];
        float bestv = wv[0];
#pragma unroll
        for (int q = 1; q < 8; ++q)
          if (wv[q] > bestv) { bestv = wv[q]; arg = wa[q]; }
        if (lane < npt) amax[(size_t)b * cap + p0 + lane] = arg;
      }
    }
    RP(4)
  }
#ifdef GRIDMM_AGG_PROF
  if (blockIdx.x == 3 && blockIdx.y == 5 && lane == 0) {
    long long* o = g_relprof[wave];
    o[0] = rp[0]; o[1] = rp[1]; o[2] = rp[2]; o[3] = rp[3]; o[4] = rp[4]; o[5] = rp[5]; o[6] = ntiles;
  }
#endif
}

}  // namespace

#ifdef GRIDMM_AGG_PROF
extern "C" int gridmm_debug_rel_prof(long long* out) {      // development aid (-DGRIDMM_AGG_PROF builds only)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_relprof), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

// Returns GRIDMM_EINVAL when the shape is outside this variant's range.
int gridmm_grid_relevance_wide(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* relevance, int32_t* amax, int B, int cap, int Dd, int L, int n_chunks,
                               hipStream_t st) {
  const int Lt = (L + 15) / 16;
  if (Dd != D || Lt < 1 || Lt * KS > 8 * PPW || !relevance) return GRIDMM_EINVAL;   // L <= 80: 240 fragment units over 8 waves
  const size_t lds = (size_t)R * PT * D * 2 + (size_t)PT * PP * sizeof(float) + 2 * PT * 8 * sizeof(float) +
                     NDMA * 4 * RW * sizeof(int);
  auto kern = grid_relevance_wide_kernel;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess)
    return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(kern, dim3(n_chunks, B), dim3(512), lds, st, (const _Float16*)slab, perm, cell_start,
                (const _Float16*)text_frag, relevance, amax, cap, L, Lt, n_chunks);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
