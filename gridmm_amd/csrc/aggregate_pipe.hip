// Instruction-relevance grid aggregation, wave-specialised two-stage pipeline (the hot variant of aggregate.hip for
// L <= 80..96 instruction tokens; same math, same outputs, same entry point).
//
// aggregate.hip runs the three phases of a tile one after the other on all waves (relevance MFMAs | per-point softmax
// numerators | accumulation), each a latency-bound chain on a few waves: ~10 us per 64 points and CU, 1.7-1.9 TB/s.
// Here the 8 waves of a workgroup split into
//   R-waves (one per 16-column text tile, fragments register-resident): relevance of tile i          -> s_wmax[i & 1]
//   B-waves (the rest):  cell lookup of tile i, then softmax numerators + accumulation of tile i - 1  (s_wmax[(i-1) & 1])
// with ONE barrier per 32-point tile, so a tile costs max(relevance, softmax + accumulation) instead of their sum, and
// the LDS-DMA of tiles i+1 .. i+R-2 flies over both.  Ring: R slots of 32 points (4 x 32 KB at D <= 512: slot of tile
// i-1 being accumulated, slot of tile i in the matrix pipe, two tiles in flight; 3 x 48 KB at D = 768).
// Row ids come from scalar loads issued a whole iteration ahead (nothing but DMA in the vector-memory queue, so the
// counted s_waitcnt vmcnt is exact and no compiler-inserted vmcnt(0) drains the stream).
#include "common.h"

namespace {

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
constexpr int PT = 32;            // points per tile
constexpr int RPW = PT / 8;       // rows DMA'd per wave and tile (8 waves)
constexpr float NEG_BIG = -3.0e38f;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int KS, int R>   // D = 32 * KS
__global__ __launch_bounds__(512) void grid_aggregate_pipe_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const _Float16* __restrict__ text_frag, float* __restrict__ cells, uint8_t* __restrict__ occ,
    float* __restrict__ relevance, const int32_t* __restrict__ chunks, int cap, int L, int Lt, int n_chunks) {
  constexpr int D = 32 * KS;
  constexpr int NCH = D / 8;                 // 16-B chunks per row
  constexpr int IPR = (NCH + 63) / 64;       // DMA instructions per row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* s_tiles = reinterpret_cast<_Float16*>(smem);                        // [R][PT][D]
  float* s_wmax = reinterpret_cast<float*>(smem + (size_t)R * PT * D * 2);      // [2][8][PT]
  int* s_cell = reinterpret_cast<int*>(s_wmax + 2 * 8 * PT);                    // [2][PT]
  int* s_cs = s_cell + 2 * PT;                                                  // [198] cell_start of this episode

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, k = blockIdx.x;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  const int c_lo = chunks[(size_t)b * (n_chunks + 1) + k], c_hi = chunks[(size_t)b * (n_chunks + 1) + k + 1];
  if (c_lo >= c_hi) return;
  const int p_lo = cs[c_lo], p_hi = cs[c_hi];
  float* cells_b = cells + (size_t)b * GRIDMM_CELLS * D;
  uint8_t* occ_b = occ + (size_t)b * GRIDMM_CELLS;

  // empty cells of this chunk: zero vector, occ = 0 (vilmodel.py:803-807)
  for (int c = c_lo + wave; c < c_hi; c += 8) {
    if (cs[c + 1] == cs[c]) {
      for (int d = lane; d < D / 4; d += 64)
        reinterpret_cast<float4*>(cells_b + (size_t)c * D)[d] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane == 0) occ_b[c] = 0;
    }
  }
  if (p_lo >= p_hi) return;
  for (int i = tid; i < GRIDMM_CELLS + 2; i += 512) s_cs[i] = cs[i];

  const bool is_r = wave < Lt;                 // relevance wave (text column tile `wave`)
  const int tb = tid - Lt * 64;                // B-thread index (>= 0 on B-waves): feature dims 4 tb .. 4 tb + 3
  const bool acc_thread = !is_r && tb < D / 4;

  const size_t plane = (size_t)Lt * KS * 64 * 8;
  const _Float16* tf_b = text_frag + (size_t)b * 2 * plane + (size_t)lane * 8;
  f16x8_t thi[KS], tlo[KS];
  if (is_r) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      thi[ks] = *reinterpret_cast<const f16x8_t*>(tf_b + ((size_t)wave * KS + ks) * 64 * 8);
      tlo[ks] = *reinterpret_cast<const f16x8_t*>(tf_b + plane + ((size_t)wave * KS + ks) * 64 * 8);
    }
  }
  const _Float16* slab_b = slab + (size_t)b * cap * D;
  const int32_t* perm_b = perm + (size_t)b * cap;
  const int ntiles = (p_hi - p_lo + PT - 1) / PT;

  auto load_ids = [&](int t, int (&ids)[RPW]) {            // scalar loads: rows wave, wave + 8, ... of tile t
    const int p0 = p_lo + t * PT;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      int p = p0 + wave + 8 * j;
      if (p >= p_hi) p = p_hi - 1;                         // short tiles repeat the last valid row
      ids[j] = __builtin_amdgcn_readfirstlane(perm_b[p]);
    }
  };
  auto dma_tile = [&](int t, const int (&ids)[RPW]) {      // position c of row r holds global chunk c ^ (r & 15)
    _Float16* dst = s_tiles + (size_t)(t % R) * PT * D;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = wave + 8 * j;
      const _Float16* row = slab_b + (size_t)ids[j] * D;
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

  // accumulation state (B-waves; every B-wave keeps the same scalars, each accumulating thread its 4 dims)
  int cur = -1;
  float m_run = NEG_BIG, s_run = 0.f;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  auto flush = [&]() {
    if (cur < 0) return;
    const float inv = 1.0f / s_run;
    if (acc_thread)
      reinterpret_cast<float4*>(cells_b + (size_t)cur * D)[tb] = make_float4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
    if (tb == 0) occ_b[cur] = 1;
  };

  __builtin_amdgcn_s_waitcnt(0);                // text fragments / cell_start: retire ordinary loads before the loop
  __syncthreads();
  int ids[RPW];
  for (int t = 0; t < R - 2 && t < ntiles; ++t) { load_ids(t, ids); dma_tile(t, ids); }
  if (R - 2 < ntiles) load_ids(R - 2, ids);     // row ids of the next tile to issue

  for (int i = 0; i <= ntiles; ++i) {
    // tile i must have landed; the R - 3 younger tiles stay in flight
    if (i + R - 3 < ntiles && R > 3) wait_vm<(R > 3 ? (R - 3) * RPW * IPR : 0)>(); else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // tile i and the products of iteration i-1 visible; slot of tile i-2 free
    if (i + R - 2 < ntiles) {
      dma_tile(i + R - 2, ids);
      if (i + R - 1 < ntiles) load_ids(i + R - 1, ids);    // lands during this iteration's compute
    }
    if (is_r) {
      // ---- relevance of tile i on the matrix pipe (text fragment = A operand: lane = point, registers = columns)
      if (i < ntiles) {
        const _Float16* s_tile = s_tiles + (size_t)(i % R) * PT * D;
        const int pi = lane & 15, g = lane >> 4;
        constexpr int GK = KS > 16 ? 2 : 4;
        f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        const f16x8_t* row0 = reinterpret_cast<const f16x8_t*>(s_tile + (size_t)pi * D);
        const f16x8_t* row1 = reinterpret_cast<const f16x8_t*>(s_tile + (size_t)(16 + pi) * D);
        f16x8_t fa[2][GK], fb[2][GK];
#pragma unroll
        for (int u = 0; u < GK; ++u) { fa[0][u] = row0[(u * 4 + g) ^ pi]; fb[0][u] = row1[(u * 4 + g) ^ pi]; }
#pragma unroll
        for (int q = 0; q < KS / GK; ++q) {
          if (q + 1 < KS / GK) {
#pragma unroll
            for (int u = 0; u < GK; ++u) {
              fa[(q + 1) & 1][u] = row0[(((q + 1) * GK + u) * 4 + g) ^ pi];
              fb[(q + 1) & 1][u] = row1[(((q + 1) * GK + u) * 4 + g) ^ pi];
            }
          }
#pragma unroll
          for (int u = 0; u < GK; ++u) {
            const int ks = q * GK + u;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(tlo[ks], fa[q & 1][u], acc0, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(thi[ks], fa[q & 1][u], acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(tlo[ks], fb[q & 1][u], acc1, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(thi[ks], fb[q & 1][u], acc3, 0, 0, 0);
          }
        }
        float x0 = NEG_BIG, x1 = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool colv = (wave * 16 + 4 * g + r) < L;
          x0 = fmaxf(x0, colv ? acc0[r] + acc2[r] : NEG_BIG);
          x1 = fmaxf(x1, colv ? acc1[r] + acc3[r] : NEG_BIG);
        }
        x0 = fmaxf(x0, __shfl_xor(x0, 16, 64)); x1 = fmaxf(x1, __shfl_xor(x1, 16, 64));
        x0 = fmaxf(x0, __shfl_xor(x0, 32, 64)); x1 = fmaxf(x1, __shfl_xor(x1, 32, 64));
        if (g == 0) {
          float* wm = s_wmax + ((i & 1) * 8 + wave) * PT;
          wm[pi] = x0;
          wm[16 + pi] = x1;
        }
      }
    } else {
      // ---- cell of each point of tile i (last wave; binary search on LDS) -> consumed next iteration
      if (wave == 7 && i < ntiles && lane < PT) {
        int cell = -1;
        const int p = p_lo + i * PT + lane;
        if (p < p_hi) {
          int lo = c_lo, hi = c_hi;  // invariant cs[lo] <= p < cs[hi]; the last c with cs[c] <= p owns p
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_cs[mid] <= p) lo = mid; else hi = mid;
          }
          cell = lo;
        }
        s_cell[(i & 1) * PT + lane] = cell;
      }
      // ---- tile i - 1: softmax numerators (every B-wave for itself, lane = point), then accumulation
      if (i >= 1) {
        const int t = i - 1;
        const int p0 = p_lo + t * PT;
        const int npt = min(PT, p_hi - p0);
        const _Float16* s_tile = s_tiles + (size_t)(t % R) * PT * D;
        const float* wm = s_wmax + (t & 1) * 8 * PT;
        float w = NEG_BIG;
        if (lane < PT)
          for (int q = 0; q < Lt; ++q) w = fmaxf(w, wm[q * PT + lane]);
        if (relevance && wave == 7 && lane < npt) relevance[(size_t)b * cap + p0 + lane] = w;   // by sorted position
        const int c = (lane < npt) ? s_cell[(t & 1) * PT + lane] : -2 - lane;   // unique sentinel: never joins a run
        if (lane >= npt) w = NEG_BIG;
        // Points are sorted by cell, so a cell is a contiguous run of lanes [rs, re].  The run bounds come from the
        // ballot of run heads (no cell-id shuffles), the run maximum from ONE segmented prefix-max scan (5 cross-lane
        // steps) read back at the run's last lane: 7 ds_bpermute round trips instead of 21.
        const int cprev = __shfl_up(c, 1, 64);
        const bool head = (lane < npt) && (lane == 0 || cprev != c);
        unsigned long long heads = __ballot(head);
        const unsigned long long below = heads & ((2ull << lane) - 1ull);          // heads at or below this lane
        const int rs = below ? 63 - __builtin_clzll(below) : lane;
        const unsigned long long above = lane < 63 ? heads & ~((2ull << lane) - 1ull) : 0ull;   // heads above this lane
        const int re = min(above ? __builtin_ctzll(above) - 1 : npt - 1, max(npt - 1, 0));
        float pre = w;
#pragma unroll
        for (int o = 1; o < PT; o <<= 1) {
          const float pu = __shfl_up(pre, o, 64);
          if (lane - o >= rs) pre = fmaxf(pre, pu);
        }
        float m = __shfl(pre, lane < npt ? re : lane, 64);                 // prefix max at the run's last lane = run max
        if (c == cur) m = fmaxf(m, m_run);                       // the run continuing from the previous tile
        const float e_lane = (lane < npt) ? expf(w - m) : 0.f;
        const float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
        const int c0 = __builtin_amdgcn_readfirstlane(c);
        const float m_last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), npt - 1));
        auto e_of = [&](int r) -> float {
          return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e_lane), r));
        };
        if (cur >= 0 && c0 == cur) {
          const float sc = expf(m_run - m0);                     // rescale of the running cell
          s_run *= sc;
#pragma unroll
          for (int d = 0; d < 4; ++d) v[d] *= sc;
        }
        const _Float16* my = s_tile + (tb & 1) * 4;              // this thread's 4 dims inside chunk (tb >> 1)
        const int chunk = tb >> 1;
        while (heads) {
          const int r0 = __builtin_ctzll(heads);
          heads &= heads - 1;
          const int r1 = heads ? __builtin_ctzll(heads) : npt;
          const int cc = __builtin_amdgcn_readlane(c, r0);
          if (cc != cur) {
            flush();
            cur = cc; s_run = 0.f;
#pragma unroll
            for (int d = 0; d < 4; ++d) v[d] = 0.f;
          }
          int r = r0;
          for (; r + 8 <= r1; r += 8) {          // 8 rows per group: all LDS reads issued before the first FMA
            float e[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) e[u] = e_of(r + u);
            if (acc_thread) {
              f16x4_t h[8];
#pragma unroll
              for (int u = 0; u < 8; ++u)
                h[u] = *reinterpret_cast<const f16x4_t*>(my + (size_t)(r + u) * D + ((chunk ^ ((r + u) & 15)) * 8));
              float a[4] = {0.f, 0.f, 0.f, 0.f}, bq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int u = 0; u < 8; u += 2)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                  a[d] += e[u] * (float)h[u][d];
                  bq[d] += e[u + 1] * (float)h[u + 1][d];
                }
#pragma unroll
              for (int d = 0; d < 4; ++d) v[d] += a[d] + bq[d];
            }
            s_run += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
          }
          for (; r < r1; ++r) {
            const float e = e_of(r);
            s_run += e;
            if (acc_thread) {
              const f16x4_t h = *reinterpret_cast<const f16x4_t*>(my + (size_t)r * D + ((chunk ^ (r & 15)) * 8));
#pragma unroll
              for (int d = 0; d < 4; ++d) v[d] += e * (float)h[d];
            }
          }
        }
        m_run = m_last;
      }
    }
  }
  if (!is_r) flush();
}

}  // namespace

// Returns GRIDMM_EINVAL when the shape is outside this variant's range (the caller then uses the generic kernel).
int gridmm_grid_aggregate_pipe(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* cells, uint8_t* occ, float* relevance, const int32_t* chunks, int B, int cap, int D,
                               int L, int n_chunks, hipStream_t st) {
  const int Lt = (L + 15) / 16;
  if (D != 512 && D != 256) return GRIDMM_EINVAL;            // D = 768: 192 VGPRs of resident fragments spill
  if (Lt < 1 || 8 - Lt < (D / 4 + 63) / 64) return GRIDMM_EINVAL;   // need enough B-waves for the feature dims
  constexpr int R = 4;
  const size_t lds = (size_t)R * PT * D * 2 + 2 * 8 * PT * sizeof(float) + 2 * PT * sizeof(int) + 200 * sizeof(int);
  dim3 grid(n_chunks, B), block(512);
#define GRIDMM_AGGP(KS)                                                                                              \
  do {                                                                                                               \
    auto kern = grid_aggregate_pipe_kernel<KS, R>;                                                                   \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,         \
                            (int)lds) != hipSuccess)                                                                 \
      return GRIDMM_EINVAL;                                                                                          \
    GRIDMM_LAUNCH(kern, grid, block, lds, st, (const _Float16*)slab, perm, cell_start, (const _Float16*)text_frag,   \
                  cells, occ, relevance, chunks, cap, L, Lt, n_chunks);                                              \
  } while (0)
  if (D == 512) GRIDMM_AGGP(16); else GRIDMM_AGGP(8);
#undef GRIDMM_AGGP
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
