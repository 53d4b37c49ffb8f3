// Instruction-relevance grid aggregation, wave-specialised two-stage pipeline (the hot variant of aggregate.hip for
// L <= 80..96 instruction tokens; same math, same outputs, same entry point).
//
// aggregate.hip runs the three phases of a tile one after the other on all waves (relevance MFMAs | per-point softmax
// numerators | accumulation), each a latency-bound chain on a few waves: ~10 us per 64 points and CU, 1.7-1.9 TB/s.
// Here the 8 waves of a workgroup split into
//   R-waves (one per 16-column text tile, fragments register-resident): relevance of tile i          -> s_wmax[i & 1]
//   B-waves (the rest):  cell lookup of tile i, then softmax numerators + accumulation of tile i - 1  (s_wmax[(i-1) & 1])
// The accumulation is a matrix product as well: out^T[dim][slot] = X^T[dim][point] . E[point][slot], where E holds the
// softmax numerator of a point in the column of its cell's slot (f16 hi + lo, exact to ~2^-22) and X^T comes straight
// out of the row-major LDS tile through ds_read_b64_tr_b16.  A cell keeps its slot (= MFMA output column, one per
// lane & 15) from tile to tile, so the open cell's partial sum never moves between lanes; a ones block yields the
// softmax denominators in the same layout.  16 dims x 32 points x 16 cells per MFMA pair instead of ~6 VALU
// instructions per point and thread.
// with ONE barrier per 32-point tile, so a tile costs max(relevance, softmax + accumulation) instead of their sum, and
// the LDS-DMA of tiles i+1 .. i+R-2 flies over both.  Ring: R slots of 32 points (4 x 32 KB at D <= 512: slot of tile
// i-1 being accumulated, slot of tile i in the matrix pipe, two tiles in flight; 3 x 48 KB at D = 768).
// Row ids come from scalar loads issued a whole iteration ahead (nothing but DMA in the vector-memory queue, so the
// counted s_waitcnt vmcnt is exact and no compiler-inserted vmcnt(0) drains the stream).
#include "common.h"

namespace {

#ifdef GRIDMM_AGG_PROF
__device__ long long g_prof[8][8];
#define PROF_T() ((long long)__builtin_readcyclecounter())
#define PROF_MARK(k) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t_ = PROF_T(); p_m[k] += t_ - pmk; pmk = t_; }
#else
#define PROF_MARK(k)
#endif

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
constexpr int PT = 32;            // points per tile
constexpr int MAXR = 12;          // rows DMA'd per R-wave and tile, at most (ceil(PT / Lt), Lt >= 3)
constexpr float NEG_BIG = -3.0e38f;
constexpr int TAB_BYTES = 320;   // per-B-wave tables: 32 x (f16 hi, f16 lo, u16 run)

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS loads of the B-waves go through asm: when the destination VGPRs of a compiler-visible LDS load were operands of
// a still pending global store or LDS-DMA, the waitcnt pass answers with s_waitcnt vmcnt(0), which drains the tile ring
// in the middle of every iteration (measured: ~2000 of 6000 cycles per tile).  The asm forms carry their own lgkmcnt.
__device__ __forceinline__ int lds_ld_b32(const void* p) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  return v;
}
__device__ __forceinline__ void lds_tr2(uint2& x0, uint2& x1, const void* a0, const void* a1) {   // no wait: see lgkm_fence
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"
               : "=&v"(x0), "=&v"(x1) : "v"((unsigned)(size_t)a0), "v"((unsigned)(size_t)a1) : "memory");
}
// s_waitcnt lgkmcnt(n) that the uses of x0/x1 cannot be scheduled above (n = LDS reads issued after the pair's)
__device__ __forceinline__ void lgkm_fence(int n, uint2& x0, uint2& x1) {
  switch (n) {
    case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1)::"memory"); break;
    case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x0), "+v"(x1)::"memory"); break;
    case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x0), "+v"(x1)::"memory"); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(x0), "+v"(x1)::"memory"); break;
    default: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x0), "+v"(x1)::"memory"); break;
  }
}
__device__ __forceinline__ void reg_fence(uint2& x0, uint2& x1) { asm volatile("" : "+v"(x0), "+v"(x1)::"memory"); }

template <int KS, int R, int NBW>   // D = 32 * KS; NBW = 16-dim blocks per B-wave (at least ceil(D / 16 / B-waves))
__global__ __launch_bounds__(512) void grid_aggregate_pipe_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const _Float16* __restrict__ text_frag, float* __restrict__ cells, uint8_t* __restrict__ occ,
    float* __restrict__ relevance, const int32_t* __restrict__ chunks, int cap, int L, int Lt, int n_chunks) {
  constexpr int D = 32 * KS;
  constexpr int NCH = D / 8;                 // 16-B chunks per row
  constexpr int IPR = (NCH + 63) / 64;       // DMA instructions per row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* s_tiles = reinterpret_cast<_Float16*>(smem);                        // [R][PT][D]
  float* s_wmax = reinterpret_cast<float*>(smem + (size_t)R * PT * D * 2);      // [2][PT][8]: row = point, column = R-wave
  int* s_cs = reinterpret_cast<int*>(s_wmax + 2 * 8 * PT);                      // [200] cell_start of this episode
  unsigned char* s_tab = reinterpret_cast<unsigned char*>(s_cs + 200);          // [8] per-B-wave tables, TAB_BYTES each
  int* s_ids = reinterpret_cast<int*>(s_tab + 8 * TAB_BYTES);                   // [8 waves][4 tiles][MAXR] slab rows to fetch
  int* s_necell = s_ids + 8 * 4 * MAXR;                                         // [200] non-empty cells of this chunk, in order
  unsigned* s_hbits = reinterpret_cast<unsigned*>(s_necell + 200);              // [ntiles] bit j of word t: a cell starts at
                                                                                // point 32 t + j of the chunk (run heads)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GRIDMM_AGG_PROF
  const long long pt0 = PROF_T();
  long long p_wait = 0, p_dma = 0, p_work = 0, p_a = 0, pt2 = 0, pta = 0, pmk = 0;
  long long p_m[6] = {0, 0, 0, 0, 0, 0};
#endif
  const int b = blockIdx.y, k = blockIdx.x;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  // Chunk k of the episode = cells [c_lo, c_hi), cut where the sorted point index crosses k * ceil(valid / n_chunks)
  // (the same boundaries build_chunks_kernel writes for the generic kernel; computed here, a 15 us serial kernel
  // less): cell_start is non-decreasing, so a boundary is the count of entries below the target.
  const int mine = tid < GRIDMM_CELLS + 2 ? cs[tid] : 0x7fffffff;
  if (tid < GRIDMM_CELLS + 2) s_cs[tid] = mine;
  const int valid = cs[GRIDMM_CELLS];
  const long target = (valid + n_chunks - 1) / n_chunks;
  const bool counted = tid <= GRIDMM_CELLS;
  const int below_lo = __syncthreads_count(counted && mine < k * target);
  const int below_hi = __syncthreads_count(counted && mine < (k + 1) * target);
  const int c_lo = k == 0 ? 0 : min(below_lo, GRIDMM_CELLS);
  const int c_hi = k + 1 == n_chunks ? GRIDMM_CELLS : min(below_hi, GRIDMM_CELLS);
  if (c_lo >= c_hi) return;
  const int p_lo = s_cs[c_lo], p_hi = s_cs[c_hi];
  float* cells_b = cells + (size_t)b * GRIDMM_CELLS * D;
  uint8_t* occ_b = occ + (size_t)b * GRIDMM_CELLS;

  // empty cells of this chunk: zero vector, occ = 0 (vilmodel.py:803-807)
  for (int c = c_lo + wave; c < c_hi; c += 8) {
    if (s_cs[c + 1] == s_cs[c]) {
      for (int d = lane; d < D / 4; d += 64)
        reinterpret_cast<float4*>(cells_b + (size_t)c * D)[d] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane == 0) occ_b[c] = 0;
    }
  }
  if (p_lo >= p_hi) return;
  for (int i = tid; i < 2 * 8 * PT; i += 512) s_wmax[i] = NEG_BIG;      // columns of absent R-waves stay at -inf
  {
    // Points are sorted by cell: the run heads of every tile are known from cell_start alone.  One bit per point (a
    // tile's heads = one aligned word) + the list of non-empty cells replace a per-point cell lookup in the loop.
    const int nw = (p_hi - p_lo + PT - 1) / PT;
    for (int i = tid; i < nw; i += 512) s_hbits[i] = 0u;
    __syncthreads();
    if (wave == 0) {
      int kbase = 0;
      for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
        const int c = c0 + lane;
        const int st = c < c_hi ? s_cs[c] : 0;
        const bool ne = c < c_hi && s_cs[c + 1] > st;
        const unsigned long long mk = __ballot(ne);
        if (ne) {
          s_necell[kbase + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] = c;
          atomicOr(&s_hbits[(st - p_lo) >> 5], 1u << ((st - p_lo) & 31));
        }
        kbase += __builtin_popcountll(mk);
      }
    }
  }

  const bool is_r = wave < Lt;                 // relevance wave (text column tile `wave`)

  const size_t plane = (size_t)Lt * KS * 64 * 8;
  const _Float16* tf_b = text_frag + (size_t)b * 2 * plane + (size_t)lane * 8;
  const _Float16* slab_b = slab + (size_t)b * cap * D;
  const int32_t* perm_b = perm + (size_t)b * cap;
  const int ntiles = (p_hi - p_lo + PT - 1) / PT;

  // The R-waves feed the ring (the B-waves have global stores in flight, which would make a counted vmcnt wait drain
  // their part of it, and they are the longer leg of an iteration): R-wave w fetches rows w, w + Lt, ... of a tile.
  // Row ids travel by LDS-DMA as well (one lane x 4 B per row, two iterations ahead of their use): a scalar load here
  // would put ~1200 cycles of memory latency into EVERY lgkmcnt wait of the iteration (SMEM returns out of order, so
  // LDS waits cannot be counted past it), and a vector load's result register makes the compiler drain the DMA queue.
  // In-order vmcnt covers the ids like the tiles.
  const int my_rows = is_r ? (PT - wave + Lt - 1) / Lt : 0;
  auto load_ids = [&](int t) {                             // -> s_ids[wave][t & 3][j]
    if (lane < my_rows) {
      int p = p_lo + t * PT + wave + Lt * lane;
      if (p >= p_hi) p = p_hi - 1;                         // short tiles repeat the last valid row
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(perm_b + p),
                                       (__attribute__((address_space(3))) void*)(s_ids + (wave * 4 + (t & 3)) * MAXR),
                                       4, 0, 0);
    }
  };
  int ids_s[MAXR];                                         // slab rows of the tile being fetched (wave-uniform)
  auto dma_prepare = [&](int t) {
    static_assert(MAXR == 12, "ids are read as three int4");
    // (asm: the compiler's waitcnt pass answers a visible ds_read here with s_waitcnt vmcnt(0), draining the ring)
    int4 idv[3];
    {
      const unsigned a = (unsigned)(size_t)(s_ids + (wave * 4 + (t & 3)) * MAXR);         // uniform address: broadcast
      asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(idv[0]), "=&v"(idv[1]), "=&v"(idv[2]) : "v"(a) : "memory");
    }
    const int idl[MAXR] = {idv[0].x, idv[0].y, idv[0].z, idv[0].w, idv[1].x, idv[1].y, idv[1].z, idv[1].w,
                           idv[2].x, idv[2].y, idv[2].z, idv[2].w};
#pragma unroll
    for (int j = 0; j < MAXR; ++j) ids_s[j] = __builtin_amdgcn_readfirstlane(idl[j]);
  };
  auto dma_rows = [&](int t, int j0, int j1) {             // position c of row r holds global chunk c ^ (r & 15)
    _Float16* dst = s_tiles + (size_t)(t % R) * PT * D;
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      if (j < my_rows) {
        const int r = wave + Lt * j;
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
    }
  };
  auto wait_vm_dyn = [&](int n) {                          // s_waitcnt vmcnt(n), n wave-uniform
    switch (n) {
      case 13: wait_vm<13>(); break;
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
  };

  // accumulation state (B-waves).  Every B-wave keeps the same scalars; B-wave bw owns the 16-dim blocks bw, bw + nbw,
  // ... and in each the MFMA output layout: lane -> slot lane & 15, dims 4 (lane >> 4) .. + 3 of the block.
  constexpr int NBLK = D / 16;
  const int bw = wave - Lt, nbw = 8 - Lt;
  const int sl = lane & 15, g = lane >> 4;
  int base = 0, n_heads = 0;                   // slot of the open cell; run heads before the current tile
  float m_run = NEG_BIG;
  f32x4_t acc[NBW], acc_s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NBW; ++u) acc[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  unsigned char* tab = s_tab + (size_t)(bw < 0 ? 0 : bw) * TAB_BYTES;
  _Float16* t_ehi = reinterpret_cast<_Float16*>(tab);               // [32] numerators, f16 hi, in fragment order
  _Float16* t_elo = reinterpret_cast<_Float16*>(tab + 64);          // [32] f16 lo
  unsigned short* t_q = reinterpret_cast<unsigned short*>(tab + 128);   // [32] run index of the point
  auto flush_rows = [&](bool doit, int cell) {   // normalise + store + clear the rows of the lanes with doit
    const float inv = __builtin_amdgcn_rcpf(acc_s[0]);
    if (doit) {                                  // one exec region for all stores (the block guard is wave-uniform)
      float* dst = cells_b + (size_t)cell * D + bw * 16 + g * 4;
#pragma unroll
      for (int u = 0; u < NBW; ++u)
        if (bw + u * nbw < NBLK)
          *reinterpret_cast<float4*>(dst + u * nbw * 16) =
              make_float4(acc[u][0] * inv, acc[u][1] * inv, acc[u][2] * inv, acc[u][3] * inv);
      if (g == 0 && bw == 0) occ_b[cell] = 1;
    }
#pragma unroll
    for (int u = 0; u < NBW; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[u][j] = doit ? 0.f : acc[u][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_s[j] = doit ? 0.f : acc_s[j];
  };

  // Queue discipline of an R-wave (in order): iteration i issues DMA(i + R - 2) then IDS(i + R); its top needs DMA(i)
  // (issued at i - R + 2) and IDS(i + R - 2) (issued at i - 2) and leaves the R - 3 younger tiles and IDS(i + R - 1)
  // in flight.
  static_assert(R == 3 || R == 4, "ring of 3 (D = 768) or 4 slots");
  auto iter_head_r = [&](int i) {
#ifdef GRIDMM_AGG_PROF
    pt2 = PROF_T(); if (i > 0) p_work += pt2 - pta;
#endif
    if (i >= 1 && i + R - 1 < ntiles) wait_vm_dyn((R - 3) * my_rows * IPR + 1);   // steady state
    else if (i + 1 < ntiles) wait_vm_dyn((R - 3) * my_rows * IPR);                // first / last iterations: tiles only
    else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef GRIDMM_AGG_PROF
    { const long long t = PROF_T(); p_m[4] += t - pt2; }
#endif
    __builtin_amdgcn_s_barrier();               // tile i and the products of iteration i-1 visible; slot of tile i-2 free
#ifdef GRIDMM_AGG_PROF
    { const long long t = PROF_T(); p_wait += t - pt2; pt2 = t; }
#endif
    if (i + R - 2 < ntiles) dma_prepare(i + R - 2);        // its rows are issued between the MFMA groups below
#ifdef GRIDMM_AGG_PROF
    pta = PROF_T(); p_dma += pta - pt2;
#endif
  };
  auto iter_head_b = [&](int i) {
#ifdef GRIDMM_AGG_PROF
    pt2 = PROF_T(); if (i > 0) p_work += pt2 - pta;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef GRIDMM_AGG_PROF
    { const long long t = PROF_T(); p_wait += t - pt2; pta = t; }
#endif
  };
  // Two loops (same barrier sequence) so that the register allocator never sees the R-waves' text fragments and the
  // B-waves' accumulators live at the same time.
  if (is_r) {
    f16x8_t thi[KS], tlo[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      thi[ks] = *reinterpret_cast<const f16x8_t*>(tf_b + ((size_t)wave * KS + ks) * 64 * 8);
      tlo[ks] = *reinterpret_cast<const f16x8_t*>(tf_b + plane + ((size_t)wave * KS + ks) * 64 * 8);
    }
    __builtin_amdgcn_s_waitcnt(0);              // text fragments: retire ordinary loads before the loop
    for (int t = 0; t < R && t < ntiles; ++t) load_ids(t);
    wait_vm<0>();
    __syncthreads();                            // cell_start, run heads, s_wmax initialised
    for (int t = 0; t < R - 2 && t < ntiles; ++t) { dma_prepare(t); dma_rows(t, 0, MAXR); }
    for (int i = 0; i <= ntiles; ++i) {
      iter_head_r(i);
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
        const bool fetch = i + R - 2 < ntiles;                    // DMA(i + R - 2): a slice of its rows per MFMA group,
        constexpr int NQ = KS / GK, RQ = (MAXR + NQ - 1) / NQ;     // so that the address path works under the MFMAs
#pragma unroll
        for (int q = 0; q < KS / GK; ++q) {
          if (fetch) dma_rows(i + R - 2, q * RQ, min((q + 1) * RQ, MAXR));
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
        if (fetch && i + R < ntiles) load_ids(i + R);              // consumed two iterations from now
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
          float* wm = s_wmax + (i & 1) * 8 * PT + wave;
          wm[pi * 8] = x0;
          wm[(16 + pi) * 8] = x1;
        }
      }
    }
  } else {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = 0; i <= ntiles; ++i) {
      iter_head_b(i);
      // ---- tile i - 1: softmax numerators (every B-wave for itself, lane = point), then accumulation
      if (i >= 1) {
#ifdef GRIDMM_AGG_PROF
        pmk = PROF_T();
#endif
        const int t = i - 1;
        const int p0 = p_lo + t * PT;
        const int npt = min(PT, p_hi - p0);
        const _Float16* s_tile = s_tiles + (size_t)(t % R) * PT * D;
        const int lp = lane & (PT - 1);                            // lanes >= PT mirror (results unused)
        float w;
        unsigned hb;
        {
          const unsigned a_w = (unsigned)(size_t)(s_wmax + ((t & 1) * PT + lp) * 8);
          const unsigned a_h = (unsigned)(size_t)(s_hbits + t);
          float4 w0, w1;
          asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b32 %2, %4\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(w0), "=&v"(w1), "=&v"(hb) : "v"(a_w), "v"(a_h) : "memory");
          w = fmaxf(fmaxf(fmaxf(w0.x, w0.y), fmaxf(w0.z, w0.w)), fmaxf(fmaxf(w1.x, w1.y), fmaxf(w1.z, w1.w)));
        }
        if (relevance && wave == 7 && lane < npt) relevance[(size_t)b * cap + p0 + lane] = w;   // by sorted position
        if (lane >= npt) w = NEG_BIG;
        // A cell is a contiguous run of lanes [rs, re]; the run heads of this tile are one word of s_hbits (bit 0 clear:
        // the first run continues the open cell of the previous tile).  The run maximum at every lane = max(segmented
        // prefix max, segmented suffix max): DPP row shifts (a VALU modifier) + two scalar readlanes for the seam between
        // the 16-lane rows; the ds_bpermute form of the same scans was a ~1000-cycle serial chain per tile.
        const unsigned hbu = (unsigned)__builtin_amdgcn_readfirstlane((int)hb);
        const bool cont = !(hbu & 1u);                             // (tile 0 starts at a cell boundary: never cont)
        const unsigned heads = hbu | 1u;
        const unsigned le = (2u << lp) - 1u;                       // lanes at or below this one
        const unsigned below = heads & le, above = heads & ~le;
        const int rs = 31 - __builtin_clz(below);
        const int re = min(above ? __builtin_ctz(above) - 1 : npt - 1, npt - 1);
        const int negb = __builtin_bit_cast(int, NEG_BIG);
        float pre = w, suf = w;
#define GRIDMM_SCAN_STEP(O)                                                                                          \
        {                                                                                                            \
          const float pu = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(                                    \
              negb, __builtin_bit_cast(int, pre), 0x110 + O, 0xf, 0xf, false)); /* row_shr:O */                      \
          const float su = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(                                    \
              negb, __builtin_bit_cast(int, suf), 0x100 + O, 0xf, 0xf, false)); /* row_shl:O */                      \
          if (lp - O >= rs) pre = fmaxf(pre, pu);                                                                    \
          if (lp + O <= re) suf = fmaxf(suf, su);                                                                    \
        }
        GRIDMM_SCAN_STEP(1) GRIDMM_SCAN_STEP(2) GRIDMM_SCAN_STEP(4) GRIDMM_SCAN_STEP(8)
#undef GRIDMM_SCAN_STEP
        {   // the seam between lanes 15 | 16: a run crossing it takes the other row's partial result
          const float p15 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pre), 15));
          const float s16 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, suf), 16));
          if (lane >= 16 && rs <= 15) pre = fmaxf(pre, p15);
          if (lane <= 15 && re >= 16) suf = fmaxf(suf, s16);
        }
        float m = fmaxf(pre, suf);
        const int q_lane = __builtin_popcount(below) - 1;          // run index inside the tile
        if (cont && q_lane == 0) m = fmaxf(m, m_run);              // the run continuing from the previous tile
        const float e_lane = (lane < npt) ? expf(w - m) : 0.f;
        const float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
        const float m_last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), npt - 1));
        const int nruns = __builtin_popcount(heads);
        const int kg0 = n_heads - (cont ? 1 : 0);                  // run q of this tile is non-empty cell kg0 + q of the chunk
        // per-wave tables: numerators (hi / lo) and run index in fragment order (k = 8 g + 4 h + jj <-> point 8 jj + 2 g + h,
        // the order in which the transpose reads deliver the points)
        if (lane < PT) {
          const _Float16 eh = (_Float16)e_lane;
          const _Float16 el = (_Float16)(e_lane - (float)eh);
          const int idx = ((lane >> 1) & 3) * 8 + (lane & 1) * 4 + (lane >> 3);
          t_ehi[idx] = eh;
          t_elo[idx] = el;
          t_q[idx] = (unsigned short)(lane < npt ? q_lane : 0xFFFF);
        }
        // The open cell ended with the previous tile: its row (slot base) is stored by the flush of this tile's first
        // pass, unless that pass needs all 16 slots.
        bool flush_old = t > 0 && !cont;
        if (flush_old && nruns >= 16) {
          flush_rows(sl == base, lds_ld_b32(s_necell + n_heads - 1));
          flush_old = false;
        }
        const int start = t == 0 ? 0 : (cont ? base : ((base + 1) & 15));
        if (cont) {
          const float sc = expf(m_run - m0);                     // rescale of the running cell
          if (sc != 1.0f) {
            const float f = sl == base ? sc : 1.0f;
#pragma unroll
            for (int u = 0; u < NBW; ++u)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[u][j] *= f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc_s[j] *= f;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // own table writes (single wave: program order)
        PROF_MARK(0)
#ifdef GRIDMM_AGG_PROF
        p_a += PROF_T() - pta;
#endif
        __builtin_amdgcn_wave_barrier();
        uint4 hv, lv, qvu;                                        // (asm for the same reason as in dma_tile)
        {
          const unsigned a = (unsigned)(size_t)(t_ehi + g * 8);
          asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:64\n\tds_read_b128 %2, %3 offset:128\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(hv), "=&v"(lv), "=&v"(qvu) : "v"(a) : "memory");
        }
        const u16x8_t qv = __builtin_bit_cast(u16x8_t, qvu);
        // transpose-read addresses: lane i of a 16-lane group points at 4 dims ((i & 3) * 4 ..) of point
        // (i >> 2) * 8 + 2 g + h; the group receives dim i of those 4 points (ds_read_b64_tr_b16)
        const int li = lane & 15;
        const int row_h0 = (li >> 2) * 8 + 2 * g;
        const f16x8_t ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f,
                              (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
        for (int q0 = 0; q0 < nruns; q0 += 16) {
          const int qs = q0 + ((sl - start) & 15);               // the run this lane's slot holds in this pass
          const u16x8_t dq = qv ^ (unsigned short)qs;
          const u16x8_t one16 = 1;
          const u16x8_t msk = __builtin_elementwise_min(dq, one16) - one16;     // 0xFFFF where the point is in run qs
          const uint4 mk = __builtin_bit_cast(uint4, msk);
          const uint4 bhu = make_uint4(hv.x & mk.x, hv.y & mk.y, hv.z & mk.z, hv.w & mk.w);
          const uint4 blu = make_uint4(lv.x & mk.x, lv.y & mk.y, lv.z & mk.z, lv.w & mk.w);
          const f16x8_t bh = __builtin_bit_cast(f16x8_t, bhu), bl = __builtin_bit_cast(f16x8_t, blu);
          acc_s = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, bh, acc_s, 0, 0, 0);
          acc_s = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, bl, acc_s, 0, 0, 0);
          PROF_MARK(1)
          int cell;                                               // cell of this lane's run: needed after the MFMAs
          asm volatile("ds_read_b32 %0, %1" : "=v"(cell)
                       : "v"((unsigned)(size_t)(s_necell + (flush_old && sl == base ? n_heads - 1
                                                                                   : min(kg0 + qs, GRIDMM_CELLS - 1))))
                       : "memory");
          // transpose reads in groups of GB blocks, one group ahead of the MFMAs that consume them
          constexpr int GB = 4, NG = (NBW + GB - 1) / GB;
          uint2 xr[NBW][2];
          auto issue_group = [&](int gi) {
#pragma unroll
            for (int u = gi * GB; u < gi * GB + GB && u < NBW; ++u) {
              const int mb = min(bw + u * nbw, NBLK - 1);        // surplus blocks recompute the last one (never stored)
              const int gc = 2 * mb + ((li & 3) >> 1);           // global 16-B chunk of this lane's 4 dims
              const int r0 = row_h0, r1 = row_h0 + 1;
              lds_tr2(xr[u][0], xr[u][1], s_tile + (size_t)r0 * D + ((gc ^ (r0 & 15)) * 8 + (li & 1) * 4),
                      s_tile + (size_t)r1 * D + ((gc ^ (r1 & 15)) * 8 + (li & 1) * 4));
            }
          };
          issue_group(0);
#pragma unroll
          for (int gi = 0; gi < NG; ++gi) {
            if (gi + 1 < NG) issue_group(gi + 1);
            const int n_after = gi + 1 < NG ? 2 * (min((gi + 2) * GB, NBW) - (gi + 1) * GB) : 0;
#pragma unroll
            for (int u = gi * GB; u < gi * GB + GB && u < NBW; ++u) {
              if (u == gi * GB) lgkm_fence(n_after, xr[u][0], xr[u][1]); else reg_fence(xr[u][0], xr[u][1]);
              const f16x8_t xa =
                  __builtin_bit_cast(f16x8_t, make_uint4(xr[u][0].x, xr[u][0].y, xr[u][1].x, xr[u][1].y));
              acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa, bh, acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = gi * GB; u < gi * GB + GB && u < NBW; ++u) {     // lo terms after the group's hi terms: no
              const f16x8_t xa =                                           // back-to-back dependent MFMAs
                  __builtin_bit_cast(f16x8_t, make_uint4(xr[u][0].x, xr[u][0].y, xr[u][1].x, xr[u][1].y));
              acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa, bl, acc[u], 0, 0, 0);
            }
          }
          const int nlast = min(nruns, q0 + 16);
          const bool doit = (qs < nlast && qs != nruns - 1) ||   // every run of this pass but the tile's last (stays open)
                            (flush_old && sl == base);
          flush_old = false;
          PROF_MARK(2)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cell)::"memory");
          flush_rows(doit, cell);
          PROF_MARK(3)
        }
        base = (start + nruns - 1) & 15;
        n_heads += __builtin_popcount(hbu);
        m_run = m_last;
      }
    }
    flush_rows(sl == base, lds_ld_b32(s_necell + n_heads - 1));       // the last cell of the chunk
  }
#ifdef GRIDMM_AGG_PROF
  if (blockIdx.x == 3 && blockIdx.y == 5 && lane == 0) {
    const long long te = PROF_T();
    long long* o = g_prof[wave];
    o[0] = p_m[0]; o[1] = p_m[1]; o[2] = p_wait; o[3] = p_dma; o[4] = p_work + (te - pta); o[5] = p_a; o[6] = p_m[4]; o[7] = ntiles;
  }
#endif
}

}  // namespace

#ifdef GRIDMM_AGG_PROF
extern "C" int gridmm_debug_agg_prof(long long* out) {      // development aid (-DGRIDMM_AGG_PROF builds only)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

// Returns GRIDMM_EINVAL when the shape is outside this variant's range (the caller then uses the generic kernel).
int gridmm_grid_aggregate_pipe(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* cells, uint8_t* occ, float* relevance, const int32_t* chunks, int B, int cap, int D,
                               int L, int n_chunks, hipStream_t st) {
  const int Lt = (L + 15) / 16;
  // D = 768 (KS = 24) does not fit: 192 VGPRs of resident text fragments + the MFMA working set spill (58 VGPRs at the
  // 256-register budget of 2 waves per SIMD), and a 3 x 48 KB ring leaves one tile of latency cover.
  if (D != 512 && D != 256) return GRIDMM_EINVAL;
  const int nbw = 8 - Lt;                                    // B-waves; each owns ceil(D / 16 / nbw) 16-dim blocks
  if (Lt < 3 || nbw < 1 || (D == 512 && nbw < 2)) return GRIDMM_EINVAL;   // Lt >= 3: at most MAXR rows per R-wave
  constexpr int R = 4;
  const size_t hb_words = (size_t)(cap + PT - 1) / PT;       // run-head bitmask of (at most) a whole episode
  const size_t lds = (size_t)R * PT * D * 2 + 2 * 8 * PT * sizeof(float) + 200 * sizeof(int) +
                     8 * TAB_BYTES + 8 * 4 * MAXR * sizeof(int) + 200 * sizeof(int) + hb_words * sizeof(unsigned);
  if (lds > 160 * 1024) return GRIDMM_EINVAL;                // D = 512: up to ~185k points per episode
  dim3 grid(n_chunks, B), block(512);
#define GRIDMM_AGGP(KS, RR, NBW)                                                                                     \
  do {                                                                                                               \
    auto kern = grid_aggregate_pipe_kernel<KS, RR, NBW>;                                                             \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,         \
                            (int)lds) != hipSuccess)                                                                 \
      return GRIDMM_EINVAL;                                                                                          \
    GRIDMM_LAUNCH(kern, grid, block, lds, st, (const _Float16*)slab, perm, cell_start, (const _Float16*)text_frag,   \
                  cells, occ, relevance, chunks, cap, L, Lt, n_chunks);                                              \
  } while (0)
  if (D == 512) {
    if (nbw >= 4) GRIDMM_AGGP(16, 4, 8); else if (nbw == 3) GRIDMM_AGGP(16, 4, 11); else GRIDMM_AGGP(16, 4, 16);
  } else {
    if (nbw >= 2) GRIDMM_AGGP(8, 4, 8); else GRIDMM_AGGP(8, 4, 16);
  }
#undef GRIDMM_AGGP
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
