// Instruction-relevance grid aggregation, wave-specialised two-stage pipeline (the hot variant of aggregate.hip for
// D = 256 / 512 and 33 <= L <= 96 instruction tokens; same math, same outputs, same entry point).
//
// aggregate.hip runs the three phases of a tile one after the other on all waves (relevance MFMAs | per-point softmax
// numerators | accumulation), each a latency-bound chain on a few waves: ~10 us per 64 points and CU, 1.7-1.9 TB/s.
// Here the 8 waves of a workgroup split into
//   R-waves (one per 16-column text tile, fragments register-resident): relevance of tile i -> s_wmax[i & 1]; they also
//            feed the ring: rows of tile i + 2 by LDS-DMA, issued in slices between the MFMA groups
//   B-waves (the rest):  softmax numerators + accumulation of tile i - 1 (s_wmax[(i-1) & 1]; agg_accum.h)
// with ONE barrier per 32-point tile, so a tile costs max(relevance, softmax + accumulation) instead of their sum, and
// the LDS-DMA of tiles i+1 .. i+R-2 flies over both.  Ring: R slots of 32 points (4 x 32 KB at D <= 512: slot of tile
// i-1 being accumulated, slot of tile i in the matrix pipe, two tiles in flight).
// The accumulation is a matrix product as well (X^T . E through ds_read_b64_tr_b16; agg_accum.h).  The run heads of a
// tile are one word of a per-chunk bitmask built from cell_start in the prologue, so there is no per-point cell lookup.
// Row ids travel by LDS-DMA two iterations ahead of their use (a scalar load would put its memory latency into every
// lgkmcnt wait, a vector load's result register makes the compiler drain the queue): nothing but DMA in the R-waves'
// vector-memory queue, so the counted s_waitcnt vmcnt is exact; the B-waves' LDS loads are asm (agg_accum.h) because a
// visible LDS load into a register that a pending global store still names gets an s_waitcnt vmcnt(0) in front.
// A workgroup takes an equal share of an episode's sorted points (whole tiles); cells that a cut splits are combined
// from their pieces by grid_aggregate_merge_kernel (see there).
// PREW instantiation: second pass of the D = 768 path (see the template comment below).
#include "agg_accum.h"

namespace {

#ifdef GRIDMM_AGG_PROF
__device__ long long g_prof[8][8];
#define PROF_T() ((long long)__builtin_readcyclecounter())
#define PROF_MARK(k) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t_ = PROF_T(); p_m[k] += t_ - pmk; pmk = t_; }
#else
#define PROF_MARK(k)
#endif

using namespace gridmm_agg;
constexpr int MAXR = 12;          // rows DMA'd per R-wave and tile, at most (ceil(PT / Lt), Lt >= 3)

// Points per chunk: an equal share of the episode's valid points, in whole tiles.
__host__ __device__ __forceinline__ int chunk_points(int valid, int n_chunks) {
  const int share = (valid + n_chunks - 1) / n_chunks;
  return max(PT, (share + PT - 1) / PT * PT);
}

// Combines the pieces of the cells that the point-balanced chunking split (records written by CellAccumulator).
// Workgroup (k, b) acts only if chunk k CLOSES a split cell (its first run started in an earlier chunk and ends here):
// it re-derives from cell_start which earlier chunks hold the other pieces (the opener's tail record and any chunks
// lying entirely inside the cell), then N = sum_i exp(m_i - M) N_i, S likewise, M = max m_i, in chunk order
// (deterministic), and writes the normalised row.  Two dependent memory round trips (cell_start, records).
template <int D>
__global__ __launch_bounds__(128) void grid_aggregate_merge_kernel(const int32_t* __restrict__ cell_start,
                                                                  const float* __restrict__ ws,
                                                                  float* __restrict__ cells, uint8_t* __restrict__ occ,
                                                                  int n_chunks) {
  constexpr int PER = (D / 4 + 127) / 128;     // float4 per thread
  __shared__ int s_cs[GRIDMM_CELLS + 2];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  for (int i = tid; i < GRIDMM_CELLS + 2; i += 128) s_cs[i] = cs[i];
  __syncthreads();
  const int valid = s_cs[GRIDMM_CELLS];
  const int target = chunk_points(valid, n_chunks);
  auto count_below = [&](int pos) {            // entries 0..196 of cell_start that are < pos (non-decreasing)
    int lo = 0, hi = GRIDMM_CELLS + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_cs[mid] < pos) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  struct Meta { int c_lo; bool hp, tp, single; };
  auto meta = [&](int kk) {
    const long lo = (long)kk * target;
    const int p_lo = (int)lo, p_hi = (int)min(lo + target, (long)valid);
    const int c_lo = count_below(p_lo + 1) - 1, c_hi = count_below(p_hi);
    return Meta{c_lo, s_cs[c_lo] < p_lo, s_cs[c_hi] > p_hi, s_cs[c_lo + 1] >= p_hi};
  };
  if ((long)k * target >= valid) return;
  const Meta me = meta(k);
  if (!me.hp || (me.single && me.tp)) return;                  // nothing closes here
  int j = k - 1;                                               // the opener: the last earlier chunk that is not wholly inside the cell
  while (j > 0) {
    const Meta mj = meta(j);
    if (!(mj.single && mj.hp && mj.tp)) break;
    --j;
  }
  auto rec_of = [&](int i) {                                   // a chunk's closing piece is record 0 unless the chunk is one run
    const int which = (i == k && !me.single) ? 0 : 1;
    return ws + (((size_t)b * n_chunks + i) * 2 + which) * (D + 4);
  };
  float M = -3.0e38f;
  for (int i = j; i <= k; ++i) M = fmaxf(M, rec_of(i)[D + 1]);
  float4 n[PER];
  float S = 0.f;
#pragma unroll
  for (int u = 0; u < PER; ++u) n[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = j; i <= k; ++i) {
    const float* r = rec_of(i);
    const float a = expf(r[D + 1] - M);
    S += a * r[D];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int q = tid + u * 128;
      if (q < D / 4) {
        const float4 v = reinterpret_cast<const float4*>(r)[q];
        n[u].x += a * v.x; n[u].y += a * v.y; n[u].z += a * v.z; n[u].w += a * v.w;
      }
    }
  }
  const float inv = 1.0f / S;
  float* dst = cells + ((size_t)b * GRIDMM_CELLS + me.c_lo) * D;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = tid + u * 128;
    if (q < D / 4) reinterpret_cast<float4*>(dst)[q] = make_float4(n[u].x * inv, n[u].y * inv, n[u].z * inv, n[u].w * inv);
  }
  if (tid == 0) occ[(size_t)b * GRIDMM_CELLS + me.c_lo] = 1;
}

// PREW: the relevance of every point is an INPUT (`relevance`, by sorted position; aggregate_rel.hip computed it): the
// R-waves only feed the ring -- the second pass of the D = 768 path, whose text fragments do not fit one wave.
template <int KS, int R, int NBW, bool PREW = false>   // D = 32 * KS; NBW = 16-dim blocks per B-wave (at least ceil(D / 16 / B-waves))
__global__ __launch_bounds__(512) void grid_aggregate_pipe_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm, const int32_t* __restrict__ cell_start,
    const _Float16* __restrict__ text_frag, float* __restrict__ cells, uint8_t* __restrict__ occ,
    float* __restrict__ relevance, int32_t* __restrict__ amax, float* __restrict__ ws, int cap, int L, int Lt,
    int n_chunks) {
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
  float* s_w = reinterpret_cast<float*>(s_necell + 200);                        // [8 tiles][PT] relevance tiles (PREW)
  int* s_warg = reinterpret_cast<int*>(s_w + 8 * PT);                           // [2][PT][8] arg-max token per R-wave (amax)
  unsigned* s_hbits = reinterpret_cast<unsigned*>(s_warg + 2 * 8 * PT);              // [ntiles] bit j of word t: a cell starts at
                                                                                // point 32 t + j of the chunk (run heads)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GRIDMM_AGG_PROF
  const long long pt0 = PROF_T();
  long long p_wait = 0, p_dma = 0, p_work = 0, p_a = 0, pt2 = 0, pta = 0, pmk = 0;
  long long p_m[6] = {0, 0, 0, 0, 0, 0};
#endif
  // XCD-aware (episode, chunk) order.  Workgroups go to the 8 XCDs round-robin by linear id; with chunk = blockIdx.x the
  // chunks of ONE episode land on 8 different XCDs and every XCD's L2 fetches every episode's text fragments (160 KB per
  // workgroup, 41 MB per launch at B = 32 -- the whole prologue).  Here the workgroups of an XCD take a contiguous
  // range of the (episode-major) chunk list, so an episode's fragments, cell_start and perm are fetched by one L2.
  int b, k;
  {
    const int T = gridDim.x * gridDim.y, lid = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lid & 7, j = lid >> 3, q = T >> 3, rem = T & 7;
    const int f = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;     // bijective for any T
    b = f / n_chunks;
    k = f - b * n_chunks;
  }
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  const bool is_r = wave < Lt;                 // relevance wave (text column tile `wave`)
  // The set-up is a lambda called at the top of each role's branch: the R-waves issue their text-fragment loads (128
  // VGPRs) BEFORE it, so that those fly under the cell_start round trip, and the register allocator still never sees the
  // fragments and the B-waves' accumulators live together.  Same barrier sequence in every branch.
  int c_lo = 0, c_hi = 0, p_lo = 0, p_hi = 0, ntiles = 0;
  float* cells_b = cells + (size_t)b * GRIDMM_CELLS * D;
  uint8_t* occ_b = occ + (size_t)b * GRIDMM_CELLS;
  auto setup = [&]() -> bool {
  // Chunk k of the episode = sorted points [p_lo, p_hi): equal shares (whole 32-point tiles) whatever the cell
  // populations -- a crowded cell (thousands of points at depth 15) is split over chunks, its pieces go to the workspace
  // as records and grid_aggregate_merge_kernel combines them.  Cells [c_lo, c_hi) overlap the chunk: cell_start is
  // non-decreasing, so both ends are counts of entries below a position.
  const int mine = tid < GRIDMM_CELLS + 2 ? cs[tid] : 0x7fffffff;
  if (tid < GRIDMM_CELLS + 2) s_cs[tid] = mine;
  const int valid = cs[GRIDMM_CELLS];
  const bool counted = tid <= GRIDMM_CELLS;
  const int target = chunk_points(valid, n_chunks);
  p_lo = (int)min((long)k * target, (long)valid);
  p_hi = (int)min((long)p_lo + target, (long)valid);
  const int n_le_lo = __syncthreads_count(counted && mine <= p_lo);
  const int n_lt_hi = __syncthreads_count(counted && mine < p_hi);
  c_lo = max(n_le_lo - 1, 0);                  // the (non-empty) cell that holds point p_lo
  c_hi = n_lt_hi;                              // one past the cell that holds point p_hi - 1

  // empty cells: zero vector, occ = 0 (vilmodel.py:803-807); an empty cell belongs to the chunk of its start position
  for (int c = wave; c < GRIDMM_CELLS; c += 8) {
    const int st = s_cs[c];
    if (s_cs[c + 1] == st && min(st / target, n_chunks - 1) == k) {
      for (int d = lane; d < D / 4; d += 64)
        reinterpret_cast<float4*>(cells_b + (size_t)c * D)[d] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane == 0) occ_b[c] = 0;
    }
  }
  if (p_lo >= p_hi) return false;
  ntiles = (p_hi - p_lo + PT - 1) / PT;
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
          const int hp = max(st - p_lo, 0);                 // (the chunk's first run starts at its first point)
          atomicOr(&s_hbits[hp >> 5], 1u << (hp & 31));
        }
        kbase += __builtin_popcountll(mk);
      }
    }
  }
  return true;
  };

  const _Float16* slab_b = slab + (size_t)b * cap * D;
  const int32_t* perm_b = perm + (size_t)b * cap;

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
    if (PREW && wave == 0 && lane < PT) {                  // the tile's relevance values travel with its row ids
      int p = p_lo + t * PT + lane;
      if (p >= p_hi) p = p_hi - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(relevance + (size_t)b * cap + p),
                                       (__attribute__((address_space(3))) void*)(s_w + (t & 7) * PT), 4, 0, 0);
    }
  };
  const int ids_instrs = (PREW && wave == 0) ? 2 : 1;      // vector-memory instructions of one load_ids()
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
      case 26: wait_vm<26>(); break;
      case 25: wait_vm<25>(); break;
      case 24: wait_vm<24>(); break;
      case 23: wait_vm<23>(); break;
      case 22: wait_vm<22>(); break;
      case 21: wait_vm<21>(); break;
      case 20: wait_vm<20>(); break;
      case 19: wait_vm<19>(); break;
      case 18: wait_vm<18>(); break;
      case 17: wait_vm<17>(); break;
      case 16: wait_vm<16>(); break;
      case 15: wait_vm<15>(); break;
      case 14: wait_vm<14>(); break;
      case 3: wait_vm<3>(); break;
      case 2: wait_vm<2>(); break;
      case 1: wait_vm<1>(); break;
      default: wait_vm<0>(); break;
    }
  };

  // accumulation (B-waves): B-wave bw owns the 16-dim blocks bw, bw + nbw, ... of every cell (agg_accum.h)
  const int bw = wave - Lt, nbw = 8 - Lt;

  // Queue discipline of an R-wave (in order): iteration i issues DMA(i + R - 2) then IDS(i + R); its top needs DMA(i)
  // (issued at i - R + 2) and IDS(i + R - 2) (issued at i - 2) and leaves the R - 3 younger tiles and IDS(i + R - 1)
  // in flight.
  static_assert(R == 3 || R == 4, "ring of 3 (D = 768) or 4 slots");
  auto iter_head_r = [&](int i) {
#ifdef GRIDMM_AGG_PROF
    pt2 = PROF_T(); if (i > 0) p_work += pt2 - pta; else p_m[1] = pt2 - pt0;
#endif
    // What may stay in flight at the top of iteration i is what was issued AFTER the ids of tile i + R - 2 (dma_prepare
    // reads them right below): the rows of ONE tile and the next id fetch.  (PREW: nobody computes on tile i in iteration
    // i, so with three slots the loaders keep tile i flying; with four slots the round-3 first version kept two tiles,
    // i.e. counted the wait past those ids -- a race that showed at D = 256, whose short iterations outran the id fetch.)
    constexpr int KEEP = PREW ? 1 : R - 3;
    if (i >= 1 && i + R - 1 < ntiles) wait_vm_dyn(KEEP * my_rows * IPR + ids_instrs);   // steady state
    else if (PREW ? i < ntiles : i + 1 < ntiles) wait_vm_dyn(KEEP * my_rows * IPR);      // first / last iterations: tiles only
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
    pt2 = PROF_T(); if (i > 0) p_work += pt2 - pta; else p_m[1] = pt2 - pt0;
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
  if constexpr (PREW) {                         // loader waves of the second pass: ring feed only
    if (!setup()) return;
    __builtin_amdgcn_s_waitcnt(0);
    for (int t = 0; t < R && t < ntiles; ++t) load_ids(t);
    wait_vm<0>();
    __syncthreads();
    for (int t = 0; t < R - 2 && t < ntiles; ++t) { dma_prepare(t); dma_rows(t, 0, MAXR); }
    for (int i = 0; i <= ntiles; ++i) {
      iter_head_r(i);
      if (i + R - 2 < ntiles) {
        dma_rows(i + R - 2, 0, MAXR);
        if (i + R < ntiles) load_ids(i + R);
      }
    }
  } else {
    f16x8_t thi[KS], tlo[KS];
    {
      const size_t plane = (size_t)Lt * KS * 64 * 8;
      const _Float16* tf = text_frag + (size_t)b * 2 * plane + (size_t)lane * 8 + (size_t)wave * KS * 64 * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        thi[ks] = *reinterpret_cast<const f16x8_t*>(tf + (size_t)ks * 64 * 8);
        tlo[ks] = *reinterpret_cast<const f16x8_t*>(tf + plane + (size_t)ks * 64 * 8);
      }
    }
    if (!setup()) return;
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
        if (amax) {   // training: also the arg-max token (first maximum, as torch.max), for the backward's routing
          int i0 = 0x7fffffff, i1 = 0x7fffffff;
          x0 = x1 = NEG_BIG;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int tok = wave * 16 + 4 * g + r;
            const float v0 = tok < L ? acc0[r] + acc2[r] : NEG_BIG, v1 = tok < L ? acc1[r] + acc3[r] : NEG_BIG;
            if (v0 > x0) { x0 = v0; i0 = tok; }
            if (v1 > x1) { x1 = v1; i1 = tok; }
          }
#pragma unroll
          for (int m = 16; m <= 32; m <<= 1) {
            const float y0 = __shfl_xor(x0, m, 64), y1 = __shfl_xor(x1, m, 64);
            const int j0 = __shfl_xor(i0, m, 64), j1 = __shfl_xor(i1, m, 64);
            if (y0 > x0 || (y0 == x0 && j0 < i0)) { x0 = y0; i0 = j0; }
            if (y1 > x1 || (y1 == x1 && j1 < i1)) { x1 = y1; i1 = j1; }
          }
          if (g == 0) {
            int* wa = s_warg + (i & 1) * 8 * PT + wave;
            wa[pi * 8] = i0;
            wa[(16 + pi) * 8] = i1;
          }
        } else {
          x0 = fmaxf(x0, __shfl_xor(x0, 16, 64)); x1 = fmaxf(x1, __shfl_xor(x1, 16, 64));
          x0 = fmaxf(x0, __shfl_xor(x0, 32, 64)); x1 = fmaxf(x1, __shfl_xor(x1, 32, 64));
        }
        if (g == 0) {
          float* wm = s_wmax + (i & 1) * 8 * PT + wave;
          wm[pi * 8] = x0;
          wm[(16 + pi) * 8] = x1;
        }
      }
    }
  }
  } else {
    if (!setup()) return;
    CellAccumulator<D, NBW> cacc;              // (declared in this branch: never live together with the text fragments)
    cacc.init(cells_b, occ_b, s_necell, s_tab + (size_t)bw * TAB_BYTES, bw, nbw, lane);
    cacc.rec = ws + ((size_t)b * n_chunks + k) * 2 * (D + 4);
    cacc.head_partial = s_cs[c_lo] < p_lo;
    cacc.tail_partial = s_cs[c_hi] > p_hi;
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
        if constexpr (PREW) {
          const unsigned a_w = (unsigned)(size_t)(s_w + (t & 7) * PT + lp);
          const unsigned a_h = (unsigned)(size_t)(s_hbits + t);
          asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(w), "=&v"(hb) : "v"(a_w), "v"(a_h) : "memory");
        } else {
          const unsigned a_w = (unsigned)(size_t)(s_wmax + ((t & 1) * PT + lp) * 8);
          const unsigned a_h = (unsigned)(size_t)(s_hbits + t);
          float4 w0, w1;
          asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b32 %2, %4\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(w0), "=&v"(w1), "=&v"(hb) : "v"(a_w), "v"(a_h) : "memory");
          w = fmaxf(fmaxf(fmaxf(w0.x, w0.y), fmaxf(w0.z, w0.w)), fmaxf(fmaxf(w1.x, w1.y), fmaxf(w1.z, w1.w)));
          if (relevance && wave == 7 && lane < npt) relevance[(size_t)b * cap + p0 + lane] = w;   // by sorted position
          if (amax && wave == 7) {              // arg-max token: first column (R-wave = token tile) that attains w
            const unsigned a_a = (unsigned)(size_t)(s_warg + ((t & 1) * PT + lp) * 8);
            int4 g0, g1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(g0), "=&v"(g1) : "v"(a_a) : "memory");
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
        cacc.tile(t, npt, w, (unsigned)__builtin_amdgcn_readfirstlane((int)hb), s_tile);
      }
    }
    cacc.finish();
  }
#ifdef GRIDMM_AGG_PROF
  if (blockIdx.x == 3 && blockIdx.y == 5 && lane == 0) {
    const long long te = PROF_T();
    long long* o = g_prof[wave];
    o[0] = te - pt0; o[1] = p_m[1]; o[2] = p_wait; o[3] = p_dma; o[4] = p_work + (te - pta); o[5] = p_a; o[6] = p_m[4]; o[7] = ntiles;
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
                               float* cells, uint8_t* occ, float* relevance, int32_t* amax, float* ws, int B, int cap,
                               int D, int L, int n_chunks, hipStream_t st) {
  const int Lt = (L + 15) / 16;
  // D = 768 (KS = 24) does not fit: 192 VGPRs of resident text fragments + the MFMA working set spill (58 VGPRs at the
  // 256-register budget of 2 waves per SIMD), and a 3 x 48 KB ring leaves one tile of latency cover.
  if (D != 512 && D != 256) return GRIDMM_EINVAL;
  const int nbw = 8 - Lt;                                    // B-waves; each owns ceil(D / 16 / nbw) 16-dim blocks
  if (Lt < 3 || nbw < 1 || (D == 512 && nbw < 2)) return GRIDMM_EINVAL;   // Lt >= 3: at most MAXR rows per R-wave
  constexpr int R = 4;
  const size_t hb_words = (size_t)(cap + PT - 1) / PT;       // run-head bitmask of (at most) a whole episode
  const size_t lds = (size_t)R * PT * D * 2 + 2 * 8 * PT * sizeof(float) + 200 * sizeof(int) +
                     8 * TAB_BYTES + 8 * 4 * MAXR * sizeof(int) + 200 * sizeof(int) + 8 * PT * sizeof(float) +
                     2 * 8 * PT * sizeof(int) + hb_words * sizeof(unsigned);
  if (lds > 160 * 1024) return GRIDMM_EINVAL;                // D = 512: up to ~185k points per episode
  dim3 grid(n_chunks, B), block(512);
#define GRIDMM_AGGP(KS, RR, NBW)                                                                                     \
  do {                                                                                                               \
    auto kern = grid_aggregate_pipe_kernel<KS, RR, NBW>;                                                             \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,         \
                            (int)lds) != hipSuccess)                                                                 \
      return GRIDMM_EINVAL;                                                                                          \
    GRIDMM_LAUNCH(kern, grid, block, lds, st, (const _Float16*)slab, perm, cell_start, (const _Float16*)text_frag,   \
                  cells, occ, relevance, amax, ws, cap, L, Lt, n_chunks);                                            \
  } while (0)
  if (D == 512) {
    if (nbw >= 4) GRIDMM_AGGP(16, 4, 8); else if (nbw == 3) GRIDMM_AGGP(16, 4, 11); else GRIDMM_AGGP(16, 4, 16);
  } else {
    if (nbw >= 2) GRIDMM_AGGP(8, 4, 8); else GRIDMM_AGGP(8, 4, 16);
  }
#undef GRIDMM_AGGP
  if (D == 512) GRIDMM_LAUNCH(grid_aggregate_merge_kernel<512>, dim3(n_chunks, B), dim3(128), 0, st, cell_start, ws, cells, occ, n_chunks);
  else GRIDMM_LAUNCH(grid_aggregate_merge_kernel<256>, dim3(n_chunks, B), dim3(128), 0, st, cell_start, ws, cells, occ, n_chunks);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

// Second pass of the two-pass paths: w (relevance by sorted position, from gridmm_grid_relevance_wide / _gemm) -> cells /
// occ.  3 loader waves + 5 accumulating waves; ring of 3 x 48 KB (D = 768) or 4 x 32 / 16 KB (D = 512 / 256).
int gridmm_grid_aggregate_prew(const void* slab, const int32_t* perm, const int32_t* cell_start, const float* w,
                               float* cells, uint8_t* occ, float* ws, int B, int cap, int D, int n_chunks,
                               hipStream_t st) {
  if (D != 768 && D != 512 && D != 256) return GRIDMM_EINVAL;
  constexpr int LOADERS = 3;
  const int R = D == 768 ? 3 : 4;
  const size_t hb_words = (size_t)(cap + PT - 1) / PT;
  const size_t lds = (size_t)R * PT * D * 2 + 2 * 8 * PT * sizeof(float) + 200 * sizeof(int) + 8 * TAB_BYTES +
                     8 * 4 * MAXR * sizeof(int) + 200 * sizeof(int) + 8 * PT * sizeof(float) +
                     2 * 8 * PT * sizeof(int) + hb_words * sizeof(unsigned);
  if (lds > 160 * 1024) return GRIDMM_EINVAL;                // D = 768: up to ~60k points per episode
#define GRIDMM_PREW(KS, RR, NBW)                                                                                   \
  do {                                                                                                             \
    auto kern = grid_aggregate_pipe_kernel<KS, RR, NBW, true>;                                                     \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                            (int)lds) != hipSuccess)                                                               \
      return GRIDMM_EINVAL;                                                                                        \
    GRIDMM_LAUNCH(kern, dim3(n_chunks, B), dim3(512), lds, st, (const _Float16*)slab, perm, cell_start,            \
                  (const _Float16*)nullptr, cells, occ, const_cast<float*>(w), (int32_t*)nullptr, ws, cap, 0, LOADERS, \
                  n_chunks);                                                                                       \
  } while (0)
  if (D == 768) {
    GRIDMM_PREW(24, 3, 10);
    GRIDMM_LAUNCH(grid_aggregate_merge_kernel<768>, dim3(n_chunks, B), dim3(128), 0, st, cell_start, ws, cells, occ, n_chunks);
  } else if (D == 512) {
    GRIDMM_PREW(16, 4, 7);
    GRIDMM_LAUNCH(grid_aggregate_merge_kernel<512>, dim3(n_chunks, B), dim3(128), 0, st, cell_start, ws, cells, occ, n_chunks);
  } else {
    GRIDMM_PREW(8, 4, 8);     // (NBW = 4 would do for 16 blocks over 5 waves, but the accumulator's single-group form -- one
                              // transpose-read group, NBW <= GB -- gave run-dependent results here: the two-group form is the tested one)
    GRIDMM_LAUNCH(grid_aggregate_merge_kernel<256>, dim3(n_chunks, B), dim3(128), 0, st, cell_start, ws, cells, occ, n_chunks);
  }
#undef GRIDMM_PREW
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
