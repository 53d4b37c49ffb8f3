// Instruction-relevance grid aggregation: ONE pass over the fp16 feature slab.
//
//   w_j    = max_l <x_j, text_l>            all L text columns, padded tokens included
//                                           (map_nav_src/models/vilmodel.py:798)
//   out[c] = sum_{j in cell c} softmax_j(w_j) x_j      (vilmodel.py:801-807; grid_proj is
//            applied to the 196 reduced vectors afterwards: W sum_j a_j x_j + b, sum_j a_j = 1)
//
// Work decomposition: the points of an episode arrive sorted by cell (gridmm_grid_bin's `perm`);
// a workgroup owns a contiguous, cell-aligned chunk of that order, so no cross-workgroup merge
// is needed and every slab row is read from HBM exactly once (1 KB contiguous per row at
// D = 512).  Per 32-point tile:
//   1. rows -> LDS (row-major, 16-B chunk index XOR (row & 15): conflict-free ds_read_b128)
//   2. relevance on MFMA f16 16x16x32: wave t owns text columns [16t, 16t+16) for the whole
//      launch, their fp16 hi+lo fragments (22 significant bits ~ fp32) live in registers;
//      A fragments come from the LDS tile; per-point max over columns by DPP shuffles, then
//      across waves through LDS
//   3. online-softmax accumulation (fp32) of the LDS-resident rows into the running cell vector
//      held in registers (2 feature dims per thread); flush on cell change.
#include "common.h"

namespace {

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
constexpr int TILE_SMALL = 32, TILE_BIG = 64;   // points per ring slot: 64 when two slots fit the LDS (D <= 512)
constexpr float NEG_BIG = -3.0e38f;

__global__ void text_fragments_kernel(const float* __restrict__ text, _Float16* __restrict__ frag, int B,
                                      int L, int D, int Lt) {
  // frag[b][plane][ct][ks][lane][e]
  const int KS = D / 32;
  const size_t per_b = (size_t)2 * Lt * KS * 64 * 8;
  const size_t total = (size_t)B * Lt * KS * 64 * 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int e = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int ks = r % KS; r /= KS;
    const int ct = r % Lt; r /= Lt;
    const int b = (int)r;
    const int col = ct * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + e;
    const float x = col < L ? text[((size_t)b * L + col) * D + k] : 0.f;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    const size_t o = (((size_t)ct * KS + ks) * 64 + lane) * 8 + e;
    frag[b * per_b + o] = h;
    frag[b * per_b + (size_t)Lt * KS * 64 * 8 + o] = l;
  }
}

// chunk boundaries in CELL index space: chunk k of episode b covers cells [cb[k], cb[k+1]); cb[k] = first cell whose
// start is >= k * ceil(valid / n_chunks) (cell_start is non-decreasing: a count of the entries below the target)
__global__ __launch_bounds__(256) void build_chunks_kernel(const int32_t* __restrict__ cell_start,
                                                           int32_t* __restrict__ chunks, int n_chunks) {
  const int b = blockIdx.x;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  int32_t* cb = chunks + (size_t)b * (n_chunks + 1);
  const int mine = threadIdx.x <= GRIDMM_CELLS ? cs[threadIdx.x] : 0x7fffffff;
  const int valid = cs[GRIDMM_CELLS];
  const int target = (valid + n_chunks - 1) / n_chunks;
  for (int k = 0; k <= n_chunks; ++k) {
    const long want = (long)k * target;
    const int below = __syncthreads_count(mine < want);
    if (threadIdx.x == 0) cb[k] = k == 0 ? 0 : (k == n_chunks ? GRIDMM_CELLS : min(below, GRIDMM_CELLS));
  }
}

// RESIDENT: wave t keeps text column tile t in registers (Lt <= 8 waves, <= 256 VGPRs).
// !RESIDENT (L > 128): 8 waves, each loops over column tiles t, t+8, ... and re-streams the
// fragments from L2 per tile -- correct for any L <= 256, slower.
//
// Pipeline: an R-slot LDS ring of 32-point tiles filled by LDS-DMA (one 1-KiB global_load_lds per
// 512-D row, XOR swizzle on the per-lane SOURCE chunk; row ids come from SCALAR loads of `perm`, so no
// vector-memory load ever queues behind the DMA bursts).  Tile t+R-1 is issued right after the barrier
// that retires tile t-1; a COUNTED s_waitcnt vmcnt keeps R-2 younger tiles (64-96 KB per CU) in flight
// across every barrier -- that is what it takes to cover the HBM latency of a 1-KiB-row gather.
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void wait_vm_dyn(int n) {  // n = (R-2) * rows-per-wave: a handful of values
  switch (n) {
    case 8: wait_vm<8>(); break;    case 10: wait_vm<10>(); break;  case 12: wait_vm<12>(); break;
    case 14: wait_vm<14>(); break;  case 16: wait_vm<16>(); break;  case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;    case 6: wait_vm<6>(); break;    case 7: wait_vm<7>(); break;
    default: wait_vm<0>(); break;
  }
}

template <int KS, bool RESIDENT, int R, int TILE>  // D = 32 * KS, R ring slots of TILE points
__global__ __launch_bounds__(512) void grid_aggregate_kernel(
    const _Float16* __restrict__ slab, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ cell_start, const _Float16* __restrict__ text_frag,
    float* __restrict__ cells, uint8_t* __restrict__ occ, float* __restrict__ relevance,
    const int32_t* __restrict__ chunks, int cap, int L, int Lt, int n_chunks) {
  constexpr int D = 32 * KS;
  constexpr int NCH = D / 8;                 // 16-B chunks per row
  constexpr int NACC = (D / 2 + 511) / 512;  // feature-dim pairs per thread (the block is always 8 waves = 512 threads)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* s_tiles = reinterpret_cast<_Float16*>(smem);                       // [R][TILE][D]
  float* s_wmax = reinterpret_cast<float*>(smem + (size_t)R * TILE * D * 2);   // [Lt][TILE]
  float* s_w = s_wmax + (size_t)Lt * TILE;                                     // [TILE]
  float* s_e = s_w + TILE;                                                     // [TILE]
  int* s_cell = reinterpret_cast<int*>(s_e + TILE);                            // [TILE]
  float* s_state = reinterpret_cast<float*>(s_cell + TILE);                    // [0]=scale [1]=m_run [2]=heads
  int* s_cs = reinterpret_cast<int*>(s_state + 4);                             // [198] cell_start of this episode

  const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = nthreads >> 6;
  const int b = blockIdx.y, k = blockIdx.x;
  const int32_t* cs = cell_start + (size_t)b * (GRIDMM_CELLS + 2);
  const int c_lo = chunks[(size_t)b * (n_chunks + 1) + k], c_hi = chunks[(size_t)b * (n_chunks + 1) + k + 1];
  if (c_lo >= c_hi) return;
  const int p_lo = cs[c_lo], p_hi = cs[c_hi];
  float* cells_b = cells + (size_t)b * GRIDMM_CELLS * D;
  uint8_t* occ_b = occ + (size_t)b * GRIDMM_CELLS;

  // empty cells of this chunk: zero vector, occ = 0 (vilmodel.py:803-807)
  for (int c = c_lo + wave; c < c_hi; c += nwaves) {
    if (cs[c + 1] == cs[c]) {
      for (int d = lane; d < D / 4; d += 64)
        reinterpret_cast<float4*>(cells_b + (size_t)c * D)[d] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane == 0) occ_b[c] = 0;
    }
  }
  if (p_lo >= p_hi) return;
  for (int i = tid; i < GRIDMM_CELLS + 2; i += nthreads) s_cs[i] = cs[i];  // binary searches run on LDS

  // this wave's text columns, register-resident for the whole chunk
  const size_t plane = (size_t)Lt * KS * 64 * 8;
  const _Float16* tf_b = text_frag + (size_t)b * 2 * plane + (size_t)lane * 8;
  // RESIDENT (8 waves, Lt <= 8 column tiles): which (column tile, 32-point passes) this wave computes.  Waves w and
  // w + 4 share a SIMD, so column tiles 0..3 go to waves 0..3 and the remaining Lt - 4 tiles are spread over waves
  // 4..7 -- split by passes when there are fewer tiles than waves -- to level the MFMA work per SIMD (L = 80: 5 tiles
  // on 5 waves put 4 passes on SIMD 0 and 2 on the others; this assignment gives 3 / 3 / 2 / 2).
  constexpr int NP = TILE / 32;
  int my_ct = -1, hp_lo = 0, hp_hi = 0;
  if (RESIDENT) {
    if (wave < 4) {
      if (wave < Lt) { my_ct = wave; hp_hi = NP; }
    } else {
      const int extra = Lt - 4, idx = wave - 4;
      if (extra == 1) {
        if (idx < NP) { my_ct = 4; hp_lo = idx; hp_hi = idx + 1; }
      } else if (extra == 2) {
        if (NP == 2 || (idx & 1) == 0) { my_ct = 4 + (idx >> 1); hp_lo = NP == 2 ? (idx & 1) : 0; hp_hi = hp_lo + 1; }
      } else if (extra > 2 && idx < extra) {
        my_ct = 4 + idx; hp_hi = NP;
      }
    }
  }
  f16x8_t thi[RESIDENT ? KS : 1], tlo[RESIDENT ? KS : 1];
  if (RESIDENT && my_ct >= 0) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      thi[RESIDENT ? ks : 0] = *reinterpret_cast<const f16x8_t*>(tf_b + ((size_t)my_ct * KS + ks) * 64 * 8);
      tlo[RESIDENT ? ks : 0] = *reinterpret_cast<const f16x8_t*>(tf_b + plane + ((size_t)my_ct * KS + ks) * 64 * 8);
    }
  }

  const _Float16* slab_b = slab + (size_t)b * cap * D;
  const int32_t* perm_b = perm + (size_t)b * cap;

  int cur = -1;            // cell being accumulated
  float m_run = NEG_BIG, s_run = 0.f;
  float v0[NACC], v1[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) v0[a] = v1[a] = 0.f;

  auto flush = [&]() {
    if (cur < 0) return;
    const float inv = 1.0f / s_run;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int dp = tid + a * nthreads;
      if (dp < D / 2) reinterpret_cast<float2*>(cells_b + (size_t)cur * D)[dp] = make_float2(v0[a] * inv, v1[a] * inv);
    }
    if (tid == 0) occ_b[cur] = 1;
  };

  const int ntiles = (p_hi - p_lo + TILE - 1) / TILE;
  const int RW = (TILE + nwaves - 1) / nwaves;  // DMA instructions per wave per tile (x NCH/64), constant
  constexpr int IPR = (NCH + 63) / 64;           // DMA instructions per row
  // LDS-DMA of tile t: wave w moves rows w, w+nwaves, ... (always RW of them: short waves / short tiles
  // repeat a valid row, so the in-order vmcnt bookkeeping is the same for every wave and tile).
  // Position c of row r holds global chunk c ^ (r & 15).
  // Row ids of a tile: ONE vector load per wave (lane l <- perm[p0 + l]), issued a whole iteration before the DMA that
  // consumes it and BEFORE that iteration's DMA batch (so the counted vmcnt at the next loop top, which lets the
  // younger DMA batches fly, already covers it); each row's address is then a v_readlane away.  Per-row scalar loads
  // put an L2 round trip (~0.4 us) in front of every one of the ~13 DMA instructions a wave issues per tile: measured
  // 5 us per tile of pure skeleton time with compute and DMA both ablated.
  auto load_ids = [&](int t) -> int {
    int p = p_lo + t * TILE + (lane & (TILE - 1));
    if (p >= p_hi) p = p_hi - 1;
    return perm_b[p];
  };
  auto dma_tile = [&](int t, int idv) {
    _Float16* dst = s_tiles + (size_t)(t % R) * TILE * D;
    for (int j = 0; j < RW; ++j) {
      int r = wave + j * nwaves;
      if (r >= TILE) r = wave;
      const int src = __builtin_amdgcn_readlane(idv, r);     // rows past the end repeat the last valid row (load_ids)
      const _Float16* row = slab_b + (size_t)src * D;
#pragma unroll
      for (int c0 = 0; c0 < NCH; c0 += 64) {
        const int c = c0 + lane;
        if (c < NCH)   // D = 768: the second 1-KiB piece of a row is half masked (still one vmcnt event)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(row + (size_t)(c ^ (r & 15)) * 8),
              (__attribute__((address_space(3))) void*)(dst + (size_t)r * D + (size_t)c0 * 8), 16, 0, 0);
      }
    }
  };
  auto lds_barrier = [&]() {  // LDS-visibility barrier that leaves the DMA (vmcnt) in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // Retire every ordinary vector load (text fragments, cell_start) HERE, with a waitcnt the compiler can see: otherwise
  // its scoreboard carries them into the loop and guards the first fragment use of every iteration with
  // s_waitcnt vmcnt(0) -- which also waits for the LDS-DMA of the NEXT tile, i.e. serialises the stream with the compute.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();                                  // s_cs visible
  for (int t = 0; t < R - 1 && t < ntiles; ++t) dma_tile(t, load_ids(t));
  const int keep = (R - 2) * RW * IPR;              // DMA instructions allowed to stay in flight
  int idv = (R - 1 < ntiles) ? load_ids(R - 1) : 0; // row ids of the next tile to issue

  for (int t = 0; t < ntiles; ++t) {
    const int p0 = p_lo + t * TILE;
    const int npt = min(TILE, p_hi - p0);
    const _Float16* s_tile = s_tiles + (size_t)(t % R) * TILE * D;
    // ---- 1. retire tile t (counted wait: the R-2 younger tiles stay in flight), publish it, issue tile t+R-1
    if (t + R - 2 < ntiles) wait_vm_dyn(keep); else wait_vm<0>();
    lds_barrier();
    if (t + R - 1 < ntiles) {                       // its slot held tile t-1: free since the barrier above
      const int idn = (t + R < ntiles) ? load_ids(t + R) : 0;   // older than the DMA batch below in the vmcnt queue
      dma_tile(t + R - 1, idv);
      idv = idn;
    }
    if (wave == nwaves - 1) {                       // cell of each point of this tile (binary search on LDS), by the
      int cell = -1;                                // LAST wave: it has no (or the least) relevance work below, so the
      const int p = p0 + lane;                      // ~8 dependent LDS round trips delay nobody on the way to the barrier
      if (lane < TILE && p < p_hi) {
        int lo = c_lo, hi = c_hi;  // invariant cs[lo] <= p < cs[hi]; the last c with cs[c] <= p owns p
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (s_cs[mid] <= p) lo = mid; else hi = mid;
        }
        cell = lo;
      }
      if (lane < TILE) s_cell[lane] = cell;
    }
    // ---- 2. relevance on the matrix pipe
    for (int ct = RESIDENT ? my_ct : wave; ct >= 0 && ct < Lt; ct += RESIDENT ? 1024 : nwaves) {
      const int i = lane & 15, g = lane >> 4;
      const int hp0 = RESIDENT ? hp_lo : 0, hp1 = RESIDENT ? hp_hi : NP;
      f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
      // Both 16-row halves always (rows past npt hold a repeated valid row, masked below): no branch in the k loop, and
      // the A fragments of group q+1 (GK k-steps x 2 halves) are read from LDS while group q is in the matrix pipe.
      constexpr int GK = KS > 16 ? 2 : 4;       // D = 768 keeps 192 VGPRs of text fragments: prefetch shallower
      static_assert(KS % GK == 0, "k-steps per group");
      for (int hp = hp0; hp < hp1; ++hp) {   // 32 points (two 16-row MFMA tiles) per pass
      acc0 = acc1 = acc2 = acc3 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const f16x8_t* row0 = reinterpret_cast<const f16x8_t*>(s_tile + (size_t)(hp * 32 + i) * D);
      const f16x8_t* row1 = reinterpret_cast<const f16x8_t*>(s_tile + (size_t)(hp * 32 + 16 + i) * D);
      f16x8_t fa[2][GK], fb[2][GK];
#pragma unroll
      for (int u = 0; u < GK; ++u) { fa[0][u] = row0[(u * 4 + g) ^ i]; fb[0][u] = row1[(u * 4 + g) ^ i]; }
#pragma unroll
      for (int q = 0; q < KS / GK; ++q) {
        if (q + 1 < KS / GK) {
#pragma unroll
          for (int u = 0; u < GK; ++u) {
            fa[(q + 1) & 1][u] = row0[(((q + 1) * GK + u) * 4 + g) ^ i];
            fb[(q + 1) & 1][u] = row1[(((q + 1) * GK + u) * 4 + g) ^ i];
          }
        }
#pragma unroll
        for (int u = 0; u < GK; ++u) {
          const int ks = q * GK + u;
          f16x8_t bh, bl;
          if (RESIDENT) {
            bh = thi[RESIDENT ? ks : 0];
            bl = tlo[RESIDENT ? ks : 0];
          } else {
            bh = *reinterpret_cast<const f16x8_t*>(tf_b + ((size_t)ct * KS + ks) * 64 * 8);
            bl = *reinterpret_cast<const f16x8_t*>(tf_b + plane + ((size_t)ct * KS + ks) * 64 * 8);
          }
          // text fragment as the A operand: the tile comes out transposed, lane (point = lane & 15, g) holds text
          // columns 4g .. 4g+3, so the max over columns is 3 in-register ops + 2 cross-lane steps (was 16 ds_bpermute
          // round trips per wave and tile with the points along the registers)
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, fa[q & 1][u], acc0, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, fa[q & 1][u], acc2, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, fb[q & 1][u], acc1, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, fb[q & 1][u], acc3, 0, 0, 0);
        }
      }
      float x0 = NEG_BIG, x1 = NEG_BIG;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool colv = (ct * 16 + 4 * g + r) < L;
        x0 = fmaxf(x0, colv ? acc0[r] + acc2[r] : NEG_BIG);
        x1 = fmaxf(x1, colv ? acc1[r] + acc3[r] : NEG_BIG);
      }
      x0 = fmaxf(x0, __shfl_xor(x0, 16, 64)); x1 = fmaxf(x1, __shfl_xor(x1, 16, 64));
      x0 = fmaxf(x0, __shfl_xor(x0, 32, 64)); x1 = fmaxf(x1, __shfl_xor(x1, 32, 64));
      if (g == 0) {
        s_wmax[ct * TILE + hp * 32 + i] = x0;
        s_wmax[ct * TILE + hp * 32 + 16 + i] = x1;
      }
      }
    }
    lds_barrier();
    // ---- 3a. per-point relevance and softmax numerators, computed by EVERY wave for itself (lane = point): the
    //          results stay in registers (broadcast by v_readlane in 3b), so there is no LDS hand-off and no third
    //          barrier.  Points are sorted by cell: a cell is a contiguous run of lanes -> segmented max by shuffles.
    float e_lane, m_last, sc;
    int c_lane;
    unsigned long long heads;
    {
      float w = NEG_BIG;
      if (lane < TILE)
        for (int q = 0; q < Lt; ++q) w = fmaxf(w, s_wmax[q * TILE + lane]);
      // by SORTED position (no perm load here: a vector load inside the loop would put every later LDS access of the
      // iteration behind a compiler-inserted s_waitcnt vmcnt(0), i.e. behind the next tile's DMA)
      if (relevance && wave == 0 && lane < npt) relevance[(size_t)b * cap + p0 + lane] = w;
      const int c = (lane < npt) ? s_cell[lane] : -2 - lane;   // unique sentinel: never joins a run
      if (lane >= npt) w = NEG_BIG;
      float pre = w, suf = w;
#pragma unroll
      for (int o = 1; o < TILE; o <<= 1) {
        const float pu = __shfl_up(pre, o, 64), sd = __shfl_down(suf, o, 64);
        const int cu = __shfl_up(c, o, 64), cd = __shfl_down(c, o, 64);
        if (lane >= o && cu == c) pre = fmaxf(pre, pu);
        if (lane + o < 64 && cd == c) suf = fmaxf(suf, sd);
      }
      float m = fmaxf(pre, suf);
      if (c == cur) m = fmaxf(m, m_run);                       // the run continuing from the previous tile
      const int cprev = __shfl_up(c, 1, 64);
      const bool head = (lane < npt) && (lane == 0 || cprev != c);
      heads = __ballot(head);
      e_lane = (lane < npt) ? expf(w - m) : 0.f;
      c_lane = c;
      const float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
      const int c0 = __builtin_amdgcn_readfirstlane(c);
      sc = (c0 == cur) ? expf(m_run - m0) : 1.0f;              // rescale of the running cell
      m_last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), npt - 1));
    }
    auto e_of = [&](int r) -> float {                          // r is wave-uniform
      return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e_lane), r));
    };
    // ---- 3b. accumulate rows, one contiguous run (cell) at a time; 2 feature dims per thread per NACC slot
    {
      if (cur >= 0 && __builtin_amdgcn_readfirstlane(c_lane) == cur) {
        s_run *= sc;
#pragma unroll
        for (int a = 0; a < NACC; ++a) { v0[a] *= sc; v1[a] *= sc; }
      }
      while (heads) {
        const int r0 = __builtin_ctzll(heads);
        heads &= heads - 1;
        const int r1 = heads ? __builtin_ctzll(heads) : npt;
        const int c = __builtin_amdgcn_readlane(c_lane, r0);
        if (c != cur) {
          flush();
          cur = c; s_run = 0.f;
#pragma unroll
          for (int a = 0; a < NACC; ++a) v0[a] = v1[a] = 0.f;
        }
        int r = r0;
        for (; r + 8 <= r1; r += 8) {          // 8 rows per group: all 16 LDS reads issued before the first FMA
          float e[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) e[u] = e_of(r + u);
#pragma unroll
          for (int a = 0; a < NACC; ++a) {
            const int dp = tid + a * nthreads;
            if (dp < D / 2) {
              f16x2_t h[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int ch = (dp >> 2) ^ ((r + u) & 15);
                h[u] = *reinterpret_cast<const f16x2_t*>(s_tile + (size_t)(r + u) * D + ch * 8 + (dp & 3) * 2);
              }
              float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;      // two accumulation chains per dim
#pragma unroll
              for (int u = 0; u < 8; u += 2) {
                a0 += e[u] * (float)h[u][0];         a1 += e[u] * (float)h[u][1];
                b0 += e[u + 1] * (float)h[u + 1][0]; b1 += e[u + 1] * (float)h[u + 1][1];
              }
              v0[a] += a0 + b0;
              v1[a] += a1 + b1;
            }
          }
          s_run += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
        }
        for (; r + 4 <= r1; r += 4) {          // 4 independent LDS reads in flight per slot
          float e[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) e[u] = e_of(r + u);
#pragma unroll
          for (int a = 0; a < NACC; ++a) {
            const int dp = tid + a * nthreads;
            if (dp < D / 2) {
              f16x2_t h[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int ch = (dp >> 2) ^ ((r + u) & 15);
                h[u] = *reinterpret_cast<const f16x2_t*>(s_tile + (size_t)(r + u) * D + ch * 8 + (dp & 3) * 2);
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                v0[a] += e[u] * (float)h[u][0];
                v1[a] += e[u] * (float)h[u][1];
              }
            }
          }
          s_run += (e[0] + e[1]) + (e[2] + e[3]);
        }
        for (; r < r1; ++r) {
          const float e = e_of(r);
          s_run += e;
#pragma unroll
          for (int a = 0; a < NACC; ++a) {
            const int dp = tid + a * nthreads;
            if (dp < D / 2) {
              const int ch = (dp >> 2) ^ (r & 15);
              const f16x2_t h = *reinterpret_cast<const f16x2_t*>(s_tile + (size_t)r * D + ch * 8 + (dp & 3) * 2);
              v0[a] += e * (float)h[0];
              v1[a] += e * (float)h[1];
            }
          }
        }
      }
      m_run = m_last;
    }
    // (the barrier at the top of the next iteration separates this tile's readers from the next writers)
  }
  flush();
}

}  // namespace

extern "C" int gridmm_text_fragments(const float* text, void* frag, int B, int L, int D,
                                     gridmm_stream_t stream) {
  if (B <= 0 || L <= 0 || D <= 0 || D % 32) return GRIDMM_EINVAL;
  const int Lt = (L + 15) / 16;
  const size_t total = (size_t)B * Lt * (D / 32) * 64 * 8;
  unsigned grid = (unsigned)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  GRIDMM_LAUNCH(text_fragments_kernel, dim3(grid), dim3(256), 0, as_stream(stream), text,
                     (_Float16*)frag, B, L, D, Lt);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

// aggregate_pipe.hip: the wave-specialised variant (GRIDMM_EINVAL when the shape is outside its range)
int gridmm_grid_aggregate_pipe(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* cells, uint8_t* occ, float* relevance, int32_t* amax, float* ws, int B, int cap,
                               int D, int L, int n_chunks, hipStream_t st);

// aggregate_rel.hip + aggregate_pipe.hip (PREW): the two-pass D = 768 path (needs the `relevance` buffer as scratch)
int gridmm_grid_relevance_wide(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* relevance, int32_t* amax, int B, int cap, int D, int L, int n_chunks,
                               hipStream_t st);
int gridmm_grid_relevance_gemm(const void* slab, const int32_t* perm, const int32_t* cell_start, const void* text_frag,
                               float* relevance, int32_t* amax, int B, int cap, int D, int L, int n_chunks,
                               hipStream_t st);
int gridmm_grid_aggregate_prew(const void* slab, const int32_t* perm, const int32_t* cell_start, const float* w,
                               float* cells, uint8_t* occ, float* ws, int B, int cap, int D, int n_chunks,
                               hipStream_t st);

static size_t chunk_table_bytes(int B, int n_chunks) {
  return ((size_t)B * (n_chunks + 1) * sizeof(int32_t) + 15) / 16 * 16;
}

extern "C" size_t gridmm_grid_aggregate_workspace(int B, int D, int n_chunks) {
  if (B <= 0 || D <= 0 || n_chunks <= 0) return 0;
  return chunk_table_bytes(B, n_chunks) + (size_t)B * n_chunks * 2 * (D + 4) * sizeof(float);
}

// amax (may be NULL): arg-max instruction token of every point, by sorted position -- the routing of the backward.
// Returns GRIDMM_OK with amax written, 1 when the generic kernel ran (amax untouched), < 0 on error.
extern "C" int gridmm_grid_aggregate_train(const void* slab, const int32_t* perm, const int32_t* cell_start,
                                           const void* text_frag, float* cells, uint8_t* occ, float* relevance,
                                           int32_t* amax, void* workspace, int B, int cap, int D, int L, int n_chunks,
                                           gridmm_stream_t stream) {
  if (B <= 0 || cap <= 0 || L <= 0 || n_chunks <= 0 || n_chunks > GRIDMM_CELLS || !workspace) return GRIDMM_EINVAL;
  // workspace: [B][n_chunks + 1] int32 cell-aligned partition (generic kernel) | [B][n_chunks][2][D + 4] f32 records
  // of split cells (pipelined kernels)
  int32_t* chunks = static_cast<int32_t*>(workspace);
  float* ws = reinterpret_cast<float*>(static_cast<char*>(workspace) + chunk_table_bytes(B, n_chunks));
  const int Lt = (L + 15) / 16;
  if (Lt > 32) return GRIDMM_EINVAL;  // L <= 512, BERT's position table (reference: max_instr_len 200 / 250, max_txt_len 300);
                                      // past 256 tokens the relevance GEMM runs as two launches over token groups
  hipStream_t st = as_stream(stream);
  if (gridmm_grid_aggregate_pipe(slab, perm, cell_start, text_frag, cells, occ, relevance, relevance ? amax : nullptr, ws,
                                 B, cap, D, L, n_chunks, st) == GRIDMM_OK)
    return GRIDMM_OK;     // D <= 512 and L <= 96: two-stage wave-specialised pipeline; otherwise the generic kernel below
  if (D == 768 && relevance && L <= 80 && (size_t)cap <= 45000 &&
      gridmm_grid_relevance_wide(slab, perm, cell_start, text_frag, relevance, amax, B, cap, D, L, n_chunks, st) ==
          GRIDMM_OK) {
    // D = 768, L <= 80: relevance pass (text fragments spread over 8 waves) + accumulation pass on the resident slab
    const int rc = gridmm_grid_aggregate_prew(slab, perm, cell_start, relevance, cells, occ, ws, B, cap, D, n_chunks, st);
    if (rc == GRIDMM_OK) return rc;
  }
  if (relevance && gridmm_grid_relevance_gemm(slab, perm, cell_start, text_frag, relevance, amax, B, cap, D, L, n_chunks,
                                               st) == GRIDMM_OK) {
    // long instructions (L > 96; any L at D = 768 with a deep memory): relevance as a GEMM over 128-point tiles
    // (aggregate_relg.hip) + the accumulation pass
    const int rc = gridmm_grid_aggregate_prew(slab, perm, cell_start, relevance, cells, occ, ws, B, cap, D, n_chunks, st);
    if (rc == GRIDMM_OK) return rc;
  }
  GRIDMM_LAUNCH(build_chunks_kernel, dim3(B), dim3(256), 0, st, cell_start, chunks, n_chunks);
  const bool resident = Lt <= 8 && D != 768;   // D = 768: 192 VGPRs of resident fragments spill (180 B/lane); streaming them from L2 measured 172 vs 185 us
  const int nwaves = 8;   // 2 per SIMD; the relevance work is levelled over them inside the kernel
  dim3 grid(n_chunks, B), block(nwaves * 64);
  // ring: 2 x 64 points (D <= 512: 2 x 64 KB) or 3 x 32 points (D = 768: 3 x 48 KB)
  const int TILE = D <= 512 ? TILE_BIG : TILE_SMALL, R = D <= 512 ? 2 : 3;
  const size_t lds = (size_t)R * TILE * D * 2 + ((size_t)Lt * TILE + 2 * TILE) * sizeof(float) +
                     TILE * sizeof(int) + 4 * sizeof(float) + 200 * sizeof(int);
#define GRIDMM_AGG_ONE(KS, RES)                                                                                   \
  do {                                                                                                            \
    auto kern = grid_aggregate_kernel<KS, RES, (KS <= 16 ? 2 : 3), (KS <= 16 ? TILE_BIG : TILE_SMALL)>;                                               \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                            (int)lds) != hipSuccess)                                                              \
      return GRIDMM_EINVAL;                                                                                       \
    GRIDMM_LAUNCH(kern, grid, block, lds, st, (const _Float16*)slab, perm, cell_start,                            \
                  (const _Float16*)text_frag, cells, occ, relevance, chunks, cap, L, Lt, n_chunks);               \
  } while (0)
#define GRIDMM_AGG(KS)                 \
  do {                                 \
    if (resident) GRIDMM_AGG_ONE(KS, true); \
    else GRIDMM_AGG_ONE(KS, false);    \
  } while (0)
  switch (D) {
    case 256: GRIDMM_AGG(8); break;
    case 512: GRIDMM_AGG(16); break;
    case 768: GRIDMM_AGG(24); break;
    default: return GRIDMM_EINVAL;
  }
#undef GRIDMM_AGG
#undef GRIDMM_AGG_ONE
  GRIDMM_CHECK_LAUNCH();
  return amax ? 1 : GRIDMM_OK;
}

extern "C" int gridmm_grid_aggregate(const void* slab, const int32_t* perm, const int32_t* cell_start,
                                     const void* text_frag, float* cells, uint8_t* occ, float* relevance,
                                     void* workspace, int B, int cap, int D, int L, int n_chunks,
                                     gridmm_stream_t stream) {
  const int rc = gridmm_grid_aggregate_train(slab, perm, cell_start, text_frag, cells, occ, relevance, nullptr, workspace, B,
                                             cap, D, L, n_chunks, stream);
  return rc > 0 ? GRIDMM_OK : rc;
}
