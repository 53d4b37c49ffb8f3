// Shared by the aggregation kernels (aggregate_pipe.hip, aggregate_wide.hip): the per-tile softmax + accumulation
// stage on one wave, and the asm LDS helpers it needs.
//
// A tile is 32 points of the cell-sorted order (lane = point).  Given each point's relevance w and the run-head bits of
// the tile, the wave computes the per-cell softmax numerators and accumulates out^T[dim][slot] = X^T[dim][point] .
// E[point][slot] on the matrix pipe: E holds the numerator of a point in the column of its cell's slot (f16 hi + lo,
// exact to ~2^-22), X^T comes straight out of the row-major LDS tile through ds_read_b64_tr_b16.  A cell keeps its slot
// (= MFMA output column, one per lane & 15) from tile to tile, so the open cell's partial sum never moves between
// lanes; a ones block yields the softmax denominators in the same layout (vilmodel.py:797-807).
#pragma once
#include "common.h"

namespace gridmm_agg {

typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
constexpr int PT = 32;            // points per tile
constexpr float NEG_BIG = -3.0e38f;
constexpr int TAB_BYTES = 320;    // per-wave tables: 32 x (f16 hi, f16 lo, u16 run)

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS loads of waves with global stores in flight go through asm: when the destination VGPRs of a compiler-visible LDS
// load were operands of a still pending global store or LDS-DMA, the waitcnt pass answers with s_waitcnt vmcnt(0), which
// drains the tile ring in the middle of every iteration (measured: ~2000 of 6000 cycles per tile).  The asm forms carry
// their own lgkmcnt.
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

// One wave's share of the accumulation: the 16-dim blocks bw, bw + nbw, ... (at most NBW of them) of every cell.
// Every accumulating wave of a workgroup runs tile() on every tile with the same inputs, so their scalars agree.
template <int D, int NBW, int GB = 4>   // GB: 16-dim blocks per transpose-read group (two groups of registers in flight)
struct CellAccumulator {
  static constexpr int NBLK = D / 16;
  float* cells_b;                   // [196][D] of this episode
  uint8_t* occ_b;                   // [196]
  const int* s_necell;              // LDS: non-empty cells of the chunk, in order
  _Float16* t_ehi;                  // LDS tables of this wave: [32] numerators hi (fragment order) ...
  _Float16* t_elo;                  //                          [32] lo
  unsigned short* t_q;              //                          [32] run index of the point
  int bw, nbw, lane, sl, g;
  int base, n_heads;                // slot of the open cell; run heads before the current tile
  float m_run;
  f32x4_t acc[NBW], acc_s;
  // Chunks are cut at POINT positions (equal work), so the first / last run of a chunk may be a piece of a cell that
  // other chunks continue: such a piece is stored un-normalised as a record (N[D], S, m) in the workspace and the
  // pieces are combined, in chunk order, by grid_aggregate_merge_kernel.  rec: this chunk's [2][D + 4] floats.
  float* rec;
  bool head_partial, tail_partial;  // the chunk's first run started in an earlier chunk / its last run continues

  __device__ __forceinline__ void init(float* cells_b_, uint8_t* occ_b_, const int* s_necell_, unsigned char* tab,
                                       int bw_, int nbw_, int lane_) {
    cells_b = cells_b_; occ_b = occ_b_; s_necell = s_necell_;
    t_ehi = reinterpret_cast<_Float16*>(tab);
    t_elo = reinterpret_cast<_Float16*>(tab + 64);
    t_q = reinterpret_cast<unsigned short*>(tab + 128);
    bw = bw_; nbw = nbw_; lane = lane_; sl = lane_ & 15; g = lane_ >> 4;
    base = 0; n_heads = 0; m_run = NEG_BIG;
    rec = nullptr; head_partial = tail_partial = false;
    acc_s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NBW; ++u) acc[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  // normalise + store + clear the rows of the lanes with doit; lanes with part (a piece of a split cell; implies doit)
  // store the raw sums, the denominator and the piece's maximum mval into record `which` instead
  __device__ __forceinline__ void flush_rows(bool doit, int cell, bool part = false, int which = 0, float mval = 0.f) {
    const float inv = __builtin_amdgcn_rcpf(acc_s[0]);
    if (doit && !part) {                         // one exec region for all stores (the block guard is wave-uniform)
      float* dst = cells_b + (size_t)cell * D + bw * 16 + g * 4;
#pragma unroll
      for (int u = 0; u < NBW; ++u)
        if (bw + u * nbw < NBLK)
          *reinterpret_cast<float4*>(dst + u * nbw * 16) =
              make_float4(acc[u][0] * inv, acc[u][1] * inv, acc[u][2] * inv, acc[u][3] * inv);
      if (g == 0 && bw == 0) occ_b[cell] = 1;
    }
    if (part) {
      float* dst = rec + (size_t)which * (D + 4) + bw * 16 + g * 4;
#pragma unroll
      for (int u = 0; u < NBW; ++u)
        if (bw + u * nbw < NBLK)
          *reinterpret_cast<float4*>(dst + u * nbw * 16) = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
      if (g == 0 && bw == 0) {
        float* tail = rec + (size_t)which * (D + 4) + D;
        tail[0] = acc_s[0];
        tail[1] = mval;
      }
    }
#pragma unroll
    for (int u = 0; u < NBW; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[u][j] = doit ? 0.f : acc[u][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_s[j] = doit ? 0.f : acc_s[j];
  }

  // t: tile index inside the chunk; npt: valid points; w: relevance of point lane (lanes < 32); hbu: run-head bits of
  // the tile (wave-uniform); s_tile: the tile in LDS, row r = point r, 16-B chunk c of a row stored at c ^ (r & 15).
  __device__ __forceinline__ void tile(int t, int npt, float w, unsigned hbu, const _Float16* s_tile) {
    const int lp = lane & (PT - 1);                              // lanes >= PT mirror (results unused)
    if (lane >= npt) w = NEG_BIG;
    // A cell is a contiguous run of lanes [rs, re]; the run heads of this tile are one word of s_hbits (bit 0 clear:
    // the first run continues the open cell of the previous tile).  The run maximum at every lane = max(segmented
    // prefix max, segmented suffix max): DPP row shifts (a VALU modifier) + two scalar readlanes for the seam between
    // the 16-lane rows; the ds_bpermute form of the same scans was a ~1000-cycle serial chain per tile.
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
      flush_rows(sl == base, lds_ld_b32(s_necell + n_heads - 1), head_partial && n_heads == 1 && sl == base, 0, m_run);
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
      int cell;                                               // cell of this lane's run: needed after the MFMAs
      asm volatile("ds_read_b32 %0, %1" : "=v"(cell)
                   : "v"((unsigned)(size_t)(s_necell + (flush_old && sl == base ? n_heads - 1
                                                                               : min(kg0 + qs, GRIDMM_CELLS - 1))))
                   : "memory");
      // transpose reads in groups of GB blocks, one group ahead of the MFMAs that consume them
      constexpr int NG = (NBW + GB - 1) / GB;
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
      const bool old_row = flush_old && sl == base;
      const bool doit = (qs < nlast && qs != nruns - 1) ||   // every run of this pass but the tile's last (stays open)
                        old_row;
      // the chunk's first run (non-empty cell 0 of the chunk) closes here: a piece of a split cell when head_partial
      const bool part = head_partial && doit && (old_row ? n_heads == 1 : kg0 + qs == 0);
      const float mval = old_row ? m_run : m0;
      flush_old = false;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cell)::"memory");
      flush_rows(doit, cell, part, 0, mval);
    }
    base = (start + nruns - 1) & 15;
    n_heads += __builtin_popcount(hbu);
    m_run = m_last;
  }

  __device__ __forceinline__ void finish() {   // the last cell of the chunk (a single-run chunk: also its first)
    const bool part = tail_partial || (head_partial && n_heads == 1);
    flush_rows(sl == base, lds_ld_b32(s_necell + n_heads - 1), part && sl == base, 1, m_run);
  }
};

}  // namespace gridmm_agg
