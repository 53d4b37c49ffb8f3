// gridmm_attention_rows: bf16x3 attention (head_dim 64) with K and V staged ONCE per workgroup in LDS, straight from
// the row-major hi/lo planes the QKV / KV GEMMs emit -- no V re-tiling pass, no per-wave re-reads of K / V from L2.
//
// Replaces gridmm_transpose_v + gridmm_attention_planes on the hot path (reference: BertSelfAttention /
// BertOutAttention, map_nav_src/models/vilmodel.py:317-379, and nn.MultiheadAttention of the grid encoder,
// transformer.py:176-177).  Same arithmetic as attention_planes_kernel (3-term bf16 split for S = K Q^T and for P V,
// fp32 online softmax, masked keys contribute exactly 0), different data path:
//
//   * workgroup = (episode b, head h, block of NW*NQ query tiles); keys are walked in chunks of KC rows; a chunk of
//     the four planes (K hi, K lo, V hi, V lo; KC x 64 bf16 each) is copied global -> LDS by LDS-DMA, 1 KiB (8 key rows
//     x 128 B) per wave-instruction.  The DMA destination is lane-linear, so the bank swizzle is applied on the SOURCE
//     side: slot `pos` of row r holds the 16-byte chunk  pos ^ (r & 6).
//   * S^T tile = K Q^T: the A operand (16 keys x 32 dims) is one ds_read_b128 per lane from that image
//     (conflict-free by the swizzle); lane (j, g) ends with S^T[key 4g+r][query j].
//   * O^T tile = V^T P^T: the A operand (16 dims x 32 key slots) comes out of the SAME row-major image through
//     ds_read_b64_tr_b16 (hardware 4x16 transpose read): group g of 16 lanes reads 4 key rows x 16 dims, lane j gets
//     dim 16n+j of the 4 keys.  Key slot 8g+e of the contraction is key 4g+e (e < 4) / 16+4g+(e-4): exactly the keys
//     whose probabilities lane (j, g) already holds in its two S^T accumulators, so P^T is the B operand as it is.
//     Output lane (j, g) holds O[query j][dims 16n+4g .. +3]: the running-max rescale is lane-local (no broadcast)
//     and the stores are 64/128-bit.
//   * NQ query tiles per wave share every K / V fragment read; wave NW of the workgroup only loads (see the chunk ring).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ void dma16(const unsigned short* gsrc, unsigned short* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int OFF>
__device__ __forceinline__ uint2 lds_tr_b64(unsigned addr) {   // no wait: see tr_fence
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}
// s_waitcnt lgkmcnt(0) that the uses of the eight pairs cannot be scheduled above
__device__ __forceinline__ void tr_fence(uint2 (&a)[4][2], uint2 (&b)[4][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]),
                 "+v"(a[3][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]),
                 "+v"(b[3][0]), "+v"(b[3][1])
               :
               : "memory");
}

#ifdef GRIDMM_ATT_PROF   // development aid: per-phase s_memtime totals over all waves (tools/att_prof.py)
__device__ unsigned long long g_att_prof[8];
#define GRIDMM_T(i) do { const long long t_ = __builtin_readcyclecounter(); prof[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define GRIDMM_T(i) do { } while (0)
#endif

// SEG: 0 = one context buffer; 1 = keys >= S1 in a second buffer, S1 % 8 == 0 (every 8-row DMA piece lies in one buffer: the
// choice is wave-uniform, scalar selects only -- per-lane selects in the loader's address path cost the step 0.4 ms when
// they sat in every call); 2 = any S1 (per-lane selects).
template <int NQ, int NW, int KC, int AB = 0, int NL = 1, int SEG = 0>
__global__ __launch_bounds__((NW + NL) * 64) void attention_rows_kernel(
    const unsigned short* __restrict__ Qh, const unsigned short* __restrict__ Ql, int64_t q_bs, int q_rs,
    const unsigned short* __restrict__ Kh, const unsigned short* __restrict__ Kl, int64_t k_bs, int k_rs,
    const unsigned short* __restrict__ Vh, const unsigned short* __restrict__ Vl, int64_t v_bs, int v_rs,
    const uint8_t* __restrict__ kmask, int mask_bs, float* __restrict__ O, int64_t o_bs, int o_rs,
    unsigned short* __restrict__ Ohi, unsigned short* __restrict__ Olo, int64_t p_bs, int p_rs, int Sq, int Sk,
    float scale, const unsigned short* __restrict__ K2h, const unsigned short* __restrict__ K2l,
    const unsigned short* __restrict__ V2h, const unsigned short* __restrict__ V2l, int64_t kv2_bs, int kv2_rs, int S1) {
  static_assert(KC % 32 == 0 && 2 * 4 * KC * 128 <= 65536, "chunk ring (also the 16-bit ds offset field)");
  constexpr int PLANE = KC * 64;                                  // u16 per plane image
  constexpr int NT = KC / 32;                                     // key tiles per chunk
  __shared__ __attribute__((aligned(16))) unsigned short kvbuf[2 * 4 * PLANE];   // 2 x (K hi | K lo | V hi | V lo)
  __shared__ unsigned s_mw[64];                                   // key validity, one word per 32 keys (Sk <= 2048)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = blockIdx.y, b = blockIdx.z;
  const int j = lane & 15, g = lane >> 4;

  // ---- per-lane LDS offsets (bytes inside a plane image, key tile 0)
  // K fragment: row 16u + j, logical chunk 4ks + g  ->  slot (4ks + g) ^ (j & 6)
  int koff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) koff[ks] = j * 128 + (((4 * ks + g) ^ (j & 6)) << 4);
  // V transpose read: lane i of group g points at 4 dims (i & 3) * 4 .. of key row 4g + (i >> 2) (+16 for the second
  // half), logical dims 16n + ..  ->  chunk 2n + ((i >> 1) & 1), slot chunk ^ (row & 6)
  unsigned vaddr0[4];
  {
    const int row = 4 * g + (j >> 2), s2 = (row >> 1) & 3;
#pragma unroll
    for (int n = 0; n < 4; ++n)
      vaddr0[n] = (unsigned)(size_t)kvbuf + (unsigned)(row * 128 + ((n ^ s2) << 5) + ((j & 3) << 3));
  }
  const float c2 = scale * 1.44269504088896340736f;
  const unsigned short* Kbh = Kh + b * k_bs + h * 64;
  const unsigned short* Kbl = Kl + b * k_bs + h * 64;
  const unsigned short* Vbh = Vh + b * v_bs + h * 64;
  const unsigned short* Vbl = Vl + b * v_bs + h * 64;
  // keys S1 .. Sk-1 live in a SECOND buffer (row key - S1): e.g. the instruction rows of the local encoder's [map | txt]
  // context, whose K / V projections are constant over an episode and are kept apart from the per-step map rows
  const unsigned short* K2bh = K2h + b * kv2_bs + h * 64;
  const unsigned short* K2bl = K2l + b * kv2_bs + h * 64;
  const unsigned short* V2bh = V2h + b * kv2_bs + h * 64;
  const unsigned short* V2bl = V2l + b * kv2_bs + h * 64;

  // ---- chunk ring: two LDS buffers; wave NW is the LOADER: it copies chunk c+1 (global -> LDS by LDS-DMA) while the
  // NW math waves work on chunk c -- a wave that issues DMA into a busy memory pipe is blocked at issue for ~200
  // cycles per 1-KiB piece (tools/att_prof.py), time the math waves do not have.  One s_barrier per chunk hands the
  // buffers over.  Rows past Sk re-read row Sk-1 (they are masked).
  auto stage = [&](int key0c, int buf) {   // piece = (plane, 8 key rows); lane -> (row r0 + lane / 8, slot lane % 8)
    if (AB == 1 || AB == 3) return;        // timing ablation: no staging
    const int lrow = lane >> 3, coff = ((lane & 7) ^ (lrow & 6)) << 3;   // r0 % 8 == 0: the slot swizzle is per lane
    const int li = NL > 1 ? wave - NW : 0;                               // the NL loader waves take every NL-th row group
#pragma unroll
    for (int r0 = 8 * li; r0 < KC; r0 += 8 * NL) {
      const int key = min(key0c + r0 + lrow, Sk - 1);
      unsigned short* d = kvbuf + buf * (4 * PLANE) + r0 * 64;
      if constexpr (SEG == 0) {
        const size_t ko = (size_t)key * k_rs + coff, vo = (size_t)key * v_rs + coff;
        dma16(Kbh + ko, d);
        dma16(Kbl + ko, d + PLANE);
        dma16(Vbh + vo, d + 2 * PLANE);
        dma16(Vbl + vo, d + 3 * PLANE);
      } else if constexpr (SEG == 1) {
        const bool s2 = key0c + r0 >= S1;               // wave-uniform: the piece's 8 keys lie in one buffer
        const int kk = s2 ? key - S1 : key;
        const size_t ko = (size_t)kk * (s2 ? kv2_rs : k_rs) + coff, vo = (size_t)kk * (s2 ? kv2_rs : v_rs) + coff;
        dma16((s2 ? K2bh : Kbh) + ko, d);
        dma16((s2 ? K2bl : Kbl) + ko, d + PLANE);
        dma16((s2 ? V2bh : Vbh) + vo, d + 2 * PLANE);
        dma16((s2 ? V2bl : Vbl) + vo, d + 3 * PLANE);
      } else {
        const bool s2 = key >= S1;
        const size_t ko = s2 ? (size_t)(key - S1) * kv2_rs + coff : (size_t)key * k_rs + coff;
        const size_t vo = s2 ? ko : (size_t)key * v_rs + coff;
        dma16((s2 ? K2bh : Kbh) + ko, d);
        dma16((s2 ? K2bl : Kbl) + ko, d + PLANE);
        dma16((s2 ? V2bh : Vbh) + vo, d + 2 * PLANE);
        dma16((s2 ? V2bl : Vbl) + vo, d + 3 * PLANE);
      }
    }
  };
#ifdef GRIDMM_ATT_PROF
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_readcyclecounter();
#endif
  if (wave >= NW) {                        // ---------------- loader wave(s)
    stage(0, 0);
    if (wave == NW) {   // validity words from independent byte loads, under the first DMA
      const uint8_t* mrow = kmask ? kmask + (size_t)b * mask_bs : nullptr;
      for (int i = 0; i < ((Sk + 63) >> 6); ++i) {
        const int k = i * 64 + lane;
        const unsigned long long bal = __ballot((k < Sk) && (!mrow || mrow[k]));
        if (lane == 0) { s_mw[2 * i] = (unsigned)bal; s_mw[2 * i + 1] = (unsigned)(bal >> 32); }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // chunk 0 and the validity words are visible
    int lb = 0;
    for (int key0c = 0; key0c < Sk; key0c += KC, lb ^= 1) {
      if (key0c + KC < Sk) {
        stage(key0c + KC, lb ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();        // chunk c+1 landed; the math waves are done with chunk c
    }
    return;
  }

  const int qt0 = (blockIdx.x * NW + wave) * NQ;                  // first query tile of this wave
  // Q^T as the B operand of S^T = K Q^T: lane (query j, k-chunk g) holds head dims 32 ks + 8g .. +8.  Query tiles past
  // Sq run on a copy of row Sq-1 and are not stored.
  bf16x8_t qh[NQ][2], ql[NQ][2];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const size_t qo = b * q_bs + (size_t)min((qt0 + t) * 16 + j, Sq - 1) * q_rs + h * 64 + 8 * g;
    qh[t][0] = *reinterpret_cast<const bf16x8_t*>(Qh + qo); qh[t][1] = *reinterpret_cast<const bf16x8_t*>(Qh + qo + 32);
    ql[t][0] = *reinterpret_cast<const bf16x8_t*>(Ql + qo); ql[t][1] = *reinterpret_cast<const bf16x8_t*>(Ql + qo + 32);
  }
  f32x4_t o[NQ][4];
  float m_run[NQ], l_run[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    m_run[t] = NEG_BIG; l_run[t] = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) o[t][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  int buf = 0;
  GRIDMM_T(0);                                                    // prologue issue
  __builtin_amdgcn_s_barrier();                                   // chunk 0 and the validity words are visible
  GRIDMM_T(1);
  for (int key0c = 0; key0c < Sk; key0c += KC, buf ^= 1) {
    const int rows = min(KC, ((Sk - key0c + 31) >> 5) << 5);      // multiple of 32
    if (key0c) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS reads of the previous chunk are done
      __builtin_amdgcn_s_barrier();                               // hand-over: next chunk landed, previous buffer free
      GRIDMM_T(2);
    }
    if (AB == 2 || AB == 3) continue;                             // timing ablation: no math (barriers stay)
    const unsigned short* kv = kvbuf + buf * (4 * PLANE);
    unsigned vaddr[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) vaddr[n] = vaddr0[n] + (unsigned)(buf * (4 * PLANE) * 2);

    // ---- the chunk's (<= KC / 32) key tiles in three lock-step phases over the wave's NQ query tiles: all score
    // tiles (independent MFMA chains), ONE max / rescale / row-sum exchange per query tile, all P V tiles.
    unsigned mw[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T)
      mw[T] = (T * 32 < rows) ? __builtin_amdgcn_readfirstlane(s_mw[(key0c >> 5) + T]) : 0u;
    f32x4_t st[NQ][NT][2];
    static_for<NT>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if (mw[T] == 0u) return;                                    // wave-uniform: absent or fully masked tile
      bf16x8_t kh[2][2], kl[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = ((T * 32 + 16 * u) * 128 + koff[ks]) >> 1;
          kh[u][ks] = *reinterpret_cast<const bf16x8_t*>(kv + off);
          kl[u][ks] = *reinterpret_cast<const bf16x8_t*>(kv + PLANE + off);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < NQ; ++t) st[t][T][u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      // term-major: the 2 * NQ accumulators of the tile take turns, dependent MFMAs are 2 * NQ apart
#define GRIDMM_S_TERM(KF, QF, KS)                                                                                  \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int t = 0; t < NQ; ++t)                    \
      st[t][T][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KF[u][KS], QF[t][KS], st[t][T][u], 0, 0, 0)
      GRIDMM_S_TERM(kl, qh, 0);
      GRIDMM_S_TERM(kh, ql, 0);
      GRIDMM_S_TERM(kl, qh, 1);
      GRIDMM_S_TERM(kh, ql, 1);
      GRIDMM_S_TERM(kh, qh, 0);
      GRIDMM_S_TERM(kh, qh, 1);
#undef GRIDMM_S_TERM
    });

    asm volatile("s_nop 0" ::: "memory");
    GRIDMM_T(4);                                                  // S tiles (issue)
    float mx[NQ], alpha[NQ], ps[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) mx[t] = NEG_BIG;
    static_for<NT>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if (mw[T] == 0u) return;
      if (mw[T] == 0xffffffffu) {
#pragma unroll
        for (int t = 0; t < NQ; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) mx[t] = fmaxf(mx[t], st[t][T][e >> 2][e & 3]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool valid = (mw[T] >> (16 * (e >> 2) + 4 * g + (e & 3))) & 1u;
#pragma unroll
          for (int t = 0; t < NQ; ++t) mx[t] = fmaxf(mx[t], valid ? st[t][T][e >> 2][e & 3] : NEG_BIG);
        }
      }
    });
#pragma unroll
    for (int t = 0; t < NQ; ++t) mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 16, 64));
#pragma unroll
    for (int t = 0; t < NQ; ++t) mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 32, 64));
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      // log2 domain: p = 2^(s c - m), c = scale log2(e) > 0 (so the raw maximum is the maximum)
      const float m_new = fmaxf(m_run[t], mx[t] * c2);
      alpha[t] = __builtin_amdgcn_exp2f(m_run[t] - m_new);        // 0 on the first valid chunk, 1 while all masked
      m_run[t] = m_new;
      ps[t] = 0.f;
    }
    uint4 ph[NQ][NT], pl[NQ][NT];
    static_for<NT>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if (mw[T] == 0u) return;
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        float p[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][T][e >> 2][e & 3], c2, -m_run[t]));
        if (mw[T] != 0xffffffffu) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (!((mw[T] >> (16 * (e >> 2) + 4 * g + (e & 3))) & 1u)) p[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) ps[t] += p[e];
        split2_bf16(p[0], p[1], ph[t][T].x, pl[t][T].x);
        split2_bf16(p[2], p[3], ph[t][T].y, pl[t][T].y);
        split2_bf16(p[4], p[5], ph[t][T].z, pl[t][T].z);
        split2_bf16(p[6], p[7], ph[t][T].w, pl[t][T].w);
      }
    });
#pragma unroll
    for (int t = 0; t < NQ; ++t) ps[t] += __shfl_xor(ps[t], 16, 64);
#pragma unroll
    for (int t = 0; t < NQ; ++t) ps[t] += __shfl_xor(ps[t], 32, 64);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      l_run[t] = l_run[t] * alpha[t] + ps[t];
#pragma unroll
      for (int n = 0; n < 4; ++n) o[t][n] *= alpha[t];            // O^T tile: every element of this lane is query j's
    }
    asm volatile("s_nop 0" ::: "memory");
    GRIDMM_T(5);                                                  // softmax
    static_for<NT>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if (mw[T] == 0u) return;
      uint2 vh2[4][2], vl2[4][2];
      static_for<4>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        vh2[n][0] = lds_tr_b64<2 * PLANE * 2 + (T * 32) * 128>(vaddr[n]);
        vh2[n][1] = lds_tr_b64<2 * PLANE * 2 + (T * 32 + 16) * 128>(vaddr[n]);
        vl2[n][0] = lds_tr_b64<3 * PLANE * 2 + (T * 32) * 128>(vaddr[n]);
        vl2[n][1] = lds_tr_b64<3 * PLANE * 2 + (T * 32 + 16) * 128>(vaddr[n]);
      });
      tr_fence(vh2, vl2);
      bf16x8_t vh[4], vl[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        vh[n] = __builtin_bit_cast(bf16x8_t, make_uint4(vh2[n][0].x, vh2[n][0].y, vh2[n][1].x, vh2[n][1].y));
        vl[n] = __builtin_bit_cast(bf16x8_t, make_uint4(vl2[n][0].x, vl2[n][0].y, vl2[n][1].x, vl2[n][1].y));
      }
      // term-major over the 4 * NQ output tiles
#define GRIDMM_O_TERM(VF, PF)                                                                                       \
  _Pragma("unroll") for (int t = 0; t < NQ; ++t) _Pragma("unroll") for (int n = 0; n < 4; ++n)                     \
      o[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(VF[n], __builtin_bit_cast(bf16x8_t, PF[t][T]), o[t][n], 0, 0, 0)
      GRIDMM_O_TERM(vl, ph);
      GRIDMM_O_TERM(vh, pl);
      GRIDMM_O_TERM(vh, ph);
#undef GRIDMM_O_TERM
    });
    asm volatile("s_nop 0" ::: "memory");
    GRIDMM_T(6);                                                  // P V tiles (issue)
  }

  __builtin_amdgcn_s_barrier();                                   // pairs with the loader's last hand-over
  // ---- finish from registers: lane (j, g) holds O[query j][16n + 4g .. +3]
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const int q = (qt0 + t) * 16 + j;
    if (q >= Sq) continue;
    const float inv = l_run[t] > 0.f ? 1.0f / l_run[t] : 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const float x[4] = {o[t][n][0] * inv, o[t][n][1] * inv, o[t][n][2] * inv, o[t][n][3] * inv};
      const int dcol = h * 64 + 16 * n + 4 * g;
      if (O) *reinterpret_cast<float4*>(O + b * o_bs + (size_t)q * o_rs + dcol) = make_float4(x[0], x[1], x[2], x[3]);
      if (Ohi) {
        uint2 hi, lo;
        split2_bf16(x[0], x[1], hi.x, lo.x);
        split2_bf16(x[2], x[3], hi.y, lo.y);
        *reinterpret_cast<uint2*>(Ohi + b * p_bs + (size_t)q * p_rs + dcol) = hi;
        *reinterpret_cast<uint2*>(Olo + b * p_bs + (size_t)q * p_rs + dcol) = lo;
      }
    }
  }
#ifdef GRIDMM_ATT_PROF
  GRIDMM_T(7);                                                    // epilogue
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_att_prof[i], (unsigned long long)prof[i]);
#endif
}

}  // namespace

#ifdef GRIDMM_ATT_PROF
extern "C" int gridmm_debug_att_prof(unsigned long long* out, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_att_prof), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_att_prof), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

#ifdef GRIDMM_DEBUG_HOOKS
// Tuning hook of the development build (tools/sweep_gemm_cfg_step.py): force the launch configuration of the calls with more
// than / at most four query tiles inside a running process (0 = the heuristic).  Not in the shipping library.
static int g_att_cfg_big = 0, g_att_cfg_small = 0;
extern "C" int gridmm_debug_attention_cfg_override(int cfg_big, int cfg_small) {
  g_att_cfg_big = cfg_big;
  g_att_cfg_small = cfg_small;
  return GRIDMM_OK;
}
#endif

// cfg: 0 = auto; 1..: tuning configurations (tools/bench_attn2.py)
static int attention_rows_impl(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                               const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo,
                               int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, float* O,
                               int64_t o_bs, int o_rs, void* O_hi, void* O_lo, int64_t p_bs, int p_rs, int B,
                               int heads, int Sq, int Sk, float scale, int cfg, const void* K2_hi, const void* K2_lo,
                               const void* V2_hi, const void* V2_lo, int64_t kv2_bs, int kv2_rs, int S1,
                               gridmm_stream_t stream) {
  if (B <= 0 || heads <= 0 || Sq <= 0 || Sk <= 0 || Sk > 2048) return GRIDMM_EINVAL;   // mask words: 64 x 32 keys
  if (S1 < 0 || S1 > Sk || (S1 < Sk && (!K2_hi || !K2_lo || !V2_hi || !V2_lo || (kv2_rs & 7) || (kv2_bs & 7)))) return GRIDMM_EINVAL;
  if ((q_rs | k_rs | v_rs) & 7 || (q_bs | k_bs | v_bs) & 7) return GRIDMM_EINVAL;      // 16-byte aligned rows
  if ((!O && !O_hi) || (O_hi && (!O_lo || (p_rs & 3) || (p_bs & 3))) || (O && ((o_rs & 3) || (o_bs & 3))))
    return GRIDMM_EINVAL;
  const int nqt = (Sq + 15) / 16;
  const int seg = S1 >= Sk ? 0 : ((S1 & 7) == 0 ? 1 : 2);
  // (1, 7, 32) for 5 .. 14 query tiles: the 216-query calls are exactly 2 x 7 tiles -- no idle math wave, as (1, 8, 32)
  // leaves in its second workgroup (in-step sweep: -19 us per step over the three calls)
  if (cfg == 0) cfg = nqt <= 4 ? 5 : (nqt <= 14 ? 24 : 9);
#ifdef GRIDMM_DEBUG_HOOKS
  if (nqt > 4 && g_att_cfg_big) cfg = g_att_cfg_big;
  if (nqt <= 4 && g_att_cfg_small) cfg = g_att_cfg_small;
#endif
#define GRIDMM_ATT_ARGS                                                                                             \
  (const unsigned short*)Q_hi, (const unsigned short*)Q_lo, q_bs, q_rs, (const unsigned short*)K_hi,               \
      (const unsigned short*)K_lo, k_bs, k_rs, (const unsigned short*)V_hi, (const unsigned short*)V_lo, v_bs, v_rs, \
      kmask, mask_bs, O, o_bs, o_rs, (unsigned short*)O_hi, (unsigned short*)O_lo, p_bs, p_rs, Sq, Sk, scale,        \
      (const unsigned short*)K2_hi, (const unsigned short*)K2_lo, (const unsigned short*)V2_hi,                      \
      (const unsigned short*)V2_lo, kv2_bs, kv2_rs, S1
#define GRIDMM_ATTS(NQ, NW, KC, AB, NL, SEG)                                                                            \
  do {                                                                                                              \
    dim3 grid((nqt + (NQ) * (NW) - 1) / ((NQ) * (NW)), heads, B), block(((NW) + (NL)) * 64);                               \
    GRIDMM_LAUNCH((attention_rows_kernel<NQ, NW, KC, AB, NL, SEG>), grid, block, 0, as_stream(stream), GRIDMM_ATT_ARGS); \
  } while (0)
#define GRIDMM_ATTL(NQ, NW, KC, AB, NL)                                                                                 \
  do {                                                                                                              \
    if (seg) return GRIDMM_EINVAL;      /* two context buffers: only the configurations instantiated for it below */ \
    GRIDMM_ATTS(NQ, NW, KC, AB, NL, 0);                                                                             \
  } while (0)
#define GRIDMM_ATTG(NQ, NW, KC, NL)     /* configurations that also take a second context buffer */                   \
  do {                                                                                                              \
    if (seg == 0) GRIDMM_ATTS(NQ, NW, KC, 0, NL, 0);                                                                \
    else if (seg == 1) GRIDMM_ATTS(NQ, NW, KC, 0, NL, 1);                                                           \
    else GRIDMM_ATTS(NQ, NW, KC, 0, NL, 2);                                                                         \
  } while (0)
#define GRIDMM_ATTX(NQ, NW, KC, AB) GRIDMM_ATTL(NQ, NW, KC, AB, 1)
#define GRIDMM_ATT(NQ, NW, KC) GRIDMM_ATTX(NQ, NW, KC, 0)
  switch (cfg) {
    case 1: GRIDMM_ATT(1, 4, 64); break;
    case 2: GRIDMM_ATT(2, 4, 64); break;
    case 3: GRIDMM_ATT(1, 8, 64); break;
    case 5: GRIDMM_ATTG(1, 4, 32, 1); break;
    case 6: GRIDMM_ATT(2, 4, 32); break;
    case 7: GRIDMM_ATT(2, 8, 64); break;
    case 8: GRIDMM_ATT(2, 8, 32); break;
    case 9: GRIDMM_ATTG(1, 8, 32, 1); break;
    case 20: GRIDMM_ATT(1, 14, 32); break;         // ONE workgroup per (episode, head) for up to 224 queries: K / V staged once
    case 21: GRIDMM_ATT(1, 14, 64); break;
    case 22: GRIDMM_ATTG(1, 14, 32, 2); break;
    case 23: GRIDMM_ATT(1, 12, 32); break;
    case 24: GRIDMM_ATTG(1, 7, 32, 1); break;          // two workgroups of 7 query tiles (216 queries = 14 tiles: no idle wave)
    case 25: GRIDMM_ATT(2, 7, 32); break;          // one workgroup, two query tiles per wave
    case 14: GRIDMM_ATTL(1, 8, 64, 0, 2); break;   // two loader waves
    case 15: GRIDMM_ATTG(1, 4, 32, 2); break;
    case 16: GRIDMM_ATTL(2, 8, 64, 0, 2); break;
    case 17: GRIDMM_ATTL(2, 4, 64, 0, 2); break;
    case 18: GRIDMM_ATTL(1, 4, 64, 0, 2); break;
    case 11: GRIDMM_ATTX(2, 4, 64, 1); break;   // ablations of cfg 2: no staging / no math
    case 12: GRIDMM_ATTX(2, 4, 64, 2); break;
    case 13: GRIDMM_ATTX(2, 4, 64, 3); break;
    default: return GRIDMM_EINVAL;
  }
#undef GRIDMM_ATT
#undef GRIDMM_ATTX
#undef GRIDMM_ATTL
#undef GRIDMM_ATTG
#undef GRIDMM_ATTS
#undef GRIDMM_ATT_ARGS
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_attention_rows_cfg(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                                         const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo,
                                         int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, float* O,
                                         int64_t o_bs, int o_rs, void* O_hi, void* O_lo, int64_t p_bs, int p_rs, int B,
                                         int heads, int Sq, int Sk, float scale, int cfg, gridmm_stream_t stream) {
  return attention_rows_impl(Q_hi, Q_lo, q_bs, q_rs, K_hi, K_lo, k_bs, k_rs, V_hi, V_lo, v_bs, v_rs, kmask, mask_bs, O, o_bs,
                             o_rs, O_hi, O_lo, p_bs, p_rs, B, heads, Sq, Sk, scale, cfg, nullptr, nullptr, nullptr, nullptr, 0,
                             0, Sk, stream);
}

// Keys [0, S1) from K / V, keys [S1, Sk) from a second pair of plane buffers (K2 / V2: row key - S1, row stride kv2_rs,
// episode stride kv2_bs); kmask spans all Sk keys.  Same arithmetic, key by key, as one concatenated buffer.
extern "C" int gridmm_attention_rows_seg(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                                         const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo,
                                         int64_t v_bs, int v_rs, int S1, const void* K2_hi, const void* K2_lo,
                                         const void* V2_hi, const void* V2_lo, int64_t kv2_bs, int kv2_rs,
                                         const uint8_t* kmask, int mask_bs, float* O, int64_t o_bs, int o_rs, void* O_hi,
                                         void* O_lo, int64_t p_bs, int p_rs, int B, int heads, int Sq, int Sk, float scale,
                                         gridmm_stream_t stream) {
  return attention_rows_impl(Q_hi, Q_lo, q_bs, q_rs, K_hi, K_lo, k_bs, k_rs, V_hi, V_lo, v_bs, v_rs, kmask, mask_bs, O, o_bs,
                             o_rs, O_hi, O_lo, p_bs, p_rs, B, heads, Sq, Sk, scale, 0, K2_hi, K2_lo, V2_hi, V2_lo, kv2_bs,
                             kv2_rs, S1, stream);
}

extern "C" int gridmm_attention_rows(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                                     const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo,
                                     int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, float* O, int64_t o_bs,
                                     int o_rs, void* O_hi, void* O_lo, int64_t p_bs, int p_rs, int B, int heads, int Sq,
                                     int Sk, float scale, gridmm_stream_t stream) {
  return gridmm_attention_rows_cfg(Q_hi, Q_lo, q_bs, q_rs, K_hi, K_lo, k_bs, k_rs, V_hi, V_lo, v_bs, v_rs, kmask, mask_bs,
                                   O, o_bs, o_rs, O_hi, O_lo, p_bs, p_rs, B, heads, Sq, Sk, scale, 0, stream);
}
