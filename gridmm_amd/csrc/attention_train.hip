// Attention of the DIFFERENTIABLE path on the bf16 matrix pipe (round 5): forward (+ log-sum-exp, dropout on the
// probabilities) and the two backward kernels in the 3-term bf16 split of the inference kernel (attention_lds.hip) instead of
// exact fp32 on v_mfma_f32_16x16x4_f32 (1/16 of the bf16 rate; attention.hip / attention_bwd.hip, which stay for operands
// that arrive without planes).  Reference: BertSelfAttention / BertOutAttention with attention_probs_dropout_prob
// (map_nav_src/models/vilmodel.py:95-157, 317-368; pretrain_src/model/vilmodel.py twins) and nn.MultiheadAttention of the
// grid encoder (models/transformer.py:176-177), forward and torch-autograd backward.
//
//   P  = softmax(Q K^T s + key mask)     O = drop(P) V         delta_q = sum_d dO[q][d] O[q][d]
//   dP = drop'(dO V^T)                   dS = P o (dP - delta) dQ = s dS K     dK = s dS^T Q     dV = drop(P)^T dO
//
// All three kernels share the data path of attention_rows_kernel: one LOADER wave copies 32-row chunks of four row-major
// bf16 planes (hi / lo of two operands) global -> LDS by LDS-DMA into a two-buffer ring (bank swizzle on the SOURCE
// address: 16-byte slot `pos` of row r holds chunk pos ^ (r & 6)); the math waves read an operand either row-wise
// (ds_read_b128: A operand "16 rows x 32 dims") or transposed (ds_read_b64_tr_b16: A operand "16 dims x 32 row slots") out
// of the same image, and the tile a lane holds after the first contraction is the B operand of the second as it is
// (row slot 8g + e = row 4g + e (e < 4) / 16 + 4g + (e - 4) of the 32-row tile).
//   forward : workgroup = (b, h, query tiles); walks KEY chunks [K | V];  S^T = K Q^T,  O^T += V^T P^T
//   dq      : workgroup = (b, h, query tiles); walks KEY chunks [K | V];  S^T = K Q^T,  dP^T = V dO^T,  dQ^T += K^T dS^T
//   dkv     : workgroup = (b, h, key tiles);   walks QUERY chunks [Q | dO]; S = Q K^T,  dP = dO V^T,
//                                                                          dV^T += dO^T drop(P),  dK^T += Q^T dS
// Scores live in the log2 domain (p = 2^(s c - lse2), c = scale log2 e); the saved statistic is lse2 = log2 sum_k 2^(s c)
// (+1e30 for a fully masked row: every probability of the backward is then exactly 0).  The dropout mask is the counter
// hash of common.h over (b, h, q, k) -- the same mask as the fp32 kernels for the same seed.
//
// SHIFTED K / V (vbar != NULL).  A bf16x3 product carries an absolute error of ~2^-18 |a||b|, and dS = P o (dP - delta)
// subtracts two numbers that share whatever is common to all keys: with K / V rows that have a large common component
// (LayerNorm bias, type embeddings: the normal case) the q / k weight gradients came out with 10-15x the error of the fp32
// kernels (tools/dbg_pretrain_grad_errors.py, tools/dbg_center_kv.py).  The callers therefore hand planes of K - K[row 0 of
// the episode] and V - V[row 0] (gridmm_linear_planes_shift) and the row V[row 0] itself as `vbar`: softmax(Q K^T) does not
// change under a shift of K, and
//   O  = drop(P) V = drop(P) V' + r_q vbar,   r_q = sum_k drop(P)_qk         (forward epilogue)
//   dP = drop'(dO V'^T + c_q),                c_q = <dO_q, vbar>             (prep pass; exact fp32)
//   dQ = s dS K' (sum_k dS_qk = 0),  dK, dV unchanged (they contain no K / V values)
// which measures at the error level of the exact-fp32 kernels again.
#include <type_traits>

#include "common.h"

namespace {

constexpr float BIG = 1.0e30f;

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
// s_waitcnt lgkmcnt(0) that the uses of the eight pairs cannot be scheduled above (tests/test_isa_hazards.py walks this file)
__device__ __forceinline__ void tr_fence(uint2 (&a)[4][2], uint2 (&b)[4][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]),
                 "+v"(a[3][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]),
                 "+v"(b[3][0]), "+v"(b[3][1])
               :
               : "memory");
}

constexpr int KC = 32;                    // rows per chunk (one 32-row tile)
constexpr int PLANE = KC * 64;            // u16 per plane image
constexpr int BUF = 4 * PLANE;            // u16 per ring buffer: operand 0 hi | lo | operand 1 hi | lo

// One chunk: rows [row0, row0 + KC) of the four planes (per-head slices: 64 columns at the pointers) -> ring buffer `dst`.
// Rows past nrows re-read row nrows - 1 (their contributions are masked by the callers).
__device__ __forceinline__ void stage_chunk(const unsigned short* a_hi, const unsigned short* a_lo, int a_rs,
                                            const unsigned short* b_hi, const unsigned short* b_lo, int b_rs, int row0, int nrows,
                                            unsigned short* dst, int lane) {
  const int lrow = lane >> 3, coff = ((lane & 7) ^ (lrow & 6)) << 3;
#pragma unroll
  for (int r0 = 0; r0 < KC; r0 += 8) {
    const int row = min(row0 + r0 + lrow, nrows - 1);
    unsigned short* d = dst + r0 * 64;
    const size_t ao = (size_t)row * a_rs + coff, bo = (size_t)row * b_rs + coff;
    dma16(a_hi + ao, d);
    dma16(a_lo + ao, d + PLANE);
    dma16(b_hi + bo, d + 2 * PLANE);
    dma16(b_lo + bo, d + 3 * PLANE);
  }
}

// Fragment of 16 rows of a plane pair held in registers as the B operand of a "rows x rows^T" product: lane (row j, g)
// holds dims 32 ks + 8 g .. + 7 (rows past nrows: a copy of row nrows - 1)
__device__ __forceinline__ void load_frag(const unsigned short* hi, const unsigned short* lo, int rs, int row, bf16x8_t (&fh)[2],
                                          bf16x8_t (&fl)[2], int g) {
  const size_t o = (size_t)row * rs + 8 * g;
  fh[0] = *reinterpret_cast<const bf16x8_t*>(hi + o); fh[1] = *reinterpret_cast<const bf16x8_t*>(hi + o + 32);
  fl[0] = *reinterpret_cast<const bf16x8_t*>(lo + o); fl[1] = *reinterpret_cast<const bf16x8_t*>(lo + o + 32);
}

// rows-as-A fragments of plane pair `pl` (0: operand 0, 1: operand 1) of the 32-row tile in `buf`: [u][ks]
__device__ __forceinline__ void read_rows(const unsigned short* buf, int pl, const int (&koff)[2], bf16x8_t (&fh)[2][2],
                                          bf16x8_t (&fl)[2][2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ((16 * u) * 128 + koff[ks]) >> 1;
      fh[u][ks] = *reinterpret_cast<const bf16x8_t*>(buf + 2 * pl * PLANE + off);
      fl[u][ks] = *reinterpret_cast<const bf16x8_t*>(buf + (2 * pl + 1) * PLANE + off);
    }
}

// transposed-as-A fragments (16 dims x 32 row slots) of plane pair PL of the tile whose image starts at byte address base
template <int PL>
__device__ __forceinline__ void read_tr(const unsigned (&vaddr)[4], bf16x8_t (&fh)[4], bf16x8_t (&fl)[4]) {
  uint2 h2[4][2], l2[4][2];
  static_for<4>([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    h2[n][0] = lds_tr_b64<(2 * PL) * PLANE * 2>(vaddr[n]);
    h2[n][1] = lds_tr_b64<(2 * PL) * PLANE * 2 + 16 * 128>(vaddr[n]);
    l2[n][0] = lds_tr_b64<(2 * PL + 1) * PLANE * 2>(vaddr[n]);
    l2[n][1] = lds_tr_b64<(2 * PL + 1) * PLANE * 2 + 16 * 128>(vaddr[n]);
  });
  tr_fence(h2, l2);
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    fh[n] = __builtin_bit_cast(bf16x8_t, make_uint4(h2[n][0].x, h2[n][0].y, h2[n][1].x, h2[n][1].y));
    fl[n] = __builtin_bit_cast(bf16x8_t, make_uint4(l2[n][0].x, l2[n][0].y, l2[n][1].x, l2[n][1].y));
  }
}

// acc[u] += A[u] (16 rows x 64 dims, hi/lo) . B (64 dims x 16 cols, hi/lo), 3-term split, both 16-row halves of the tile
__device__ __forceinline__ void mma_rows(const bf16x8_t (&ah)[2][2], const bf16x8_t (&al)[2][2], const bf16x8_t (&bh)[2],
                                         const bf16x8_t (&bl)[2], f32x4_t (&acc)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[u][ks], bh[ks], acc[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[u][ks], bl[ks], acc[u], 0, 0, 0);
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[u][ks], bh[ks], acc[u], 0, 0, 0);
}

// o[n] += A^T[n] (16 dims x 32 row slots, hi/lo) . X (32 row slots x 16 cols: the 8 values a lane holds, hi/lo)
__device__ __forceinline__ void mma_tr(const bf16x8_t (&ah)[4], const bf16x8_t (&al)[4], const uint4& xh, const uint4& xl,
                                       f32x4_t (&o)[4]) {
  const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, xh), bl = __builtin_bit_cast(bf16x8_t, xl);
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[n], bh, o[n], 0, 0, 0);
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[n], bl, o[n], 0, 0, 0);
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[n], bh, o[n], 0, 0, 0);
}

__device__ __forceinline__ void split8(const float (&x)[8], uint4& hi, uint4& lo) {
  split2_bf16(x[0], x[1], hi.x, lo.x);
  split2_bf16(x[2], x[3], hi.y, lo.y);
  split2_bf16(x[4], x[5], hi.z, lo.z);
  split2_bf16(x[6], x[7], hi.w, lo.w);
}

// Per-lane LDS offsets of the two read patterns (see attention_lds.hip)
__device__ __forceinline__ void lane_offsets(const unsigned short* ring, int j, int g, int (&koff)[2], unsigned (&vaddr0)[4]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) koff[ks] = j * 128 + (((4 * ks + g) ^ (j & 6)) << 4);
  const int row = 4 * g + (j >> 2), s2 = (row >> 1) & 3;
#pragma unroll
  for (int n = 0; n < 4; ++n) vaddr0[n] = (unsigned)(size_t)ring + (unsigned)(row * 128 + ((n ^ s2) << 5) + ((j & 3) << 3));
}

// =====================================================================================================================
// forward: O = drop(softmax(Q K^T s + mask)) V, lse2 out.  NW math waves of one query tile each + one loader wave.
template <int NW>
__global__ __launch_bounds__((NW + 1) * 64) void attention_rows_train_kernel(
    const unsigned short* __restrict__ Qh, const unsigned short* __restrict__ Ql, int64_t q_bs, int q_rs,
    const unsigned short* __restrict__ Kh, const unsigned short* __restrict__ Kl, int64_t k_bs, int k_rs,
    const unsigned short* __restrict__ Vh, const unsigned short* __restrict__ Vl, int64_t v_bs, int v_rs,
    const uint8_t* __restrict__ kmask, int mask_bs, float* __restrict__ O, int64_t o_bs, int o_rs,
    unsigned short* __restrict__ Ohi, unsigned short* __restrict__ Olo, int64_t p_bs, int p_rs, float* __restrict__ lse2,
    int Sqp, int Sq, int Sk, float scale, float drop_p, unsigned long long seed,
    const unsigned long long* __restrict__ seed_dev, const float* __restrict__ vbar, int64_t vb_bs) {
  __shared__ __attribute__((aligned(16))) unsigned short ring[2 * BUF];
  __shared__ unsigned s_mw[64];                                   // key validity, one word per 32 keys (Sk <= 2048)
  if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  const int j = lane & 15, g = lane >> 4;
  const unsigned short *Kbh = Kh + b * k_bs + h * 64, *Kbl = Kl + b * k_bs + h * 64;
  const unsigned short *Vbh = Vh + b * v_bs + h * 64, *Vbl = Vl + b * v_bs + h * 64;
  if (wave == NW) {                        // ---------------- loader
    stage_chunk(Kbh, Kbl, k_rs, Vbh, Vbl, v_rs, 0, Sk, ring, lane);
    const uint8_t* mrow = kmask ? kmask + (size_t)b * mask_bs : nullptr;
    for (int i = 0; i < ((Sk + 63) >> 6); ++i) {
      const int k = i * 64 + lane;
      const unsigned long long bal = __ballot((k < Sk) && (!mrow || mrow[k]));
      if (lane == 0) { s_mw[2 * i] = (unsigned)bal; s_mw[2 * i + 1] = (unsigned)(bal >> 32); }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int lb = 0;
    for (int key0 = 0; key0 < Sk; key0 += KC, lb ^= 1) {
      if (key0 + KC < Sk) {
        stage_chunk(Kbh, Kbl, k_rs, Vbh, Vbl, v_rs, key0 + KC, Sk, ring + (lb ^ 1) * BUF, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  int koff[2];
  unsigned vaddr0[4];
  lane_offsets(ring, j, g, koff, vaddr0);
  const float c2 = scale * 1.44269504088896340736f;
  const int q = (blockIdx.x * NW + wave) * 16 + j;                // (query tiles past Sq run on a copy of row Sq - 1)
  bf16x8_t qh[2], ql[2];
  load_frag(Qh + b * q_bs + h * 64, Ql + b * q_bs + h * 64, q_rs, min(q, Sq - 1), qh, ql, g);
  f32x4_t o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -BIG, l_run = 0.f, r_run = 0.f;                   // r: row sum of the DROPPED probabilities (shifted V only)
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const unsigned drow = ((unsigned)(b * heads + h) * Sq + (unsigned)q) * (unsigned)Sk;

  int buf = 0;
  __builtin_amdgcn_s_barrier();
  for (int key0 = 0; key0 < Sk; key0 += KC, buf ^= 1) {
    if (key0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const unsigned mw = __builtin_amdgcn_readfirstlane(s_mw[key0 >> 5]);
    if (mw == 0u) continue;                                       // wave-uniform: fully masked tile
    const unsigned short* kv = ring + buf * BUF;
    bf16x8_t kh[2][2], kl[2][2];
    read_rows(kv, 0, koff, kh, kl);
    f32x4_t st[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
    mma_rows(kh, kl, qh, ql, st);                                 // S^T[key 16u + 4g + r][query j]
    float mx = -BIG;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool valid = (mw >> (16 * (e >> 2) + 4 * g + (e & 3))) & 1u;
      mx = fmaxf(mx, valid ? st[e >> 2][e & 3] : -BIG);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * c2);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float p[8], ps = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool valid = (mw >> (16 * (e >> 2) + 4 * g + (e & 3))) & 1u;
      p[e] = valid ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[e >> 2][e & 3], c2, -m_run)) : 0.f;
      ps += p[e];
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;                                   // (the normaliser stays un-dropped, vilmodel.py:143)
    if (drop_p > 0.f) {
      float pd = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p[e] = dropout_keep(seed, drow + (unsigned)(key0 + 16 * (e >> 2) + 4 * g + (e & 3)), drop_p) ? p[e] * keep_scale : 0.f;
        pd += p[e];
      }
      if (vbar) {
        pd += __shfl_xor(pd, 16, 64);
        pd += __shfl_xor(pd, 32, 64);
        r_run = r_run * alpha + pd;
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] *= alpha;
    uint4 ph, pl;
    split8(p, ph, pl);
    unsigned vaddr[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) vaddr[n] = vaddr0[n] + (unsigned)(buf * BUF * 2);
    bf16x8_t vh[4], vl[4];
    read_tr<1>(vaddr, vh, vl);
    mma_tr(vh, vl, ph, pl, o);                                    // O^T[dim 16n + 4g + r][query j]
  }
  __builtin_amdgcn_s_barrier();                                   // pairs with the loader's last hand-over
  if (q >= Sq) return;
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  if (g == 0) lse2[((size_t)b * heads + h) * Sqp + q] = l_run > 0.f ? m_run + __builtin_amdgcn_logf(l_run) : BIG;
  const float rq = vbar ? (drop_p > 0.f ? r_run : l_run) * inv : 0.f;       // sum_k drop(P)_qk (1 without dropout, 0 if all masked)
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float x[4] = {o[n][0] * inv, o[n][1] * inv, o[n][2] * inv, o[n][3] * inv};
    const int dcol = h * 64 + 16 * n + 4 * g;
    if (vbar) {
      const float4 vb = *reinterpret_cast<const float4*>(vbar + b * vb_bs + dcol);
      x[0] += rq * vb.x; x[1] += rq * vb.y; x[2] += rq * vb.z; x[3] += rq * vb.w;
    }
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

// =====================================================================================================================
// backward, pass 0: delta[b][h][q] = <dO, O> over the head's 64 dims, and the bf16 hi/lo planes of dO ([B][Sq][H]).
// One wave per (b, q) row; a 16-lane group covers one head of a 256-column slice.
__global__ __launch_bounds__(256) void attention_bwd_prep_kernel(const float* __restrict__ dO, int64_t do_bs, int do_rs,
                                                                 const float* __restrict__ O, int64_t o_bs, int o_rs,
                                                                 unsigned short* __restrict__ dOh,
                                                                 unsigned short* __restrict__ dOl, float* __restrict__ delta,
                                                                 float* __restrict__ cq, const float* __restrict__ vbar,
                                                                 int64_t vb_bs, int heads, int Sq, int Sqp, int B) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * Sq) return;
  const int b = (int)(row / Sq), q = (int)(row - (long)b * Sq), H = heads * 64;
  for (int e0 = 4 * lane; e0 < H; e0 += 256) {
    const float4 a = *reinterpret_cast<const float4*>(dO + b * do_bs + (size_t)q * do_rs + e0);
    const float4 c = *reinterpret_cast<const float4*>(O + b * o_bs + (size_t)q * o_rs + e0);
    float s = (a.x * c.x + a.y * c.y) + (a.z * c.z + a.w * c.w);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    if ((lane & 15) == 0) delta[((size_t)b * heads + (e0 >> 6)) * Sqp + q] = s;
    if (vbar) {                                                   // c_q = <dO_q, vbar> (shifted V, see the header)
      const float4 vb = *reinterpret_cast<const float4*>(vbar + b * vb_bs + e0);
      float t = (a.x * vb.x + a.y * vb.y) + (a.z * vb.z + a.w * vb.w);
      t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
      if ((lane & 15) == 0) cq[((size_t)b * heads + (e0 >> 6)) * Sqp + q] = t;
    }
    uint2 hi, lo;
    split2_bf16(a.x, a.y, hi.x, lo.x);
    split2_bf16(a.z, a.w, hi.y, lo.y);
    *reinterpret_cast<uint2*>(dOh + (size_t)row * H + e0) = hi;
    *reinterpret_cast<uint2*>(dOl + (size_t)row * H + e0) = lo;
  }
}

// =====================================================================================================================
// backward, dQ: a math wave owns 16 queries and walks the keys.
template <int NW>
__global__ __launch_bounds__((NW + 1) * 64) void attention_rows_bwd_dq_kernel(
    const unsigned short* __restrict__ Qh, const unsigned short* __restrict__ Ql, int64_t q_bs, int q_rs,
    const unsigned short* __restrict__ Kh, const unsigned short* __restrict__ Kl, int64_t k_bs, int k_rs,
    const unsigned short* __restrict__ Vh, const unsigned short* __restrict__ Vl, int64_t v_bs, int v_rs,
    const uint8_t* __restrict__ kmask, int mask_bs, const unsigned short* __restrict__ dOh,
    const unsigned short* __restrict__ dOl, const float* __restrict__ lse2, const float* __restrict__ delta,
    const float* __restrict__ cq, float* __restrict__ dQ, int64_t dq_bs, int dq_rs, int Sq, int Sk, int Sqp, float scale, float drop_p,
    unsigned long long seed, const unsigned long long* __restrict__ seed_dev, unsigned short* __restrict__ dQh,
    unsigned short* __restrict__ dQl) {
  __shared__ __attribute__((aligned(16))) unsigned short ring[2 * BUF];
  __shared__ unsigned s_mw[64];
  if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y, H = heads * 64;
  const int j = lane & 15, g = lane >> 4;
  const unsigned short *Kbh = Kh + b * k_bs + h * 64, *Kbl = Kl + b * k_bs + h * 64;
  const unsigned short *Vbh = Vh + b * v_bs + h * 64, *Vbl = Vl + b * v_bs + h * 64;
  if (wave == NW) {                        // ---------------- loader (as in the forward)
    stage_chunk(Kbh, Kbl, k_rs, Vbh, Vbl, v_rs, 0, Sk, ring, lane);
    const uint8_t* mrow = kmask ? kmask + (size_t)b * mask_bs : nullptr;
    for (int i = 0; i < ((Sk + 63) >> 6); ++i) {
      const int k = i * 64 + lane;
      const unsigned long long bal = __ballot((k < Sk) && (!mrow || mrow[k]));
      if (lane == 0) { s_mw[2 * i] = (unsigned)bal; s_mw[2 * i + 1] = (unsigned)(bal >> 32); }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int lb = 0;
    for (int key0 = 0; key0 < Sk; key0 += KC, lb ^= 1) {
      if (key0 + KC < Sk) {
        stage_chunk(Kbh, Kbl, k_rs, Vbh, Vbl, v_rs, key0 + KC, Sk, ring + (lb ^ 1) * BUF, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  int koff[2];
  unsigned vaddr0[4];
  lane_offsets(ring, j, g, koff, vaddr0);
  const float c2 = scale * 1.44269504088896340736f;
  const int q = (blockIdx.x * NW + wave) * 16 + j, qc = min(q, Sq - 1);
  bf16x8_t qh[2], ql[2], doh[2], dol[2];
  load_frag(Qh + b * q_bs + h * 64, Ql + b * q_bs + h * 64, q_rs, qc, qh, ql, g);
  load_frag(dOh + (size_t)b * Sq * H + h * 64, dOl + (size_t)b * Sq * H + h * 64, H, qc, doh, dol, g);
  const size_t st_off = ((size_t)b * heads + h) * Sqp;
  const float lse_j = lse2[st_off + qc], delta_j = delta[st_off + qc], c_j = cq ? cq[st_off + qc] : 0.f;
  f32x4_t acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const unsigned drow = ((unsigned)(b * heads + h) * Sq + (unsigned)q) * (unsigned)Sk;

  int buf = 0;
  __builtin_amdgcn_s_barrier();
  for (int key0 = 0; key0 < Sk; key0 += KC, buf ^= 1) {
    if (key0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const unsigned mw = __builtin_amdgcn_readfirstlane(s_mw[key0 >> 5]);
    if (mw == 0u) continue;
    const unsigned short* kv = ring + buf * BUF;
    bf16x8_t kh[2][2], kl[2][2], vh[2][2], vl[2][2];
    read_rows(kv, 0, koff, kh, kl);
    read_rows(kv, 1, koff, vh, vl);
    f32x4_t st[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}}, dpt[2] = {st[0], st[0]};
    mma_rows(kh, kl, qh, ql, st);                                 // S^T[key][query j]
    mma_rows(vh, vl, doh, dol, dpt);                              // dP^T[key][query j]
    float ds[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ko = 16 * (e >> 2) + 4 * g + (e & 3);
      const bool valid = (mw >> ko) & 1u;
      const float p = valid ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[e >> 2][e & 3], c2, -lse_j)) : 0.f;
      float dp = dpt[e >> 2][e & 3] + c_j;
      if (drop_p > 0.f) dp = dropout_keep(seed, drow + (unsigned)(key0 + ko), drop_p) ? dp * keep_scale : 0.f;
      ds[e] = p * (dp - delta_j);
    }
    uint4 dsh, dsl;
    split8(ds, dsh, dsl);
    unsigned vaddr[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) vaddr[n] = vaddr0[n] + (unsigned)(buf * BUF * 2);
    bf16x8_t kth[4], ktl[4];
    read_tr<0>(vaddr, kth, ktl);
    mma_tr(kth, ktl, dsh, dsl, acc);                              // dQ^T[dim 16n + 4g + r][query j]
  }
  __builtin_amdgcn_s_barrier();
  if (q >= Sq) return;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float4 o = make_float4(acc[n][0] * scale, acc[n][1] * scale, acc[n][2] * scale, acc[n][3] * scale);
    const size_t off = b * dq_bs + (size_t)q * dq_rs + h * 64 + 16 * n + 4 * g;
    *reinterpret_cast<float4*>(dQ + off) = o;
    if (dQh) {               // the bf16 planes of dQ (same strides): the dY operand of the query projection's backward
      uint2 hi, lo;
      split2_bf16(o.x, o.y, hi.x, lo.x);
      split2_bf16(o.z, o.w, hi.y, lo.y);
      *reinterpret_cast<uint2*>(dQh + off) = hi;
      *reinterpret_cast<uint2*>(dQl + off) = lo;
    }
  }
}

// =====================================================================================================================
// backward, dK / dV: a math wave owns 16 keys and walks the queries ([Q | dO] chunks + their lse2 / delta).
template <int NW>
__global__ __launch_bounds__((NW + 1) * 64) void attention_rows_bwd_dkv_kernel(
    const unsigned short* __restrict__ Qh, const unsigned short* __restrict__ Ql, int64_t q_bs, int q_rs,
    const unsigned short* __restrict__ Kh, const unsigned short* __restrict__ Kl, int64_t k_bs, int k_rs,
    const unsigned short* __restrict__ Vh, const unsigned short* __restrict__ Vl, int64_t v_bs, int v_rs,
    const uint8_t* __restrict__ kmask, int mask_bs, const unsigned short* __restrict__ dOh,
    const unsigned short* __restrict__ dOl, const float* __restrict__ lse2, const float* __restrict__ delta,
    const float* __restrict__ cq, float* __restrict__ dK, int64_t dk_bs, int dk_rs, float* __restrict__ dV, int64_t dv_bs, int dv_rs, int Sq, int Sk, int Sqp,
    float scale, float drop_p, unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
    unsigned short* __restrict__ dKh, unsigned short* __restrict__ dKl, unsigned short* __restrict__ dVh,
    unsigned short* __restrict__ dVl) {
  __shared__ __attribute__((aligned(16))) unsigned short ring[2 * BUF];
  __shared__ __attribute__((aligned(16))) float s_ls[2][KC], s_dl[2][KC], s_cq[2][KC];
  if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y, H = heads * 64;
  const int j = lane & 15, g = lane >> 4;
  const unsigned short *Qbh = Qh + b * q_bs + h * 64, *Qbl = Ql + b * q_bs + h * 64;
  const unsigned short *Dbh = dOh + (size_t)b * Sq * H + h * 64, *Dbl = dOl + (size_t)b * Sq * H + h * 64;
  const size_t st_off = ((size_t)b * heads + h) * Sqp;
  if (wave == NW) {                        // ---------------- loader: [Q | dO] chunks and the row statistics of their queries
    auto stats = [&](int q0, int lb) {
      if (lane < KC) {
        const int q = q0 + lane;
        s_ls[lb][lane] = q < Sq ? lse2[st_off + q] : BIG;       // rows past Sq: probability exactly 0
        s_dl[lb][lane] = q < Sq ? delta[st_off + q] : 0.f;
        s_cq[lb][lane] = (cq && q < Sq) ? cq[st_off + q] : 0.f;
      }
    };
    stage_chunk(Qbh, Qbl, q_rs, Dbh, Dbl, H, 0, Sq, ring, lane);
    stats(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int lb = 0;
    for (int q0 = 0; q0 < Sq; q0 += KC, lb ^= 1) {
      if (q0 + KC < Sq) {
        stage_chunk(Qbh, Qbl, q_rs, Dbh, Dbl, H, q0 + KC, Sq, ring + (lb ^ 1) * BUF, lane);
        stats(q0 + KC, lb ^ 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  int koff[2];
  unsigned vaddr0[4];
  lane_offsets(ring, j, g, koff, vaddr0);
  const float c2 = scale * 1.44269504088896340736f;
  const int key = (blockIdx.x * NW + wave) * 16 + j, kc = min(key, Sk - 1);
  const uint8_t* mrow = kmask ? kmask + (size_t)b * mask_bs : nullptr;
  const bool key_valid = key < Sk && (!mrow || mrow[kc]);
  bf16x8_t kh[2], kl[2], vh[2], vl[2];
  load_frag(Kh + b * k_bs + h * 64, Kl + b * k_bs + h * 64, k_rs, kc, kh, kl, g);
  load_frag(Vh + b * v_bs + h * 64, Vl + b * v_bs + h * 64, v_rs, kc, vh, vl, g);
  f32x4_t ak[4], av[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) { ak[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; av[n] = ak[n]; }
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const unsigned dbase = (unsigned)(b * heads + h) * (unsigned)Sq;
  const bool any_key = __any(key_valid);                          // wave-uniform

  int buf = 0;
  __builtin_amdgcn_s_barrier();
  for (int q0 = 0; q0 < Sq; q0 += KC, buf ^= 1) {
    if (q0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (!any_key) continue;
    const unsigned short* qd = ring + buf * BUF;
    bf16x8_t qh[2][2], ql[2][2], doh[2][2], dol[2][2];
    read_rows(qd, 0, koff, qh, ql);
    read_rows(qd, 1, koff, doh, dol);
    const float4 l0 = *reinterpret_cast<const float4*>(&s_ls[buf][4 * g]), l1 = *reinterpret_cast<const float4*>(&s_ls[buf][16 + 4 * g]);
    const float4 d0 = *reinterpret_cast<const float4*>(&s_dl[buf][4 * g]), d1 = *reinterpret_cast<const float4*>(&s_dl[buf][16 + 4 * g]);
    const float4 c0 = *reinterpret_cast<const float4*>(&s_cq[buf][4 * g]), c1 = *reinterpret_cast<const float4*>(&s_cq[buf][16 + 4 * g]);
    const float ls[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w}, dl[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    f32x4_t s[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}}, dp[2] = {s[0], s[0]};
    mma_rows(qh, ql, kh, kl, s);                                  // S[query 16u + 4g + r][key j]
    mma_rows(doh, dol, vh, vl, dp);                               // dP[query][key j]
    float pd[8], ds[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int qo = 16 * (e >> 2) + 4 * g + (e & 3);
      const float p = key_valid ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[e >> 2][e & 3], c2, -ls[e])) : 0.f;
      float m = 1.0f;
      if (drop_p > 0.f)
        m = dropout_keep(seed, (dbase + (unsigned)(q0 + qo)) * (unsigned)Sk + (unsigned)key, drop_p) ? keep_scale : 0.f;
      pd[e] = p * m;                                              // what multiplied V in the forward
      ds[e] = p * ((dp[e >> 2][e & 3] + cs[e]) * m - dl[e]);
    }
    uint4 pdh, pdl, dsh, dsl;
    split8(pd, pdh, pdl);
    split8(ds, dsh, dsl);
    unsigned vaddr[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) vaddr[n] = vaddr0[n] + (unsigned)(buf * BUF * 2);
    bf16x8_t th[4], tl[4];
    read_tr<1>(vaddr, th, tl);
    mma_tr(th, tl, pdh, pdl, av);                                 // dV^T[dim][key j] += dO^T drop(P)
    read_tr<0>(vaddr, th, tl);
    mma_tr(th, tl, dsh, dsl, ak);                                 // dK^T[dim][key j] += Q^T dS
  }
  __builtin_amdgcn_s_barrier();
  if (key >= Sk) return;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int dcol = h * 64 + 16 * n + 4 * g;
    const float4 k4 = make_float4(ak[n][0] * scale, ak[n][1] * scale, ak[n][2] * scale, ak[n][3] * scale);
    const float4 v4 = make_float4(av[n][0], av[n][1], av[n][2], av[n][3]);
    const size_t ko = b * dk_bs + (size_t)key * dk_rs + dcol, vo = b * dv_bs + (size_t)key * dv_rs + dcol;
    *reinterpret_cast<float4*>(dK + ko) = k4;
    *reinterpret_cast<float4*>(dV + vo) = v4;
    if (dKh) {               // planes of dK / dV (same strides as the fp32 gradients)
      uint2 hi, lo;
      split2_bf16(k4.x, k4.y, hi.x, lo.x);
      split2_bf16(k4.z, k4.w, hi.y, lo.y);
      *reinterpret_cast<uint2*>(dKh + ko) = hi;
      *reinterpret_cast<uint2*>(dKl + ko) = lo;
      split2_bf16(v4.x, v4.y, hi.x, lo.x);
      split2_bf16(v4.z, v4.w, hi.y, lo.y);
      *reinterpret_cast<uint2*>(dVh + vo) = hi;
      *reinterpret_cast<uint2*>(dVl + vo) = lo;
    }
  }
}

template <int NW, class... A>
void launch_fwd(int tiles, int heads, int B, hipStream_t st, A... a) {
  GRIDMM_LAUNCH((attention_rows_train_kernel<NW>), dim3((tiles + NW - 1) / NW, heads, B), dim3((NW + 1) * 64), 0, st, a...);
}
template <int NW, class... A>
void launch_dq(int tiles, int heads, int B, hipStream_t st, A... a) {
  GRIDMM_LAUNCH((attention_rows_bwd_dq_kernel<NW>), dim3((tiles + NW - 1) / NW, heads, B), dim3((NW + 1) * 64), 0, st, a...);
}
template <int NW, class... A>
void launch_dkv(int tiles, int heads, int B, hipStream_t st, A... a) {
  GRIDMM_LAUNCH((attention_rows_bwd_dkv_kernel<NW>), dim3((tiles + NW - 1) / NW, heads, B), dim3((NW + 1) * 64), 0, st, a...);
}
// math waves per workgroup for `tiles` 16-row tiles: no idle wave for the common sequence lengths (57 -> 4, 216 -> 2 x 7)
inline int waves_for(int tiles) { return tiles <= 4 ? 4 : (tiles <= 7 || tiles % 7 == 0 || tiles > 16 ? 7 : 8); }

}  // namespace

extern "C" int gridmm_attention_rows_train(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                                           const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo,
                                           int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, float* O, int64_t o_bs,
                                           int o_rs, void* O_hi, void* O_lo, int64_t p_bs, int p_rs, float* lse2, int Sqp,
                                           const float* vbar, int64_t vb_bs, int B,
                                           int heads, int Sq, int Sk, float scale, float dropout_p, unsigned long long seed,
                                           const unsigned long long* seed_dev, gridmm_stream_t stream) {
  if (B <= 0 || heads <= 0 || Sq <= 0 || Sk <= 0 || Sk > 2048 || !lse2 || Sqp < Sq || Sqp % 16) return GRIDMM_EINVAL;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return GRIDMM_EINVAL;
  if ((q_rs | k_rs | v_rs) & 7 || (q_bs | k_bs | v_bs) & 7) return GRIDMM_EINVAL;
  if ((!O && !O_hi) || (O_hi && (!O_lo || (p_rs & 3) || (p_bs & 3))) || (O && ((o_rs & 3) || (o_bs & 3)))) return GRIDMM_EINVAL;
  if (vbar && ((vb_bs & 3) || ((uintptr_t)vbar & 15))) return GRIDMM_EINVAL;
  const int tiles = (Sq + 15) / 16;
  hipStream_t st = as_stream(stream);
#define GRIDMM_A (const unsigned short*)Q_hi, (const unsigned short*)Q_lo, q_bs, q_rs, (const unsigned short*)K_hi,                 \
      (const unsigned short*)K_lo, k_bs, k_rs, (const unsigned short*)V_hi, (const unsigned short*)V_lo, v_bs, v_rs, kmask, mask_bs, \
      O, o_bs, o_rs, (unsigned short*)O_hi, (unsigned short*)O_lo, p_bs, p_rs, lse2, Sqp, Sq, Sk, scale, dropout_p, seed, seed_dev, \
      vbar, vb_bs
  switch (waves_for(tiles)) {
    case 4: launch_fwd<4>(tiles, heads, B, st, GRIDMM_A); break;
    case 7: launch_fwd<7>(tiles, heads, B, st, GRIDMM_A); break;
    default: launch_fwd<8>(tiles, heads, B, st, GRIDMM_A); break;
  }
#undef GRIDMM_A
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" size_t gridmm_attention_rows_bwd_workspace(int B, int heads, int Sq) {
  const size_t Sqp = (size_t)(Sq + 15) / 16 * 16;
  return 2 * (((size_t)B * heads * Sqp * 4 + 255) & ~(size_t)255) + (size_t)B * Sq * heads * 64 * 2 * 2 + 256;
}

extern "C" int gridmm_attention_rows_bwd_planes(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                                         const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo,
                                         int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, const float* O, int64_t o_bs,
                                         int o_rs, const float* dO, int64_t do_bs, int do_rs, const float* lse2,
                                         const float* vbar, int64_t vb_bs, void* workspace,
                                         size_t workspace_bytes, float* dQ, int64_t dq_bs, int dq_rs, float* dK, int64_t dk_bs,
                                         int dk_rs, float* dV, int64_t dv_bs, int dv_rs, void* dQ_hi, void* dQ_lo, void* dK_hi,
                                         void* dK_lo, void* dV_hi, void* dV_lo, int B, int heads, int Sq, int Sk, int Sqp,
                                         float scale, float dropout_p, unsigned long long seed,
                                         const unsigned long long* seed_dev, gridmm_stream_t stream) {
  if (B <= 0 || heads <= 0 || Sq <= 0 || Sk <= 0 || Sqp < Sq || Sqp % 16 || !O || !dO || !lse2 || !workspace || !dQ || !dK || !dV)
    return GRIDMM_EINVAL;
  if ((dQ_hi && !dQ_lo) || (dK_hi && (!dK_lo || !dV_hi || !dV_lo)) || (!dK_hi && dV_hi)) return GRIDMM_EINVAL;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return GRIDMM_EINVAL;
  if ((q_rs | k_rs | v_rs) & 7 || (q_bs | k_bs | v_bs) & 7) return GRIDMM_EINVAL;
  if ((o_rs | do_rs | dq_rs | dk_rs | dv_rs) & 3 || (o_bs | do_bs | dq_bs | dk_bs | dv_bs) & 3) return GRIDMM_EINVAL;
  if (workspace_bytes < gridmm_attention_rows_bwd_workspace(B, heads, Sq)) return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  const int H = heads * 64;
  if (vbar && ((vb_bs & 3) || ((uintptr_t)vbar & 15))) return GRIDMM_EINVAL;
  const size_t stat = ((size_t)B * heads * Sqp * 4 + 255) & ~(size_t)255;
  float* delta = (float*)workspace;
  float* cqv = vbar ? (float*)((char*)workspace + stat) : nullptr;
  unsigned short* dOh = (unsigned short*)((char*)workspace + 2 * stat);
  unsigned short* dOl = dOh + (size_t)B * Sq * H;
  const long rows = (long)B * Sq;
  GRIDMM_LAUNCH(attention_bwd_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dO, do_bs, do_rs, O, o_bs, o_rs, dOh,
                dOl, delta, cqv, vbar, vb_bs, heads, Sq, Sqp, B);
  GRIDMM_CHECK_LAUNCH();
#define GRIDMM_A (const unsigned short*)Q_hi, (const unsigned short*)Q_lo, q_bs, q_rs, (const unsigned short*)K_hi,                 \
      (const unsigned short*)K_lo, k_bs, k_rs, (const unsigned short*)V_hi, (const unsigned short*)V_lo, v_bs, v_rs, kmask, mask_bs, \
      (const unsigned short*)dOh, (const unsigned short*)dOl, lse2, (const float*)delta, (const float*)cqv
  const int qt = (Sq + 15) / 16, kt = (Sk + 15) / 16;
  switch (waves_for(qt)) {
    case 4: launch_dq<4>(qt, heads, B, st, GRIDMM_A, dQ, dq_bs, dq_rs, Sq, Sk, Sqp, scale, dropout_p, seed, seed_dev, (unsigned short*)dQ_hi, (unsigned short*)dQ_lo); break;
    case 7: launch_dq<7>(qt, heads, B, st, GRIDMM_A, dQ, dq_bs, dq_rs, Sq, Sk, Sqp, scale, dropout_p, seed, seed_dev, (unsigned short*)dQ_hi, (unsigned short*)dQ_lo); break;
    default: launch_dq<8>(qt, heads, B, st, GRIDMM_A, dQ, dq_bs, dq_rs, Sq, Sk, Sqp, scale, dropout_p, seed, seed_dev, (unsigned short*)dQ_hi, (unsigned short*)dQ_lo); break;
  }
  GRIDMM_CHECK_LAUNCH();
  switch (waves_for(kt)) {
    case 4: launch_dkv<4>(kt, heads, B, st, GRIDMM_A, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, Sq, Sk, Sqp, scale, dropout_p, seed, seed_dev, (unsigned short*)dK_hi, (unsigned short*)dK_lo, (unsigned short*)dV_hi, (unsigned short*)dV_lo); break;
    case 7: launch_dkv<7>(kt, heads, B, st, GRIDMM_A, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, Sq, Sk, Sqp, scale, dropout_p, seed, seed_dev, (unsigned short*)dK_hi, (unsigned short*)dK_lo, (unsigned short*)dV_hi, (unsigned short*)dV_lo); break;
    default: launch_dkv<8>(kt, heads, B, st, GRIDMM_A, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, Sq, Sk, Sqp, scale, dropout_p, seed, seed_dev, (unsigned short*)dK_hi, (unsigned short*)dK_lo, (unsigned short*)dV_hi, (unsigned short*)dV_lo); break;
  }
#undef GRIDMM_A
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_attention_rows_bwd(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                                         const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo,
                                         int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, const float* O, int64_t o_bs,
                                         int o_rs, const float* dO, int64_t do_bs, int do_rs, const float* lse2,
                                         const float* vbar, int64_t vb_bs, void* workspace,
                                         size_t workspace_bytes, float* dQ, int64_t dq_bs, int dq_rs, float* dK, int64_t dk_bs,
                                         int dk_rs, float* dV, int64_t dv_bs, int dv_rs, int B, int heads, int Sq, int Sk, int Sqp,
                                         float scale, float dropout_p, unsigned long long seed,
                                         const unsigned long long* seed_dev, gridmm_stream_t stream) {
  return gridmm_attention_rows_bwd_planes(Q_hi, Q_lo, q_bs, q_rs, K_hi, K_lo, k_bs, k_rs, V_hi, V_lo, v_bs, v_rs, kmask, mask_bs, O,
                                          o_bs, o_rs, dO, do_bs, do_rs, lse2, vbar, vb_bs, workspace, workspace_bytes, dQ, dq_bs,
                                          dq_rs, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs, nullptr, nullptr, nullptr, nullptr, nullptr,
                                          nullptr, B, heads, Sq, Sk, Sqp, scale, dropout_p, seed, seed_dev, stream);
}
