// gridmm_attention: multi-head attention core (head_dim 64) in exact fp32 on the f32 matrix
// pipe (v_mfma_f32_16x16x4_f32), online softmax, masked keys contribute exactly 0.
//
// One 64-lane wave owns a (batch, head, 16-query tile) and walks the keys 16 at a time, fully
// in registers (sequences here are <= ~500 keys and L2-resident, so no LDS staging):
//   S^T = K_tile Q^T      16 MFMAs   lane (j = lane&15, g = lane>>4) ends with S^T[key 4g+r][query j]
//   softmax statistics    per query j: in-lane over r, then across g with two xor-shuffles
//   O  += P V_tile        16 MFMAs   A operand = the S^T registers as they are (key(g,s) = 4g+s on
//                                    both operands), output column j of tile n <-> head dim 4j+n,
//                                    so Q/K/V/O all move as 128-bit accesses.
// The k index of every MFMA is a free permutation (sums commute), which is what lets all four
// tensors be read as float4: for S^T the 64 head dims are visited as d = 16s' + 4g + e.
#include "common.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;

__global__ __launch_bounds__(256) void attention_kernel(
    const float* __restrict__ Q, int64_t q_bs, int q_rs, const float* __restrict__ K, int64_t k_bs,
    int k_rs, const float* __restrict__ V, int64_t v_bs, int v_rs, const uint8_t* __restrict__ kmask,
    int mask_bs, float* __restrict__ O, int64_t o_bs, int o_rs, unsigned short* __restrict__ Ohi,
    unsigned short* __restrict__ Olo, int64_t p_bs, int p_rs, int Sq, int Sk, float scale,
    float* __restrict__ lse, int Sqp, float drop_p, unsigned long long seed,
    const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;   // per-replay part of the seed (captured training steps)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = (blockIdx.x * 4 + wave) * 16;
  if (q0 >= Sq) return;
  const int h = blockIdx.y, b = blockIdx.z;
  const int j = lane & 15, g = lane >> 4;

  const float* qrow = Q + b * q_bs + (size_t)min(q0 + j, Sq - 1) * q_rs + h * 64 + 4 * g;
  float4 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float4 v = *reinterpret_cast<const float4*>(qrow + 16 * s);
    qf[s] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
  }

  f32x4_t o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  const float* Kb = K + b * k_bs + h * 64 + 4 * g;
  const float* Vb = V + b * v_bs + h * 64 + 4 * j;
  const uint8_t* mb = kmask ? kmask + (size_t)b * mask_bs : nullptr;

  // K/V fragments of one 16-key tile; the next tile is fetched while the current one is in the matrix pipe
  auto load_kv = [&](int key0, float4 (&kf)[4], float4 (&vf)[4]) {
    const float* krow = Kb + (size_t)min(key0 + j, Sk - 1) * k_rs;
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const float4*>(krow + 16 * s);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      vf[s] = *reinterpret_cast<const float4*>(Vb + (size_t)min(key0 + 4 * g + s, Sk - 1) * v_rs);
  };
  auto tile_valid = [&](int key0) -> bool {   // wave-uniform: any unmasked key in the tile
    if (key0 >= Sk) return false;
    if (!mb) return true;
    const int kk = key0 + j;
    return __any((kk < Sk) && mb[kk]);
  };
  auto next_valid = [&](int key0) -> int {    // first tile start >= key0 with an unmasked key (or >= Sk)
    while (key0 < Sk && !tile_valid(key0)) key0 += 16;
    return key0;
  };

  float4 kf[4], vf[4], kn[4], vn[4];
  int key0 = next_valid(0);
  if (key0 < Sk) load_kv(key0, kf, vf);
  while (key0 < Sk) {
    const int key1 = next_valid(key0 + 16);
    if (key1 < Sk) load_kv(key1, kn, vn);      // prefetch (register double buffer)

    f32x4_t st = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].x, qf[s].x, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].y, qf[s].y, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].z, qf[s].z, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].w, qf[s].w, st, 0, 0, 0);
    }

    bool valid[4];
    float mx = NEG_BIG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = key0 + 4 * g + r;
      valid[r] = (kk < Sk) && (!mb || mb[kk]);
      mx = fmaxf(mx, valid[r] ? st[r] : NEG_BIG);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);  // 0 on the first valid tile (m_run = -1e30)
    float p[4], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = valid[r] ? expf(st[r] - m_new) : 0.f;
      ps += p[r];
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;

    if (drop_p > 0.f) {   // dropout on the probabilities (vilmodel.py:143): the row sum above stays un-dropped
      const float keep_scale = 1.0f / (1.0f - drop_p);
      const unsigned int row = ((unsigned int)(b * gridDim.y + h) * Sq + (q0 + j)) * Sk;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        p[r] = dropout_keep(seed, row + key0 + 4 * g + r, drop_p) ? p[r] * keep_scale : 0.f;
    }
    // rescale the output rows: row (query) 4g+r takes alpha from the lane whose j == 4g+r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
      for (int n = 0; n < 4; ++n) o[n][r] *= a;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[s], vf[s].x, o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[s], vf[s].y, o[1], 0, 0, 0);
      o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[s], vf[s].z, o[2], 0, 0, 0);
      o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[s], vf[s].w, o[3], 0, 0, 0);
    }
    key0 = key1;
#pragma unroll
    for (int s = 0; s < 4; ++s) { kf[s] = kn[s]; vf[s] = vn[s]; }
  }

  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  // training: log-sum-exp of the scaled scores per query (+BIG for a fully masked row, so that the
  // backward's exp(s - lse) is exactly 0 there)
  if (lse && g == 0 && q0 + j < Sqp)
    lse[((size_t)b * gridDim.y + h) * Sqp + q0 + j] = l_run > 0.f ? m_run + logf(l_run) : -NEG_BIG;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a = __shfl(inv, 4 * g + r, 64);
    const int q = q0 + 4 * g + r;
    if (q < Sq) {
      const float x[4] = {o[0][r] * a, o[1][r] * a, o[2][r] * a, o[3][r] * a};
      if (O) *reinterpret_cast<float4*>(O + b * o_bs + (size_t)q * o_rs + h * 64 + 4 * j) = make_float4(x[0], x[1], x[2], x[3]);
      if (Ohi) {
        u16x4_t hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned short hh = f32_to_bf16_rne(x[e]);
          hi[e] = hh;
          lo[e] = f32_to_bf16_rne(x[e] - bf16_bits_to_f32(hh));
        }
        *reinterpret_cast<u16x4_t*>(Ohi + b * p_bs + (size_t)q * p_rs + h * 64 + 4 * j) = hi;
        *reinterpret_cast<u16x4_t*>(Olo + b * p_bs + (size_t)q * p_rs + h * 64 + 4 * j) = lo;
      }
    }
  }
}

}  // namespace

extern "C" int gridmm_attention(const float* Q, int64_t q_bs, int q_rs, const float* K, int64_t k_bs,
                                int k_rs, const float* V, int64_t v_bs, int v_rs, const uint8_t* kmask,
                                int mask_bs, float* O, int64_t o_bs, int o_rs, void* O_hi, void* O_lo,
                                int64_t p_bs, int p_rs, int B, int heads, int Sq, int Sk, float scale,
                                gridmm_stream_t stream) {
  if (B <= 0 || heads <= 0 || Sq <= 0 || Sk <= 0) return GRIDMM_EINVAL;
  if ((!O && !O_hi) || (O_hi && (!O_lo || (p_rs & 3) || (p_bs & 3)))) return GRIDMM_EINVAL;
  if ((q_rs | k_rs | v_rs | o_rs) & 3) return GRIDMM_EINVAL;
  if ((q_bs | k_bs | v_bs | o_bs) & 3) return GRIDMM_EINVAL;
  dim3 grid((Sq + 63) / 64, heads, B), block(256);
  GRIDMM_LAUNCH(attention_kernel, grid, block, 0, as_stream(stream), Q, q_bs, q_rs, K, k_bs, k_rs, V,
                     v_bs, v_rs, kmask, mask_bs, O, o_bs, o_rs, (unsigned short*)O_hi, (unsigned short*)O_lo, p_bs,
                     p_rs, Sq, Sk, scale, (float*)nullptr, 0, 0.f, 0ull, (const unsigned long long*)nullptr);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

// _planes: also the bf16 hi/lo planes of O (row stride p_rs, episode stride p_bs) -- the A operand of the output
// projection and, in the backward, an operand of its weight gradient (gridmm_linear_planes_tn): no split pass over O.
extern "C" int gridmm_attention_train_planes(const float* Q, int64_t q_bs, int q_rs, const float* K, int64_t k_bs,
                                             int k_rs, const float* V, int64_t v_bs, int v_rs, const uint8_t* kmask,
                                             int mask_bs, float* O, int64_t o_bs, int o_rs, void* O_hi, void* O_lo,
                                             int64_t p_bs, int p_rs, float* lse, int Sqp, int B, int heads, int Sq, int Sk,
                                             float scale, float dropout_p, unsigned long long seed,
                                             const unsigned long long* seed_dev, gridmm_stream_t stream) {
  if (B <= 0 || heads <= 0 || Sq <= 0 || Sk <= 0 || !O || !lse || Sqp < Sq || Sqp % 16) return GRIDMM_EINVAL;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return GRIDMM_EINVAL;
  if ((q_rs | k_rs | v_rs | o_rs) & 3) return GRIDMM_EINVAL;
  if ((q_bs | k_bs | v_bs | o_bs) & 3) return GRIDMM_EINVAL;
  if (O_hi && (!O_lo || (p_rs & 3) || (p_bs & 3))) return GRIDMM_EINVAL;
  dim3 grid((Sq + 63) / 64, heads, B), block(256);
  GRIDMM_LAUNCH(attention_kernel, grid, block, 0, as_stream(stream), Q, q_bs, q_rs, K, k_bs, k_rs, V,
                     v_bs, v_rs, kmask, mask_bs, O, o_bs, o_rs, (unsigned short*)O_hi, (unsigned short*)O_lo,
                     p_bs, p_rs, Sq, Sk, scale, lse, Sqp, dropout_p, seed, seed_dev);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_attention_train(const float* Q, int64_t q_bs, int q_rs, const float* K, int64_t k_bs,
                                      int k_rs, const float* V, int64_t v_bs, int v_rs, const uint8_t* kmask,
                                      int mask_bs, float* O, int64_t o_bs, int o_rs, float* lse, int Sqp, int B,
                                      int heads, int Sq, int Sk, float scale, float dropout_p,
                                      unsigned long long seed, const unsigned long long* seed_dev,
                                      gridmm_stream_t stream) {
  return gridmm_attention_train_planes(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, kmask, mask_bs, O, o_bs, o_rs, nullptr,
                                       nullptr, 0, 0, lse, Sqp, B, heads, Sq, Sk, scale, dropout_p, seed, seed_dev, stream);
}

// ================================================================================================
// bf16x3 attention (the hot-path one): the same online-softmax walk, but S^T = K Q^T and O += P V run on
// the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, 3-term hi/lo split, fp32 accumulate): 24 MFMAs of
// 16 cycles per 32 keys instead of 64 f32 MFMAs of 32 cycles.
//   * Q and K arrive as the hi/lo planes the QKV GEMM already emits.
//   * V arrives re-tiled per head by gridmm_transpose_v: VT[b][h][key tile of 32][d][32 slots], slot 8g+e
//     holding key 4g+e (e<4) / 16+4g+(e-4) -- exactly the keys whose probabilities lane (.,g) already has
//     in its two S^T accumulators, so P feeds the second MFMA without any cross-lane movement and the B
//     operand of P V is ONE 16-byte load per lane.  Output head dim d = 4j+n keeps stores 64/128-bit.
//   * Sequences here are short (57..296), so parallelism comes from splitting the KEYS of one 16-query
//     tile over the 4 waves of a workgroup (32-key tiles round-robin) and merging the partial
//     (max, sum, O) through LDS -- 4x more waves in flight, 4x shorter dependent chains.
// ================================================================================================
namespace {

__global__ __launch_bounds__(256) void transpose_v_kernel(const unsigned short* __restrict__ Vh,
                                                          const unsigned short* __restrict__ Vl, int64_t v_bs,
                                                          int v_rs, unsigned short* __restrict__ Th,
                                                          unsigned short* __restrict__ Tl, int heads, int Sk,
                                                          int Skp) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[2][32][72];   // [plane][key][d], 144-B rows
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  {   // 128-bit loads: thread -> (key = tid / 8, head dims 8 * (tid % 8) .. +8)
    const int key = tid >> 3, dc = (tid & 7) * 8;
    uint4 a = make_uint4(0u, 0u, 0u, 0u), c = a;
    if (kt * 32 + key < Sk) {
      const size_t o = b * v_bs + (size_t)(kt * 32 + key) * v_rs + h * 64 + dc;
      a = *reinterpret_cast<const uint4*>(Vh + o);
      c = *reinterpret_cast<const uint4*>(Vl + o);
    }
    *reinterpret_cast<uint4*>(&tile[0][key][dc]) = a;
    *reinterpret_cast<uint4*>(&tile[1][key][dc]) = c;
  }
  __syncthreads();
  const int d = tid & 63, g = tid >> 6;
  u16x8_t a, c;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int key = e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4);
    a[e] = tile[0][key][d];
    c[e] = tile[1][key][d];
  }
  const size_t o = ((((size_t)b * heads + h) * (Skp / 32) + kt) * 64 + d) * 32 + 8 * g;
  *reinterpret_cast<u16x8_t*>(Th + o) = a;
  *reinterpret_cast<u16x8_t*>(Tl + o) = c;
}

__device__ __forceinline__ bf16x8_t ld8(const unsigned short* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

// NQ query tiles (16 queries each) per wave share every K / V fragment load -- K/V re-reads from L2 were
// the bottleneck of the one-tile-per-wave version (Sq/16 re-reads of every head's keys).
// KSPLIT = 4: the workgroup covers <= NQ query tiles; its 4 waves split the KEYS (32-key tiles round-robin)
//             and merge (max, sum, O) through LDS  -> short sequences (Sq <= 64) still fill the chip.
// KSPLIT = 1: wave w of workgroup x owns query tiles (4x + w) * NQ .. +NQ and walks all keys.
template <int NQ, int KSPLIT>
__global__ __launch_bounds__(256) void attention_planes_kernel(
    const unsigned short* __restrict__ Qh, const unsigned short* __restrict__ Ql, int64_t q_bs, int q_rs,
    const unsigned short* __restrict__ Kh, const unsigned short* __restrict__ Kl, int64_t k_bs, int k_rs,
    const unsigned short* __restrict__ Th, const unsigned short* __restrict__ Tl, int Skp,
    const uint8_t* __restrict__ kmask, int mask_bs, float* __restrict__ O, int64_t o_bs, int o_rs,
    unsigned short* __restrict__ Ohi, unsigned short* __restrict__ Olo, int64_t p_bs, int p_rs, int heads,
    int Sq, int Sk, float scale) {
  constexpr int MQ = (KSPLIT == 4) ? NQ * 16 : 1;
  __shared__ __attribute__((aligned(16))) float s_o[KSPLIT == 4 ? 4 : 1][MQ][KSPLIT == 4 ? 64 : 4];
  __shared__ float s_m[4][MQ], s_l[4][MQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qt0 = (KSPLIT == 4 ? blockIdx.x : blockIdx.x * 4 + wave) * NQ;   // first query tile of this wave
  const int h = blockIdx.y, b = blockIdx.z;
  const int j = lane & 15, g = lane >> 4;
  // Key-validity bits of the whole row, one 32-bit word per 32-key tile (Sk <= 512), built once per wave from
  // independent byte loads: inside the key loop the mask is ALU work, not 8 dependent global loads per tile.
  __shared__ unsigned s_mw[4][16];
  {
    const uint8_t* mrow = kmask ? kmask + (size_t)b * mask_bs : nullptr;
    for (int i = 0; i < (Sk + 63) >> 6; ++i) {
      const int k = i * 64 + lane;
      const unsigned long long bal = __ballot((k < Sk) && (!mrow || mrow[k]));
      if (lane == 0) { s_mw[wave][2 * i] = (unsigned)bal; s_mw[wave][2 * i + 1] = (unsigned)(bal >> 32); }
    }
  }
  __syncthreads();
  if (KSPLIT == 1 && qt0 * 16 >= Sq) return;

  // Q^T as the B operand: lane (query j, k-chunk g) holds head dims 32ks + 8g .. +8
  bf16x8_t qh[NQ][2], ql[NQ][2];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const size_t qo = b * q_bs + (size_t)min((qt0 + t) * 16 + j, Sq - 1) * q_rs + h * 64 + 8 * g;
    qh[t][0] = ld8(Qh + qo); qh[t][1] = ld8(Qh + qo + 32);
    ql[t][0] = ld8(Ql + qo); ql[t][1] = ld8(Ql + qo + 32);
  }
  f32x4_t o[NQ][4];
  float m_run[NQ], l_run[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    m_run[t] = NEG_BIG; l_run[t] = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) o[t][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const unsigned short* Kbh = Kh + b * k_bs + h * 64 + 8 * g;
  const unsigned short* Kbl = Kl + b * k_bs + h * 64 + 8 * g;
  const size_t tbase = ((size_t)b * heads + h) * (Skp / 32) * 64 * 32 + (size_t)(4 * j) * 32 + 8 * g;

  // K / V fragments of one 32-key tile.  Two register sets ping-pong: the next (unmasked) tile is fetched while the
  // current one is in the matrix pipe -- at NQ = 4 the kernel runs one wave per SIMD, so nothing else hides the loads.
  struct KV { bf16x8_t kh[2][2], kl[2][2], vh[4], vl[4]; };
  auto load_tile = [&](int key0, KV& t) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t ko = (size_t)min(key0 + 16 * u + j, Sk - 1) * k_rs;
      t.kh[u][0] = ld8(Kbh + ko); t.kh[u][1] = ld8(Kbh + ko + 32);
      t.kl[u][0] = ld8(Kbl + ko); t.kl[u][1] = ld8(Kbl + ko + 32);
    }
    const size_t to = tbase + (size_t)(key0 >> 5) * 64 * 32;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      t.vh[n] = ld8(Th + to + n * 32);
      t.vl[n] = ld8(Tl + to + n * 32);
    }
  };
  auto next_tile = [&](int key0) -> int {   // first 32-key tile >= key0 (this wave's stride) with a valid key
    while (key0 < Sk && s_mw[wave][key0 >> 5] == 0u) key0 += 32 * KSPLIT;
    return key0;
  };
  auto compute_tile = [&](int key0, const KV& kv) {
    const auto& kh = kv.kh; const auto& kl = kv.kl; const auto& vh = kv.vh; const auto& vl = kv.vl;
    bool valid[8];
    const unsigned mword = s_mw[wave][key0 >> 5];
#pragma unroll
    for (int e = 0; e < 8; ++e) valid[e] = (mword >> (16 * (e >> 2) + 4 * g + (e & 3))) & 1u;
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      if ((qt0 + t) * 16 >= Sq) continue;          // wave-uniform
      f32x4_t st[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl[u][0], qh[t][0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh[u][0], ql[t][0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl[u][1], qh[t][1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh[u][1], ql[t][1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh[u][0], qh[t][0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh[u][1], qh[t][1], s, 0, 0, 0);
        st[u] = s;
      }
      float sv[8], mx = NEG_BIG;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sv[e] = st[e >> 2][e & 3] * scale;
        mx = fmaxf(mx, valid[e] ? sv[e] : NEG_BIG);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[t], mx);
      const float alpha = __expf(m_run[t] - m_new);
      float ps = 0.f, p[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p[e] = valid[e] ? __expf(sv[e] - m_new) : 0.f;
        ps += p[e];
      }
      uint4 ph, pl;
      split2_bf16(p[0], p[1], ph.x, pl.x);
      split2_bf16(p[2], p[3], ph.y, pl.y);
      split2_bf16(p[4], p[5], ph.z, pl.z);
      split2_bf16(p[6], p[7], ph.w, pl.w);
      const bf16x8_t pah = __builtin_bit_cast(bf16x8_t, ph), pal = __builtin_bit_cast(bf16x8_t, pl);
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      l_run[t] = l_run[t] * alpha + ps;
      m_run[t] = m_new;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
        for (int n = 0; n < 4; ++n) o[t][n][r] *= a;
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        o[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pal, vh[n], o[t][n], 0, 0, 0);
        o[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pah, vl[n], o[t][n], 0, 0, 0);
        o[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pah, vh[n], o[t][n], 0, 0, 0);
      }
    }
  };

  KV ta, tb;
  int key0 = next_tile(KSPLIT == 4 ? wave * 32 : 0);
  if (key0 < Sk) load_tile(key0, ta);
  while (key0 < Sk) {
    int key1 = next_tile(key0 + 32 * KSPLIT);
    if (key1 < Sk) load_tile(key1, tb);
    compute_tile(key0, ta);
    if (key1 >= Sk) break;
    key0 = next_tile(key1 + 32 * KSPLIT);
    if (key0 < Sk) load_tile(key0, ta);
    compute_tile(key1, tb);
  }

  auto store4 = [&](int q, int dcol, const float (&x)[4]) {
    if (O) *reinterpret_cast<float4*>(O + b * o_bs + (size_t)q * o_rs + h * 64 + dcol) = make_float4(x[0], x[1], x[2], x[3]);
    if (Ohi) {
      uint2 hi, lo;
      split2_bf16(x[0], x[1], hi.x, lo.x);
      split2_bf16(x[2], x[3], hi.y, lo.y);
      *reinterpret_cast<uint2*>(Ohi + b * p_bs + (size_t)q * p_rs + h * 64 + dcol) = hi;
      *reinterpret_cast<uint2*>(Olo + b * p_bs + (size_t)q * p_rs + h * 64 + dcol) = lo;
    }
  };

  if (KSPLIT == 1) {   // finish from registers: lane (j,g) holds O[q = 4g+r][d = 4j+n]
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      const float inv = l_run[t] > 0.f ? 1.0f / l_run[t] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = __shfl(inv, 4 * g + r, 64);
        const int q = (qt0 + t) * 16 + 4 * g + r;
        if (q < Sq) {
          const float x[4] = {o[t][0][r] * a, o[t][1][r] * a, o[t][2][r] * a, o[t][3][r] * a};
          store4(q, 4 * j, x);
        }
      }
    }
    return;
  }
  // ---- merge the 4 key-partitions through LDS
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(&s_o[wave][t * 16 + 4 * g + r][4 * j]) =
          make_float4(o[t][0][r], o[t][1][r], o[t][2][r], o[t][3][r]);
    if (g == 0) { s_m[wave][t * 16 + j] = m_run[t]; s_l[wave][t * 16 + j] = l_run[t]; }
  }
  __syncthreads();
  // thread -> (query ql = tid / 16 + 16 * pass, dims 4 * (tid % 16) .. +4)
  for (int ql = threadIdx.x >> 4; ql < NQ * 16; ql += 16) {
    const int dc = 4 * (threadIdx.x & 15);
    float M = NEG_BIG;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, s_m[w][ql]);
    float L = 0.f, x[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = __expf(s_m[w][ql] - M);   // 0 for a partition that saw no valid key, unless all did
      L += s_l[w][ql] * f;
      const float4 v = *reinterpret_cast<const float4*>(&s_o[w][ql][dc]);
      x[0] += v.x * f; x[1] += v.y * f; x[2] += v.z * f; x[3] += v.w * f;
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    const int qq = qt0 * 16 + ql;
    if (qq < Sq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] *= inv;
      store4(qq, dc, x);
    }
  }
}

}  // namespace

extern "C" int gridmm_transpose_v(const void* V_hi, const void* V_lo, int64_t v_bs, int v_rs, void* T_hi,
                                  void* T_lo, int B, int heads, int Sk, int Skp, gridmm_stream_t stream) {
  if (B <= 0 || heads <= 0 || Sk <= 0 || Skp < Sk || Skp % 32) return GRIDMM_EINVAL;
  dim3 grid(Skp / 32, heads, B), block(256);
  GRIDMM_LAUNCH(transpose_v_kernel, grid, block, 0, as_stream(stream), (const unsigned short*)V_hi,
                (const unsigned short*)V_lo, v_bs, v_rs, (unsigned short*)T_hi, (unsigned short*)T_lo, heads, Sk, Skp);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_attention_planes(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                                       const void* K_lo, int64_t k_bs, int k_rs, const void* T_hi, const void* T_lo,
                                       int Skp, const uint8_t* kmask, int mask_bs, float* O, int64_t o_bs, int o_rs,
                                       void* O_hi, void* O_lo, int64_t p_bs, int p_rs, int B, int heads, int Sq, int Sk,
                                       float scale, gridmm_stream_t stream) {
  if (B <= 0 || heads <= 0 || Sq <= 0 || Sk <= 0 || Skp < Sk || Skp % 32 || Sk > 512) return GRIDMM_EINVAL;   // mask words: 16 x 32 keys
  if ((q_rs | k_rs) & 7 || (q_bs | k_bs) & 7) return GRIDMM_EINVAL;
  if ((!O && !O_hi) || (O_hi && (!O_lo || (p_rs & 3) || (p_bs & 3))) || (O && ((o_rs & 3) || (o_bs & 3))))
    return GRIDMM_EINVAL;
  const int nqt = (Sq + 15) / 16;
#define GRIDMM_ATT_ARGS                                                                                              \
  (const unsigned short*)Q_hi, (const unsigned short*)Q_lo, q_bs, q_rs, (const unsigned short*)K_hi,                \
      (const unsigned short*)K_lo, k_bs, k_rs, (const unsigned short*)T_hi, (const unsigned short*)T_lo, Skp, kmask, \
      mask_bs, O, o_bs, o_rs, (unsigned short*)O_hi, (unsigned short*)O_lo, p_bs, p_rs, heads, Sq, Sk, scale
  if (nqt <= 4) {   // short query side: every wave sees all (<= 4) query tiles and a quarter of the keys
    dim3 grid(1, heads, B), block(256);
    if (nqt <= 2) GRIDMM_LAUNCH((attention_planes_kernel<2, 4>), grid, block, 0, as_stream(stream), GRIDMM_ATT_ARGS);
    else GRIDMM_LAUNCH((attention_planes_kernel<4, 4>), grid, block, 0, as_stream(stream), GRIDMM_ATT_ARGS);
  } else {          // 4 query tiles per wave, 16 per workgroup
    dim3 grid((nqt + 15) / 16, heads, B), block(256);
    GRIDMM_LAUNCH((attention_planes_kernel<4, 1>), grid, block, 0, as_stream(stream), GRIDMM_ATT_ARGS);
  }
#undef GRIDMM_ATT_ARGS
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
