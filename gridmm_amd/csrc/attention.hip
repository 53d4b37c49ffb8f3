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
    unsigned short* __restrict__ Olo, int64_t p_bs, int p_rs, int Sq, int Sk, float scale) {
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
                     p_rs, Sq, Sk, scale);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
