// Backward of gridmm_attention_train (softmax(Q K^T * scale + key mask) V, head_dim 64), exact fp32 on the
// f32 matrix pipe (v_mfma_f32_16x16x4_f32), probabilities recomputed from the saved log-sum-exp.
//
//   P  = exp(S - lse)                 dP = dO V^T            delta_q = sum_d dO[q][d] O[q][d]
//   dS = P o (dP - delta)             dQ = scale dS K        dK = scale dS^T Q        dV = P^T dO
//
// Two kernels, both register-resident like the forward (sequences <= ~500, everything L2-resident):
//   attention_bwd_dq_kernel : a wave owns 16 queries, walks the keys      (S^T tiles: lane (query j, g) holds keys 4g+r)
//   attention_bwd_dkv_kernel: a wave owns 16 keys,    walks the queries   (S   tiles: lane (key j,   g) holds queries 4g+r)
// In both, the tile a lane ends up holding is exactly the A operand of the second contraction (k index = the
// 4 rows a lane holds, same trick as P V in the forward), so nothing moves across lanes.
#include "common.h"

namespace {


// rows-style fragment (A operand "row = lane&15"): 4 float4 at head dims 16s + 4g
__device__ __forceinline__ void load_rows(const float* base, size_t row_off, int g, float4 (&f)[4]) {
  const float* p = base + row_off + 4 * g;
#pragma unroll
  for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const float4*>(p + 16 * s);
}
// cols-style fragment (B operand of the second contraction): rows 4g+s, head dims 4j .. 4j+3
__device__ __forceinline__ void load_cols(const float* base, int row0, int rs, int nrows, int g, int j,
                                          float4 (&f)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
    f[s] = *reinterpret_cast<const float4*>(base + (size_t)min(row0 + 4 * g + s, nrows - 1) * rs + 4 * j);
}
__device__ __forceinline__ f32x4_t mma16(const float4 (&a)[4], const float4 (&b)[4]) {
  f32x4_t c = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].x, b[s].x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].y, b[s].y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].z, b[s].z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].w, b[s].w, c, 0, 0, 0);
  }
  return c;
}
__device__ __forceinline__ void mma_acc(f32x4_t (&o)[4], const float (&a)[4], const float4 (&b)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s].x, o[0], 0, 0, 0);
    o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s].y, o[1], 0, 0, 0);
    o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s].z, o[2], 0, 0, 0);
    o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s].w, o[3], 0, 0, 0);
  }
}

// delta[b][h][q] = sum_d dO[b][q][h][d] * O[b][q][h][d]; one thread per (b, h, q)
__global__ void attention_delta_kernel(const float* __restrict__ dO, int64_t do_bs, int do_rs,
                                       const float* __restrict__ O, int64_t o_bs, int o_rs,
                                       float* __restrict__ delta, int heads, int Sq, int Sqp, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * heads * Sqp) return;
  const int q = i % Sqp, h = (i / Sqp) % heads, b = i / ((size_t)Sqp * heads);
  float s = 0.f;
  if (q < Sq) {
    const float4* a = reinterpret_cast<const float4*>(dO + b * do_bs + (size_t)q * do_rs + h * 64);
    const float4* c = reinterpret_cast<const float4*>(O + b * o_bs + (size_t)q * o_rs + h * 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 x = a[k], y = c[k];
      s += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
  }
  delta[i] = s;
}

__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(
    const float* __restrict__ Q, int64_t q_bs, int q_rs, const float* __restrict__ K, int64_t k_bs, int k_rs,
    const float* __restrict__ V, int64_t v_bs, int v_rs, const uint8_t* __restrict__ kmask, int mask_bs,
    const float* __restrict__ dO, int64_t do_bs, int do_rs, const float* __restrict__ lse,
    const float* __restrict__ delta, float* __restrict__ dQ, int64_t dq_bs, int dq_rs, int Sq, int Sk, int Sqp,
    float scale, float drop_p, unsigned long long seed,
    const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = (blockIdx.x * 4 + wave) * 16;
  if (q0 >= Sq) return;
  const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  const int j = lane & 15, g = lane >> 4;
  const int qj = min(q0 + j, Sq - 1);

  float4 qf[4], dof[4];
  load_rows(Q + b * q_bs + h * 64, (size_t)qj * q_rs, g, qf);
  load_rows(dO + b * do_bs + h * 64, (size_t)qj * do_rs, g, dof);
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = make_float4(qf[s].x * scale, qf[s].y * scale, qf[s].z * scale, qf[s].w * scale);
  const size_t st_off = ((size_t)b * heads + h) * Sqp;
  const float lse_j = lse[st_off + q0 + j], delta_j = delta[st_off + q0 + j];

  f32x4_t acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float* Kb = K + b * k_bs + h * 64;
  const float* Vb = V + b * v_bs + h * 64;
  const uint8_t* mb = kmask ? kmask + (size_t)b * mask_bs : nullptr;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;

  for (int key0 = 0; key0 < Sk; key0 += 16) {
    if (mb) {
      const int kk = key0 + j;
      if (!__any((kk < Sk) && mb[kk])) continue;
    }
    float4 kf[4], vf[4], kc[4];
    const int kj = min(key0 + j, Sk - 1);
    load_rows(Kb, (size_t)kj * k_rs, g, kf);
    load_rows(Vb, (size_t)kj * v_rs, g, vf);
    load_cols(Kb, key0, k_rs, Sk, g, j, kc);
    const f32x4_t st = mma16(kf, qf);     // S^T[key 4g+r][query j] (scaled)
    const f32x4_t dpt = mma16(vf, dof);   // dP^T[key 4g+r][query j]
    float ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = key0 + 4 * g + r;
      const bool valid = (kk < Sk) && (!mb || mb[kk]);
      const float p = valid ? expf(st[r] - lse_j) : 0.f;
      float dp = dpt[r];
      if (drop_p > 0.f)   // d/dP of the dropped probabilities: mask / (1 - p); delta = dO.O already includes it
        dp = dropout_keep(seed, ((unsigned int)(b * heads + h) * Sq + (q0 + j)) * Sk + kk, drop_p) ? dp * keep_scale : 0.f;
      ds[r] = p * (dp - delta_j);
    }
    // dQ[query][d] += sum_key dS[query][key] K[key][d]: A needs lane (i = query, k = g) -> the lane holding
    // query i's column is (j = i, g) itself, with keys 4g+s as its 4 k-steps.
    mma_acc(acc, ds, kc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * g + r;
    if (q < Sq)
      *reinterpret_cast<float4*>(dQ + b * dq_bs + (size_t)q * dq_rs + h * 64 + 4 * j) =
          make_float4(acc[0][r] * scale, acc[1][r] * scale, acc[2][r] * scale, acc[3][r] * scale);
  }
}

__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(
    const float* __restrict__ Q, int64_t q_bs, int q_rs, const float* __restrict__ K, int64_t k_bs, int k_rs,
    const float* __restrict__ V, int64_t v_bs, int v_rs, const uint8_t* __restrict__ kmask, int mask_bs,
    const float* __restrict__ dO, int64_t do_bs, int do_rs, const float* __restrict__ lse,
    const float* __restrict__ delta, float* __restrict__ dK, int64_t dk_bs, int dk_rs, float* __restrict__ dV,
    int64_t dv_bs, int dv_rs, int Sq, int Sk, int Sqp, float scale, float drop_p, unsigned long long seed,
    const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int key0 = (blockIdx.x * 4 + wave) * 16;
  if (key0 >= Sk) return;
  const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  const int j = lane & 15, g = lane >> 4;
  const int kj = min(key0 + j, Sk - 1);
  const uint8_t* mb = kmask ? kmask + (size_t)b * mask_bs : nullptr;
  const bool key_valid = (key0 + j < Sk) && (!mb || mb[key0 + j]);

  float4 kf[4], vf[4];
  load_rows(K + b * k_bs + h * 64, (size_t)kj * k_rs, g, kf);
  load_rows(V + b * v_bs + h * 64, (size_t)kj * v_rs, g, vf);
#pragma unroll
  for (int s = 0; s < 4; ++s) kf[s] = make_float4(kf[s].x * scale, kf[s].y * scale, kf[s].z * scale, kf[s].w * scale);

  f32x4_t ak[4], av[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) { ak[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; av[n] = ak[n]; }
  const float* Qb = Q + b * q_bs + h * 64;
  const float* dOb = dO + b * do_bs + h * 64;
  const size_t st_off = ((size_t)b * heads + h) * Sqp;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;

  if (__any(key_valid)) {
    for (int q0 = 0; q0 < Sq; q0 += 16) {
      float4 qf[4], dof[4], qc[4], doc[4];
      const int qj = min(q0 + j, Sq - 1);
      load_rows(Qb, (size_t)qj * q_rs, g, qf);
      load_rows(dOb, (size_t)qj * do_rs, g, dof);
      load_cols(Qb, q0, q_rs, Sq, g, j, qc);
      load_cols(dOb, q0, do_rs, Sq, g, j, doc);
      const float4 l4 = *reinterpret_cast<const float4*>(lse + st_off + q0 + 4 * g);
      const float4 d4 = *reinterpret_cast<const float4*>(delta + st_off + q0 + 4 * g);
      const f32x4_t st = mma16(qf, kf);    // S[query 4g+r][key j] (scaled)
      const f32x4_t dp = mma16(dof, vf);   // dP[query 4g+r][key j]
      const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
      float p[4], ds[4], pd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool valid = key_valid && (q0 + 4 * g + r < Sq);
        p[r] = valid ? expf(st[r] - ls[r]) : 0.f;
        float m = 1.0f;
        if (drop_p > 0.f)
          m = dropout_keep(seed, ((unsigned int)(b * heads + h) * Sq + (q0 + 4 * g + r)) * Sk + key0 + j, drop_p)
                  ? keep_scale : 0.f;
        pd[r] = p[r] * m;                       // dropped probabilities (what multiplied V in the forward)
        ds[r] = p[r] * (dp[r] * m - dl[r]);
      }
      mma_acc(av, pd, doc);   // dV[key][d] += sum_q P~[q][key] dO[q][d]
      mma_acc(ak, ds, qc);    // dK[key][d] += sum_q dS[q][key] Q[q][d]
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kk = key0 + 4 * g + r;
    if (kk < Sk) {
      *reinterpret_cast<float4*>(dK + b * dk_bs + (size_t)kk * dk_rs + h * 64 + 4 * j) =
          make_float4(ak[0][r] * scale, ak[1][r] * scale, ak[2][r] * scale, ak[3][r] * scale);
      *reinterpret_cast<float4*>(dV + b * dv_bs + (size_t)kk * dv_rs + h * 64 + 4 * j) =
          make_float4(av[0][r], av[1][r], av[2][r], av[3][r]);
    }
  }
}

}  // namespace

extern "C" int gridmm_attention_bwd(const float* Q, int64_t q_bs, int q_rs, const float* K, int64_t k_bs, int k_rs,
                                    const float* V, int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs,
                                    const float* O, int64_t o_bs, int o_rs, const float* dO, int64_t do_bs,
                                    int do_rs, const float* lse, float* delta, float* dQ, int64_t dq_bs, int dq_rs,
                                    float* dK, int64_t dk_bs, int dk_rs, float* dV, int64_t dv_bs, int dv_rs, int B,
                                    int heads, int Sq, int Sk, int Sqp, float scale, float dropout_p,
                                    unsigned long long seed, const unsigned long long* seed_dev,
                                    gridmm_stream_t stream) {
  if (B <= 0 || heads <= 0 || Sq <= 0 || Sk <= 0 || Sqp < Sq || Sqp % 16) return GRIDMM_EINVAL;
  if ((q_rs | k_rs | v_rs | o_rs | do_rs | dq_rs | dk_rs | dv_rs) & 3) return GRIDMM_EINVAL;
  if ((q_bs | k_bs | v_bs | o_bs | do_bs | dq_bs | dk_bs | dv_bs) & 3) return GRIDMM_EINVAL;
  hipStream_t st = as_stream(stream);
  const size_t n = (size_t)B * heads * Sqp;
  GRIDMM_LAUNCH(attention_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dO, do_bs, do_rs, O,
                o_bs, o_rs, delta, heads, Sq, Sqp, B);
  GRIDMM_CHECK_LAUNCH();
  // the K scale is folded into the S = Q K^T product once: dq kernel scales Q, dkv kernel scales K
  GRIDMM_LAUNCH(attention_bwd_dq_kernel, dim3((Sq + 63) / 64, heads, B), dim3(256), 0, st, Q, q_bs, q_rs, K, k_bs,
                k_rs, V, v_bs, v_rs, kmask, mask_bs, dO, do_bs, do_rs, lse, delta, dQ, dq_bs, dq_rs, Sq, Sk, Sqp,
                scale, dropout_p, seed, seed_dev);
  GRIDMM_CHECK_LAUNCH();
  GRIDMM_LAUNCH(attention_bwd_dkv_kernel, dim3((Sk + 63) / 64, heads, B), dim3(256), 0, st, Q, q_bs, q_rs, K, k_bs,
                k_rs, V, v_bs, v_rs, kmask, mask_bs, dO, do_bs, do_rs, lse, delta, dK, dk_bs, dk_rs, dV, dv_bs, dv_rs,
                Sq, Sk, Sqp, scale, dropout_p, seed, seed_dev);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
