// Optimizer step for training: gradient-norm clipping + AdamW, fused.
//
// Reference: pretrain_src/optim/adamw.py:56-112 (the HuggingFace "weight decay fix" AdamW: eps added to sqrt(v)
// before the bias-corrected step, decay applied AFTER the update with the raw lr) driven by train_r2r.py:288-303
// (clip_grad_norm_(5.0) then optimizer.step()); fine-tune: agent_base.py:201-205 (clip 40, torch AdamW semantics are
// selected with decay_first = 1).
//   pass 1  gridmm_grad_sumsq   : sum of squares of every gradient tensor into ONE device float (no host sync)
//   pass 2  gridmm_adamw_step   : per tensor, reads that float, scales g by min(1, max_norm / (norm + 1e-6)) exactly as
//                                 torch.nn.utils.clip_grad_norm_ does, and updates p / exp_avg / exp_avg_sq in place.
// Both are HBM-bound streaming kernels: 16 B/param read + 12 B/param written for fp32.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float ld(const T* p, size_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st(T* p, size_t i, float v) { p[i] = (T)v; }

template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* __restrict__ g, size_t n, float* __restrict__ acc) {
  __shared__ float s_w[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = ld(g, i);
    s += x * x;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(T* __restrict__ p, const T* __restrict__ g, T* __restrict__ m,
                                                    T* __restrict__ v, size_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float step_size, int decay_first,
                                                    const float* __restrict__ sumsq, float max_norm,
                                                    const float* __restrict__ dyn) {
  if (dyn) { lr = dyn[0]; step_size = dyn[1]; eps = dyn[2]; }   // captured steps: this step's scalars from device memory
  float scale = 1.f;
  if (sumsq) {
    const float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
    scale = c < 1.f ? c : 1.f;
  }
  // rnd(): round to the tensor's dtype.  The reference updates fp16 tensors with one in-place torch op after the other
  // (clip_grad_norm_'s mul_, then adamw.py:88-109), each computing in fp32 and rounding its result to fp16 -- which is
  // what makes g*g*(1-b2) below ~6e-8 vanish from the second moment (and the update explode to m/eps) there.  The same
  // roundings are taken here; for fp32 tensors rnd() is the identity.
  auto rnd = [](float x) { return (float)(T)x; };
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = rnd(ld(g, i) * scale);
    float pi = ld(p, i);
    const float mi = rnd(rnd(b1 * ld(m, i)) + (1.f - b1) * gi);
    const float vi = rnd(rnd(b2 * ld(v, i)) + (1.f - b2) * gi * gi);
    if (decay_first && wd > 0.f) pi = rnd(pi - lr * wd * pi);          // torch.optim.AdamW order
    pi = rnd(pi - step_size * (mi / rnd(rnd(sqrtf(vi)) + eps)));
    if (!decay_first && wd > 0.f) pi = rnd(pi - lr * wd * pi);         // adamw.py:108-109
    st(p, i, pi);
    st(m, i, mi);
    st(v, i, vi);
  }
}

// ---- multi-tensor forms: ONE launch over all fp32 parameter tensors (the model has ~370 of them; per-tensor launches
// are launch-bound).  desc[t] = {p, g, m, v, n, lr, step_size, eps, wd}; chunk_first[t] = first 16K-element chunk of
// tensor t in the grid (prefix sums, chunk_first[T] = total); a workgroup finds its tensor by binary search.
struct TensorDesc {
  void* p; const void* g; void* m; void* v;   // dtype 0: float, 1: _Float16 (parameter, gradient and both moments alike)
  long long n;
  float lr, step_size, eps, wd;
  int dtype, pad;
};
static_assert(sizeof(TensorDesc) == 64, "record layout shared with gridmm_amd/optim.py (_REC)");
constexpr int MT_CHUNK = 16384;

__device__ __forceinline__ int find_tensor(const int* __restrict__ chunk_first, int T, int chunk) {
  int lo = 0, hi = T;          // largest t with chunk_first[t] <= chunk
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_first[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void multi_sumsq_kernel(const TensorDesc* __restrict__ desc,
                                                          const int* __restrict__ chunk_first, int T,
                                                          float* __restrict__ acc) {
  __shared__ float s_w[4];
  const int t = find_tensor(chunk_first, T, blockIdx.x);
  const TensorDesc d = desc[t];
  const long long i0 = (long long)(blockIdx.x - chunk_first[t]) * MT_CHUNK;
  const long long i1 = i0 + MT_CHUNK < d.n ? i0 + MT_CHUNK : d.n;
  float s = 0.f;
  if (d.dtype == 0) {
    const float* g = static_cast<const float*>(d.g);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) { const float x = g[i]; s += x * x; }
  } else {
    const _Float16* g = static_cast<const _Float16*>(d.g);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) { const float x = (float)g[i]; s += x * x; }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) acc[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);   // one partial per chunk: no atomics
}

// the per-chunk partials in a fixed order (thread t takes chunks t, t + 256, ...; then the wave / LDS tree): the
// gradient norm -- and with it the clip factor of every parameter -- is bit-reproducible from run to run
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float s_w[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// one chunk of one tensor; same arithmetic (and, for fp16, the same per-op roundings) as adamw_kernel<T>
template <typename T>
__device__ __forceinline__ void adamw_range(T* __restrict__ p, const T* __restrict__ g, T* __restrict__ m,
                                            T* __restrict__ v, long long i0, long long i1, float scale,
                                            const TensorDesc& d, float b1, float b2, int decay_first) {
  auto rnd = [](float x) { return (float)(T)x; };
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float gi = rnd(ld(g, i) * scale);
    float pi = ld(p, i);
    const float mi = rnd(rnd(b1 * ld(m, i)) + (1.f - b1) * gi);
    const float vi = rnd(rnd(b2 * ld(v, i)) + (1.f - b2) * gi * gi);
    if (decay_first && d.wd > 0.f) pi = rnd(pi - d.lr * d.wd * pi);
    pi = rnd(pi - d.step_size * (mi / rnd(rnd(sqrtf(vi)) + d.eps)));
    if (!decay_first && d.wd > 0.f) pi = rnd(pi - d.lr * d.wd * pi);
    st(p, i, pi);
    st(m, i, mi);
    st(v, i, vi);
  }
}

// ---- optimizer-owned weight planes (round 5).  A trained 2-D weight is needed as bf16 hi / lo planes in BOTH orientations every
// step (forward: W [N][K]; dX: W^T [K][N]); round 4 re-packed every updated weight with a transposing split launch per
// weight and step (~124 launches, ~1.4 ms of the 161 M-parameter step).  For a tensor with a plane record the update walks
// 64 x 64 tiles instead of flat 16 K-element chunks (same element arithmetic, bit for bit), writes the row planes straight
// from the registers and the transposed planes through an LDS tile: the weight's planes are current the moment the optimizer
// step ends, and no pack launch follows.  N % 64 == 0, K % 64 == 0, fp32; ldw / ldt: row pitches of the two plane pairs (a
// member of a fused q | k | v projection writes its row block / column block of the shared planes).
struct PlaneDesc {
  unsigned short *hi, *lo, *thi, *tlo;   // hi == NULL: no planes for this tensor
  int N, K, ldw, ldt;
};
static_assert(sizeof(PlaneDesc) == 48, "record layout shared with gridmm_amd/optim.py (_PREC)");

__device__ __forceinline__ void adamw_tiles(const TensorDesc& d, const PlaneDesc& pl, int chunk, float scale, float b1, float b2,
                                            int decay_first) {
  __shared__ float tile[64][65];          // the updated weights of one 64 x 64 tile (the transposing splitter's layout)
  float* p = static_cast<float*>(d.p);
  const float* g = static_cast<const float*>(d.g);
  float* m = static_cast<float*>(d.m);
  float* v = static_cast<float*>(d.v);
  const int tiles_k = pl.K >> 6, ntiles = (pl.N >> 6) * tiles_k;
  const int tid = threadIdx.x;
  const int lc = tid & 63, lr = tid >> 6;                 // update role: column lc, rows lr, lr + 4, ... (256-byte row segments)
  const int rr = tid >> 2, rc = (tid & 3) * 16;           // store roles: row / transposed row rr, 16 elements from rc
  for (int t = 4 * chunk; t < 4 * chunk + 4 && t < ntiles; ++t) {
    const int tr = t / tiles_k, tc = t - tr * tiles_k;
    const size_t base = (size_t)(tr * 64) * pl.K + tc * 64 + lc;
    float ge[16], pe[16], me[16], ve[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const size_t o = base + (size_t)(lr + 4 * i) * pl.K;
      ge[i] = g[o]; pe[i] = p[o]; me[i] = m[o]; ve[i] = v[o];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {            // (adamw_range<float>: rnd() is the identity)
      const size_t o = base + (size_t)(lr + 4 * i) * pl.K;
      const float gi = ge[i] * scale;
      float pi = pe[i];
      const float mi = b1 * me[i] + (1.f - b1) * gi;
      const float vi = b2 * ve[i] + (1.f - b2) * gi * gi;
      if (decay_first && d.wd > 0.f) pi = pi - d.lr * d.wd * pi;
      pi = pi - d.step_size * (mi / (sqrtf(vi) + d.eps));
      if (!decay_first && d.wd > 0.f) pi = pi - d.lr * d.wd * pi;
      p[o] = pi; m[o] = mi; v[o] = vi;
      tile[lr + 4 * i][lc] = pi;
    }
    __syncthreads();
    {   // row planes: row tr * 64 + rr, columns tc * 64 + rc .. + 15
      unsigned int hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split2_bf16(tile[rr][rc + 2 * e], tile[rr][rc + 2 * e + 1], hi[e], lo[e]);
      const size_t o = (size_t)(tr * 64 + rr) * pl.ldw + tc * 64 + rc;
      uint4* ph = reinterpret_cast<uint4*>(pl.hi + o);
      uint4* pq = reinterpret_cast<uint4*>(pl.lo + o);
      ph[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]); ph[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
      pq[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]); pq[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
    }
    {   // transposed planes: row tc * 64 + rr of W^T, columns tr * 64 + rc .. + 15
      unsigned int hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split2_bf16(tile[rc + 2 * e][rr], tile[rc + 2 * e + 1][rr], hi[e], lo[e]);
      const size_t o = (size_t)(tc * 64 + rr) * pl.ldt + tr * 64 + rc;
      uint4* ph = reinterpret_cast<uint4*>(pl.thi + o);
      uint4* pq = reinterpret_cast<uint4*>(pl.tlo + o);
      ph[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]); ph[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
      pq[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]); pq[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void multi_adamw_kernel(const TensorDesc* __restrict__ desc,
                                                          const int* __restrict__ chunk_first, int T, float b1, float b2,
                                                          int decay_first, const float* __restrict__ sumsq,
                                                          float max_norm, const PlaneDesc* __restrict__ planes) {
  const int t = find_tensor(chunk_first, T, blockIdx.x);
  const TensorDesc d = desc[t];
  float scale = 1.f;
  if (sumsq) {
    const float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
    scale = c < 1.f ? c : 1.f;
  }
  if (planes && planes[t].hi && d.dtype == 0) {       // (workgroup-uniform)
    adamw_tiles(d, planes[t], blockIdx.x - chunk_first[t], scale, b1, b2, decay_first);
    return;
  }
  const long long i0 = (long long)(blockIdx.x - chunk_first[t]) * MT_CHUNK;
  const long long i1 = i0 + MT_CHUNK < d.n ? i0 + MT_CHUNK : d.n;
  if (d.dtype == 0)
    adamw_range(static_cast<float*>(d.p), static_cast<const float*>(d.g), static_cast<float*>(d.m),
                static_cast<float*>(d.v), i0, i1, scale, d, b1, b2, decay_first);
  else   // the reference's fp16 grid_proj: fp16 gradient and moments, every intermediate rounded to fp16 (adamw_kernel)
    adamw_range(static_cast<_Float16*>(d.p), static_cast<const _Float16*>(d.g), static_cast<_Float16*>(d.m),
                static_cast<_Float16*>(d.v), i0, i1, scale, d, b1, b2, decay_first);
}

inline unsigned grid_for(size_t n, size_t cap = 4096) {
  size_t b = (n + 255) / 256;
  return (unsigned)(b > cap ? cap : (b ? b : 1));
}

}  // namespace

// ---- gradient accumulation of a multi-step backward in ONE launch.  A fine-tuning iteration backpropagates through the
// 7 .. 15 navigation steps of a rollout (map_nav_src/r2r/agent.py:268-451, agent_base.py:190-199): every step hands every
// parameter a gradient, and autograd's AccumulateGrad adds them with one launch per parameter and step (~2 000 launches of
// ~4 us per iteration).  The custom Functions of gridmm_amd.autograd keep those gradients aside instead
// (deferred_param_grads) and this kernel sums them at the end of the backward: dst = ((dst + src0) + src1) + ... in
// list order -- the order and the fp32 roundings of the sequential in-place adds it replaces (bit-identical).
struct SumDesc {
  float* dst;
  const float* src[7];
  long long n;
  int n_src, pad;
};
static_assert(sizeof(SumDesc) == 80, "record layout shared with gridmm_amd/autograd.py (_SUM_REC)");

__global__ __launch_bounds__(256) void multi_grad_accumulate_kernel(const SumDesc* __restrict__ desc,
                                                                    const int* __restrict__ chunk_first, int T) {
  const int t = find_tensor(chunk_first, T, blockIdx.x);
  const SumDesc d = desc[t];
  const long long i0 = (long long)(blockIdx.x - chunk_first[t]) * MT_CHUNK;
  const long long i1 = i0 + MT_CHUNK < d.n ? i0 + MT_CHUNK : d.n;
  unsigned long long al = (unsigned long long)(size_t)d.dst;
  for (int k = 0; k < d.n_src; ++k) al |= (unsigned long long)(size_t)d.src[k];
  long long i = i0;
  if ((al & 15) == 0) {                                 // MT_CHUNK % 4 == 0: chunk starts keep the alignment
    const long long v1 = i0 + ((i1 - i0) & ~3LL);
    for (i = i0 + 4 * threadIdx.x; i < v1; i += 4 * 256) {
      float4 a = *reinterpret_cast<const float4*>(d.dst + i);
      for (int k = 0; k < d.n_src; ++k) {
        const float4 b = *reinterpret_cast<const float4*>(d.src[k] + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      *reinterpret_cast<float4*>(d.dst + i) = a;
    }
    i = v1;
  }
  for (i += threadIdx.x; i < i1; i += 256) {
    float a = d.dst[i];
    for (int k = 0; k < d.n_src; ++k) a += d.src[k][i];
    d.dst[i] = a;
  }
}

extern "C" int gridmm_multi_grad_accumulate(const void* desc, const int* chunk_first, int n_tensors, int n_chunks,
                                            gridmm_stream_t stream) {
  if (n_tensors <= 0 || n_chunks <= 0 || !desc || !chunk_first) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(multi_grad_accumulate_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), (const SumDesc*)desc,
                chunk_first, n_tensors);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_grad_sumsq(const void* g, int64_t n, int dtype, float* acc, gridmm_stream_t stream) {
  if (n <= 0 || !acc || (dtype != 0 && dtype != 1)) return GRIDMM_EINVAL;
  hipStream_t st_ = as_stream(stream);
  if (dtype == 0) GRIDMM_LAUNCH((sumsq_kernel<float>), dim3(grid_for(n / 8, 512)), dim3(256), 0, st_, (const float*)g, (size_t)n, acc);   // <= 512 atomics on *acc
  else GRIDMM_LAUNCH((sumsq_kernel<_Float16>), dim3(grid_for(n / 8, 512)), dim3(256), 0, st_, (const _Float16*)g, (size_t)n, acc);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_adamw_step(void* p, const void* g, void* m, void* v, int64_t n, int dtype, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, float step_size, int decay_first,
                                 const float* sumsq, float max_norm, const float* dyn, gridmm_stream_t stream) {
  if (n <= 0 || (dtype != 0 && dtype != 1)) return GRIDMM_EINVAL;
  hipStream_t st_ = as_stream(stream);
  if (dtype == 0)
    GRIDMM_LAUNCH((adamw_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st_, (float*)p, (const float*)g, (float*)m,
                  (float*)v, (size_t)n, lr, beta1, beta2, eps, weight_decay, step_size, decay_first, sumsq, max_norm, dyn);
  else
    GRIDMM_LAUNCH((adamw_kernel<_Float16>), dim3(grid_for(n)), dim3(256), 0, st_, (_Float16*)p, (const _Float16*)g,
                  (_Float16*)m, (_Float16*)v, (size_t)n, lr, beta1, beta2, eps, weight_decay, step_size, decay_first,
                  sumsq, max_norm, dyn);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_multi_grad_sumsq(const void* desc, const int* chunk_first, int n_tensors, int n_chunks,
                                       float* partial, float* out, gridmm_stream_t stream) {
  if (n_tensors <= 0 || n_chunks <= 0 || !partial || !out) return GRIDMM_EINVAL;
  hipStream_t st_ = as_stream(stream);
  GRIDMM_LAUNCH(multi_sumsq_kernel, dim3(n_chunks), dim3(256), 0, st_, (const TensorDesc*)desc, chunk_first, n_tensors,
                partial);
  GRIDMM_CHECK_LAUNCH();
  GRIDMM_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, st_, partial, n_chunks, out);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}

extern "C" int gridmm_multi_adamw_step(const void* desc, const int* chunk_first, int n_tensors, int n_chunks, float beta1,
                                       float beta2, int decay_first, const float* sumsq, float max_norm, const void* planes,
                                       gridmm_stream_t stream) {
  if (n_tensors <= 0 || n_chunks <= 0) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(multi_adamw_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), (const TensorDesc*)desc,
                chunk_first, n_tensors, beta1, beta2, decay_first, sumsq, max_norm, (const PlaneDesc*)planes);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
