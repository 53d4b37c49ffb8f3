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

__global__ __launch_bounds__(256) void multi_adamw_kernel(const TensorDesc* __restrict__ desc,
                                                          const int* __restrict__ chunk_first, int T, float b1, float b2,
                                                          int decay_first, const float* __restrict__ sumsq,
                                                          float max_norm) {
  const int t = find_tensor(chunk_first, T, blockIdx.x);
  const TensorDesc d = desc[t];
  float scale = 1.f;
  if (sumsq) {
    const float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
    scale = c < 1.f ? c : 1.f;
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
                                       float beta2, int decay_first, const float* sumsq, float max_norm,
                                       gridmm_stream_t stream) {
  if (n_tensors <= 0 || n_chunks <= 0) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(multi_adamw_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), (const TensorDesc*)desc,
                chunk_first, n_tensors, beta1, beta2, decay_first, sumsq, max_norm);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
