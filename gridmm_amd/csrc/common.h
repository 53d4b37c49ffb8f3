// Shared device helpers for the gfx950 kernels of libgridmm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gridmm.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));

#define GRIDMM_CHECK_LAUNCH()                                   \
  do {                                                          \
    hipError_t e_ = hipGetLastError();                          \
    if (e_ != hipSuccess) return -1000 - (int)e_;               \
  } while (0)

// torch (and anything else in the process) can leave a benign sticky error (e.g. hipErrorNotReady from an
// event query) in the per-thread last-error slot: clear it before the launch we are about to check.
#define GRIDMM_LAUNCH(...)            \
  do {                                \
    (void)hipGetLastError();          \
    hipLaunchKernelGGL(__VA_ARGS__);  \
  } while (0)

constexpr int MAX_H_BWD = 1024;   // widest row the LayerNorm backward stages in LDS

static inline hipStream_t as_stream(gridmm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Zero-fill as a KERNEL (n floats) instead of hipMemsetAsync on the training path.  Observed on ROCm 7.2 inside the
// captured pre-training step (~2000 nodes, default pre-recorded graph packets): the column sums / partial sums that
// follow these clears with atomics came out wrong from the second replay on, i.e. the memset nodes had not cleared
// their destination in time; small graphs do not show it (tools/dbg_memset_node.py).  A kernel node replays like
// every other launch.
static __global__ void gridmm_zero_f32_kernel(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
static inline bool gridmm_zero_f32(float* p, size_t n, hipStream_t st) {
  if (n == 0) return true;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  (void)hipGetLastError();
  hipLaunchKernelGGL(gridmm_zero_f32_kernel, dim3(blocks), dim3(256), 0, st, p, n);
  return hipGetLastError() == hipSuccess;
}

// fp32 -> bf16 bits, round-to-nearest-even (inputs are finite on this path).
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float x) {
  unsigned int u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}

// two fp32 -> packed bf16x2 (lo half = a, hi half = b), round-to-nearest-even, ONE VALU instruction on gfx950
__device__ __forceinline__ unsigned int cvt_pk_bf16(float a, float b) {
  unsigned int r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// hi/lo bf16 split of two floats: hi = rne(x), lo = rne(x - hi); returns packed pairs
__device__ __forceinline__ void split2_bf16(float a, float b, unsigned int& hi, unsigned int& lo) {
  hi = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  lo = cvt_pk_bf16(ra, rb);
}

// 64-lane wave reductions (wave = 64 on CDNA).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Counter-based dropout for the attention probabilities (training): element (b, h, q, k) is kept iff
// hash(seed, linear index) >= p.  Stateless, so the forward and both backward kernels regenerate the same mask.
__host__ __device__ __forceinline__ unsigned int gridmm_hash32(unsigned int x) {   // murmur3 finaliser
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned int idx, float p) {
  const unsigned int x = gridmm_hash32((idx * 0x9E3779B1u) ^ (unsigned int)seed) ^ (unsigned int)(seed >> 32);
  return (float)(gridmm_hash32(x) >> 8) * (1.0f / 16777216.0f) >= p;
}
