// ABI bookkeeping for libgridmm_hip.so (the kernels live in the sibling .hip files) and the HBM streaming-read probe that
// bench.py reports as the MEASURED peak next to the 8 TB/s specification.
#include "common.h"

extern "C" int gridmm_abi_version(void) { return 29; }

namespace {
// Every workgroup streams its own contiguous slice once with 16-byte loads, eight in flight per lane, and leaves one float
// (so that nothing is optimised away).  2048 workgroups of 256 lanes: 8 per CU.
__global__ __launch_bounds__(256) void hbm_read_probe_kernel(const f32x4_t* __restrict__ p, size_t n4, float* __restrict__ out) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t beg = (size_t)blockIdx.x * per, end = beg + per < n4 ? beg + per : n4;
  float4 a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  size_t i = beg + threadIdx.x;
  for (; i + 7 * 256 < end; i += 8 * 256) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4_t v = __builtin_nontemporal_load(p + i + u * 256);
      a[u].x += v[0]; a[u].y += v[1]; a[u].z += v[2]; a[u].w += v[3];
    }
  }
  for (; i < end; i += 256) {
    const f32x4_t v = p[i];
    a[0].x += v[0]; a[0].y += v[1]; a[0].z += v[2]; a[0].w += v[3];
  }
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += (a[u].x + a[u].y) + (a[u].z + a[u].w);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, s);
}
}  // namespace

// Read `bytes` (multiple of 16) at `p` once; out: >= 2048 floats (zeroed by the caller; content is a by-product).  A
// measurement helper: time it over a window much larger than the 256 MiB Infinity Cache to get the HBM streaming-read rate.
extern "C" int gridmm_hbm_read_probe(const void* p, size_t bytes, float* out, gridmm_stream_t stream) {
  if (!p || !out || bytes < 16 || bytes % 16 || ((uintptr_t)p & 15)) return GRIDMM_EINVAL;
  GRIDMM_LAUNCH(hbm_read_probe_kernel, dim3(2048), dim3(256), 0, as_stream(stream), (const f32x4_t*)p, bytes / 16, out);
  GRIDMM_CHECK_LAUNCH();
  return GRIDMM_OK;
}
