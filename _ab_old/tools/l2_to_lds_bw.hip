// Micro-benchmark: per-CU bandwidth of L2-resident data into LDS / registers on gfx950.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)      mode 1: global_load_dwordx4 -> VGPR (sum)
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128
// build: hipcc --offload-arch=gfx950 -O3 l2_to_lds_bw.hip -o /tmp/l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// GEMM-like pattern: every wave-instruction reads 8 rows x 128 B with a row stride of `ld` uint4 (K=768 planes: 96)
template <int WAVES, int INFLIGHT>
__global__ __launch_bounds__(WAVES * 64) void bw_strided(const uint4* __restrict__ src, size_t n_vec, int iters,
                                                         float* __restrict__ sink, int ld, size_t win, int share) {
  __shared__ __attribute__((aligned(16))) uint4 lds[WAVES * 64 * INFLIGHT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint4* base = src + (((size_t)(blockIdx.x % share) * win) & (n_vec - 1));
  size_t row0 = (size_t)wave * 8 + (size_t)blockIdx.x * 131;
  int kcol = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) {
      const size_t idx = ((row0 + (lane >> 3)) * ld + kcol + (lane & 7)) & (win - 1);   // win: power of two
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + idx),
                                       (__attribute__((address_space(3))) void*)(lds + (k * WAVES + wave) * 64), 16, 0, 0);
      row0 += WAVES * 8;
    }
    kcol = (kcol + 8) % ld;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (lds[tid].x == 0x12345678u) sink[0] = 1.f;
}

template <int MODE, int WAVES, int INFLIGHT>
__global__ __launch_bounds__(WAVES * 64) void bw_kernel(const uint4* __restrict__ src, size_t n_vec, int iters,
                                                        float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) uint4 lds[WAVES * 64 * INFLIGHT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // each workgroup walks a private 1 MiB window of an L2/MALL-resident buffer (shared by 1/8 of the WGs)
  const size_t win = 65536;  // uint4 per window = 1 MiB
  const uint4* base = src + ((size_t)(blockIdx.x % 8) * win) % n_vec;
  uint4 acc = make_uint4(0, 0, 0, 0);
  size_t off = tid;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) {
      const uint4* p = base + (off % win);
      off += WAVES * 64;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(lds + (k * WAVES + wave) * 64), 16, 0, 0);
      } else {
        uint4 v = *p;
        if (MODE == 1) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        else lds[(k * WAVES + wave) * 64 + lane] = v;
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (MODE != 1) acc = lds[tid];
  if (acc.x == 0x12345678u) sink[0] = 1.f;
}

template <int MODE, int WAVES, int INFLIGHT>
void run(const uint4* d, size_t n_vec, float* sink, int wgs, const char* name) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bw_kernel<MODE, WAVES, INFLIGHT><<<wgs, WAVES * 64>>>(d, n_vec, 10, sink);
  hipEventRecord(e0);
  bw_kernel<MODE, WAVES, INFLIGHT><<<wgs, WAVES * 64>>>(d, n_vec, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * iters * INFLIGHT * WAVES * 64 * 16;
  printf("%-34s wgs=%4d waves=%d inflight=%2d KB/WG : %8.1f GB/s total, %6.1f GB/s per WG\n", name, wgs, WAVES,
         INFLIGHT * WAVES, bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
}

template <int WAVES, int INFLIGHT>
void run_strided(const uint4* d, size_t n_vec, float* sink, int wgs, int ld, size_t win, int share, const char* name) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bw_strided<WAVES, INFLIGHT><<<wgs, WAVES * 64>>>(d, n_vec, 10, sink, ld, win, share);
  hipEventRecord(e0);
  bw_strided<WAVES, INFLIGHT><<<wgs, WAVES * 64>>>(d, n_vec, iters, sink, ld, win, share);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * iters * INFLIGHT * WAVES * 64 * 16;
  printf("%-44s wgs=%4d : %8.1f GB/s total, %6.1f GB/s per WG\n", name, wgs, bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
}

int main() {
  {
    const size_t n_big = (size_t)256 << 16;  // 256 MiB
    uint4* big; float* sk;
    hipMalloc(&big, n_big * 16); hipMalloc(&sk, 4);
    hipMemset(big, 1, n_big * 16);
    for (int wgs : {256, 512}) {
      run_strided<4, 8>(big, n_big, sk, wgs, 96, (size_t)1 << 16, 8, "strided 8x128B, 1 MiB window/XCD-group");
      run_strided<4, 8>(big, n_big, sk, wgs, 96, (size_t)1 << 16, 1, "strided 8x128B, ONE 1 MiB window (all WGs)");
      run_strided<4, 8>(big, n_big, sk, wgs, 96, (size_t)32 << 16, 1, "strided 8x128B, ONE 32 MiB window (all WGs)");
      run_strided<4, 8>(big, n_big, sk, wgs, 96, (size_t)256 << 16, 1, "strided 8x128B, ONE 256 MiB window");
      run_strided<4, 8>(big, n_big, sk, wgs, 8, (size_t)32 << 16, 1, "contiguous 1 KiB, ONE 32 MiB window");
    }
    hipFree(big);
  }
  const size_t n_vec = 8 * 65536;  // 8 MiB: L2/MALL resident
  uint4* d; float* sink;
  hipMalloc(&d, n_vec * 16); hipMalloc(&sink, 4);
  hipMemset(d, 1, n_vec * 16);
  for (int wgs : {256, 512}) {
    run<0, 4, 8>(d, n_vec, sink, wgs, "LDS-DMA 4 waves x8");
    run<0, 4, 16>(d, n_vec, sink, wgs, "LDS-DMA 4 waves x16");
    run<0, 8, 8>(d, n_vec, sink, wgs, "LDS-DMA 8 waves x8");
    run<1, 4, 8>(d, n_vec, sink, wgs, "global_load->VGPR 4 waves x8");
    run<1, 8, 8>(d, n_vec, sink, wgs, "global_load->VGPR 8 waves x8");
    run<2, 4, 8>(d, n_vec, sink, wgs, "global_load->VGPR->ds_write 4w x8");
    run<2, 8, 8>(d, n_vec, sink, wgs, "global_load->VGPR->ds_write 8w x8");
  }
  return 0;
}
