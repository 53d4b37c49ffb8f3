// Does data written by one kernel stay in the writing XCD's L2 for the NEXT kernel of the stream?  Producer: workgroup w
// writes chunk w (CH bytes).  Consumer A: workgroup w reads chunk w (same XCD: w % 8).  Consumer B: workgroup w reads chunk
// w + 1 (written by the neighbouring XCD).  Buffer sizes from L2-resident (8 MB) to beyond (256 MB).
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_affinity.hip -o /tmp/xcd && /tmp/xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void produce(float4* __restrict__ buf, size_t per_wg) {
  float4* p = buf + (size_t)blockIdx.x * per_wg;
  for (size_t i = threadIdx.x; i < per_wg; i += 256) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void consume(const float4* __restrict__ buf, size_t per_wg, int shift, int nwg,
                                               float* __restrict__ out) {
  const float4* p = buf + (size_t)((blockIdx.x + shift) % nwg) * per_wg;
  float s = 0.f;
  for (size_t i = threadIdx.x; i < per_wg; i += 256) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  if (s == -1.f) out[0] = s;
}
int main() {
  float* out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t mb : {8, 16, 24, 48, 96, 256}) {
    const int nwg = 2048;
    const size_t bytes = mb << 20, per_wg = bytes / nwg / 16;
    float4* buf; hipMalloc(&buf, bytes);
    for (int shift : {0, 1, 4, 8}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        produce<<<nwg, 256>>>(buf, per_wg);
        hipEventRecord(e0);
        consume<<<nwg, 256>>>(buf, per_wg, shift, nwg, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
      printf("%4zu MB  consumer reads chunk (w + %d)  %7.1f us  %6.0f GB/s%s\n", mb, shift, best * 1e3, bytes / best / 1e6,
             shift % 8 == 0 ? "   (same XCD as the producer)" : "");
    }
    hipFree(buf);
  }
  return 0;
}
