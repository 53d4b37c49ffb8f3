// Host-side helpers of the agent loop's collation (no device work, no stream): the pieces of the reference's per-episode
// Python that stay on the CPU but do not vectorise in NumPy.
#include "common.h"

#include <vector>

// Number of legs of the route from `cur[b]` to every target node: the length of FloydGraph.path(x, y)
// (map_nav_src/models/graph_utils.py:75-100) -- path(x, y) = [y] if there is no pivot, else path(x, k) + path(k, y) with
// k = the pivot recorded for (x, y) -- evaluated on the CURRENT pivot matrix, as the reference's recursion does.
//   via   [B][cap][cap] int32 [host]  pivot ids (-1: direct edge or unknown)
//   cur   [B] int64 [host], tgt [B][T] int64 [host], mask [B][T] uint8 [host] (0: entry skipped, hops = 1)
//   hops  [B][T] float64 [host] out: 0 for tgt == cur, else the number of legs
extern "C" int gridmm_route_lengths(const int32_t* via, int B, int cap, const int64_t* cur, const int64_t* tgt,
                                    const uint8_t* mask, int T, double* hops) {
  if (!via || !cur || !tgt || !hops || B < 0 || cap <= 0 || T < 0) return GRIDMM_EINVAL;
  std::vector<std::pair<int, int>> stack;
  for (int b = 0; b < B; ++b) {
    const int32_t* v = via + (size_t)b * cap * cap;
    const int64_t c = cur[b];
    if (c < 0 || c >= cap) return GRIDMM_EINVAL;
    for (int t = 0; t < T; ++t) {
      const int64_t y0 = tgt[(size_t)b * T + t];
      double& out = hops[(size_t)b * T + t];
      if (y0 == c) { out = 0.0; continue; }
      out = 1.0;
      if ((mask && !mask[(size_t)b * T + t]) || y0 < 0 || y0 >= cap) continue;
      if (v[c * cap + y0] < 0) continue;
      long n = 0, guard = 0;
      stack.clear();
      stack.emplace_back((int)c, (int)y0);
      while (!stack.empty()) {
        auto [x, y] = stack.back();
        stack.pop_back();
        if (x == y) continue;
        const int k = v[(size_t)x * cap + y];
        if (k < 0) { ++n; continue; }
        if (k >= cap || ++guard > 4L * cap * cap) return GRIDMM_EINVAL;   // (a corrupt pivot matrix must not loop forever)
        stack.emplace_back(k, y);
        stack.emplace_back(x, k);
      }
      out = (double)n;
    }
  }
  return GRIDMM_OK;
}
