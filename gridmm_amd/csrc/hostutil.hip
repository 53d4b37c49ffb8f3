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

// ---- navigation-input collation of a lock-step batch (host): what r2r/agent.py:96-205 (_nav_gmap_variable +
// _nav_vp_variable) and the fused-logit loops of models/vilmodel.py:881-899 compute per episode in Python / NumPy, from the
// (B, cap, ...) arrays of the batch's topological maps.  Two calls: *_plan orders the graph nodes of every episode and
// returns the longest sequence (the caller picks the padded node axis G from it), *_fill writes every array of the step.
#include <cmath>

namespace {
constexpr double kMaxDist = 30.0, kMaxStep = 10.0, kUnreachable = 95959595.0;

struct PosFeat { float q[4]; float line, graph, hops; };

// graph_utils.batched_pos_features for one (origin, target) pair: float64 geometry, float32 trigonometry of the float32
// angles, distances scaled in float64 and rounded once.
inline PosFeat pos_feature(const double* origin, const double* target, double base_h, double base_e, double graph,
                           double hops) {
  const double dx = target[0] - origin[0], dy = target[1] - origin[1], dz = target[2] - origin[2];
  const double flat = std::fmax(std::sqrt(dx * dx + dy * dy), 1e-8);
  const double full = std::fmax(std::sqrt(dx * dx + dy * dy + dz * dz), 1e-8);
  double heading = std::asin(dx / flat);
  if (dy < 0) heading = M_PI - heading;
  heading -= base_h;
  const double elevation = std::asin(dz / full) - base_e;
  const float h = (float)heading, e = (float)elevation;
  PosFeat f;
  f.q[0] = sinf(h); f.q[1] = cosf(h); f.q[2] = sinf(e); f.q[3] = cosf(e);
  f.line = (float)(full / kMaxDist);
  f.graph = (float)(graph / kMaxDist);
  f.hops = (float)(hops / kMaxStep);
  return f;
}

inline void write_feat(float* dst, const PosFeat& f, int afs) {
  const int reps = afs / 4 > 1 ? afs / 4 : 1;
  for (int r = 0; r < reps; ++r)
    for (int i = 0; i < 4; ++i) dst[4 * r + i] = f.q[i];
  dst[afs] = f.line; dst[afs + 1] = f.graph; dst[afs + 2] = f.hops;
}

inline long route_len(const int32_t* v, int cap, int x0, int y0, std::vector<std::pair<int, int>>& stack) {
  if (x0 == y0) return 0;
  long n = 0, guard = 0;
  stack.clear();
  stack.emplace_back(x0, y0);
  while (!stack.empty()) {
    auto [x, y] = stack.back();
    stack.pop_back();
    if (x == y) continue;
    const int k = v[(size_t)x * cap + y];
    if (k < 0) { ++n; continue; }
    if (k >= cap || ++guard > 4L * cap * cap) return -1;
    stack.emplace_back(k, y);
    stack.emplace_back(x, k);
  }
  return n;
}
}  // namespace

// order [B][cap] int64 out: node ids of the episode's map sequence (visited in id order, then unvisited in id order; the
// unvisited only when !enc_full_graph); m / n_vis / n_unv [B] int64 out.  seen_eff [B][cap] uint8 out = the "visited" flags
// the step uses (the current node only with act_visited_nodes).  Returns max_b m (>= 0) or GRIDMM_EINVAL.
extern "C" int gridmm_collate_nav_plan(const uint8_t* seen, const int64_t* n, const int64_t* cur, int B, int cap,
                                       int enc_full_graph, int act_visited_nodes, int64_t* order, int64_t* m, int64_t* n_vis,
                                       int64_t* n_unv, uint8_t* seen_eff) {
  if (!seen || !n || !cur || !order || !m || !n_vis || !n_unv || !seen_eff || B < 0 || cap <= 0) return GRIDMM_EINVAL;
  long M = 0;
  for (int b = 0; b < B; ++b) {
    const int nb = (int)n[b];
    if (nb < 0 || nb > cap || cur[b] < 0 || cur[b] >= cap) return GRIDMM_EINVAL;
    uint8_t* se = seen_eff + (size_t)b * cap;
    long vis = 0;
    for (int k = 0; k < cap; ++k) {
      se[k] = (k < nb) && (act_visited_nodes ? (k == cur[b]) : (seen[(size_t)b * cap + k] != 0));
      vis += se[k];
    }
    int64_t* o = order + (size_t)b * cap;
    long j = 0;
    if (enc_full_graph)
      for (int k = 0; k < nb; ++k) if (se[k]) o[j++] = k;
    for (int k = 0; k < nb; ++k) if (!se[k]) o[j++] = k;
    m[b] = j;
    n_vis[b] = enc_full_graph ? vis : 0;
    n_unv[b] = nb - vis;
    for (; j < cap; ++j) o[j] = 0;
    if (m[b] > M) M = m[b];
  }
  return (int)M;
}

// Every array of the step's nav_inputs that the host builds (all outputs zero-filled / defaulted here):
//   gpos [B][G][F] f32, vpos [B][V1][2F] f32, pair [B][G][G] f32, steps [B][G] i64, visited [B][G] u8, slot [B][G] i64,
//   inv [B][G] f32, gmask [B][G] u8, cand_of_node [B][G] i32, cand_visited [B][V1] u8.     F = afs + 3.
// Inputs: the maps (pos f64 [B][cap][3], dist f64 [B][cap][cap], via i32, step i64 [B][cap]), the plan, cid [B][Cw] i64 +
// nc [B] i64 (candidate node ids of the panorama), start [B], heading / elevation [B] f64, cnt [B][slots] i32 (views
// accumulated per node embedding).  Returns GRIDMM_OK, GRIDMM_EINVAL, or 1 + (b * G + row) of the first graph node that has
// no embedding yet (cnt == 0): the caller raises the reference's KeyError.
extern "C" int gridmm_collate_nav_fill(const double* pos, const double* dist, const int32_t* via, const int64_t* step,
                                       const int64_t* order, const int64_t* m, const int64_t* n_vis, const uint8_t* seen_eff,
                                       const int64_t* cur, const int64_t* start, const int64_t* cid, const int64_t* nc,
                                       const double* heading, const double* elevation, const int32_t* cnt, int B, int cap,
                                       int Cw, int slots, int G, int V1, int afs, int enc_full_graph, float* gpos, float* vpos,
                                       float* pair, int64_t* steps, uint8_t* visited, int64_t* slot, float* inv, uint8_t* gmask,
                                       int32_t* cand_of_node, uint8_t* cand_visited) {
  if (!pos || !dist || !via || !step || !order || !m || !n_vis || !seen_eff || !cur || !start || !nc || !heading ||
      !elevation || !cnt || !gpos || !vpos || !pair || !steps || !visited || !slot || !inv || !gmask || !cand_of_node ||
      !cand_visited || B < 0 || cap <= 0 || G <= 0 || V1 <= 0 || afs < 4 || (Cw > 0 && !cid))
    return GRIDMM_EINVAL;
  const int F = afs + 3;
  std::vector<std::pair<int, int>> stack;
  int missing = 0;
  for (int b = 0; b < B; ++b) {
    const double* P = pos + (size_t)b * cap * 3;
    const double* D = dist + (size_t)b * cap * cap;
    const int32_t* Vv = via + (size_t)b * cap * cap;
    const int64_t* o = order + (size_t)b * cap;
    const uint8_t* se = seen_eff + (size_t)b * cap;
    const int mb = (int)m[b], c = (int)cur[b], ncb = (int)nc[b];
    if (mb + 1 > G || ncb > Cw || ncb + 1 > V1 || start[b] < 0 || start[b] >= cap || c < 0 || c >= cap) return GRIDMM_EINVAL;
    float* gp = gpos + (size_t)b * G * F;
    float* vp = vpos + (size_t)b * V1 * 2 * F;
    float* pr = pair + (size_t)b * G * G;
    for (size_t i = 0; i < (size_t)G * F; ++i) gp[i] = 0.f;
    for (size_t i = 0; i < (size_t)V1 * 2 * F; ++i) vp[i] = 0.f;
    for (size_t i = 0; i < (size_t)G * G; ++i) pr[i] = 0.f;
    for (int r = 0; r < afs / 4; ++r) { gp[4 * r + 1] = 1.f; gp[4 * r + 3] = 1.f; }   // the stop token: (sin 0, cos 0, sin 0, cos 0) in every repeat of the quad (get_angle_fts)
    auto feat = [&](int t) {
      double graph = 0.0, hops = 0.0;
      if (t != c) {
        graph = D[(size_t)c * cap + t];
        if (!std::isfinite(graph)) graph = kUnreachable;
        const long r = Vv[(size_t)c * cap + t] < 0 ? 1 : route_len(Vv, cap, c, t, stack);
        hops = r < 0 ? 1.0 : (double)r;
      }
      return pos_feature(P + 3 * c, P + 3 * t, heading[b], elevation[b], graph, hops);
    };
    for (int g = 0; g < G; ++g) {
      const size_t i = (size_t)b * G + g;
      steps[i] = 0; visited[i] = 0; slot[i] = 0; inv[i] = 1.f; gmask[i] = g < mb + 1; cand_of_node[i] = -2;
    }
    for (int v = 0; v < V1; ++v) cand_visited[(size_t)b * V1 + v] = 0;
    // graph nodes
    for (int j = 0; j < mb; ++j) {
      const int id = (int)o[j];
      const size_t i = (size_t)b * G + j + 1;
      write_feat(gp + (size_t)(j + 1) * F, feat(id), afs);
      steps[i] = step[(size_t)b * cap + id];
      visited[i] = j < n_vis[b];
      slot[i] = id + 1;
      if (id + 1 >= slots) return GRIDMM_EINVAL;
      const int cc = cnt[(size_t)b * slots + id + 1];
      if (cc == 0 && !missing) missing = 1 + (int)i;
      inv[i] = 1.0f / (float)(cc > 1 ? cc : 1);
      for (int k = 0; k < mb; ++k) {
        if (k == j) continue;
        double d = D[(size_t)id * cap + o[k]];
        if (!std::isfinite(d)) d = kUnreachable;
        pr[(size_t)(j + 1) * G + k + 1] = (float)d;
      }
    }
    // the start node's features in every row of the view-point block, the candidates' features behind them
    {
      std::vector<float> sf((size_t)F);
      write_feat(sf.data(), feat((int)start[b]), afs);
      for (int v = 0; v < V1; ++v)
        for (int i = 0; i < F; ++i) vp[(size_t)v * 2 * F + i] = sf[i];
    }
    for (int k = 0; k < ncb; ++k) {
      const int id = (int)cid[(size_t)b * Cw + k];
      if (id < 0 || id >= cap) return GRIDMM_EINVAL;
      write_feat(vp + (size_t)(k + 1) * 2 * F + F, feat(id), afs);
      if (enc_full_graph && se[id]) cand_visited[(size_t)b * V1 + k + 1] = 1;
    }
    // fusion maps (vilmodel.py:884-899): the candidate column of every unvisited node (the last match wins)
    for (int j = (int)n_vis[b]; j < mb; ++j) {
      int col = -1;
      for (int k = 0; k < ncb; ++k)
        if (cid[(size_t)b * Cw + k] == o[j] && !cand_visited[(size_t)b * V1 + k + 1]) col = k + 1;
      cand_of_node[(size_t)b * G + j + 1] = col;
    }
  }
  return missing ? missing : GRIDMM_OK;
}
