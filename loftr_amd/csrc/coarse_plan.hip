// Host-side planner of the persistent coarse transformer (encoder_fused.hip: coarse_persistent_kernel).
//   reference: src/loftr/loftr_module/transformer.py:80-101 (the layer schedule), :35-58 (what a call computes)
// Builds the dependency graph of the K / F / X items of one LocalFeatureTransformer.forward and orders the queue by a list schedule
// on 256 workers (critical path first, estimated durations).  Workgroups pop the queue IN ORDER and wait for an item's dependencies,
// so any topological order is correct; a good one keeps the waits short.  order = 1 is the reference's call order (call by call:
// all K, all F, all X items) -- the order a launch-per-call implementation executes, kept for the bit-identity test.
#include <algorithm>
#include <queue>
#include <vector>
#include "coarse_plan.h"

namespace {
struct Node { uint32_t what, signal, dep[4], sig2[2]; float dur; std::vector<int> pred; };

void build_graph(const PctShape& s, std::vector<Node>& nodes, float tX, float tK, float tF, float tKf) {
  const int N = s.N, nc = s.n_calls();
  // writer[img][p][t]: node that last wrote tile t (-1: the input);  readers[img][p]: calls that took K / V of the CURRENT content
  std::vector<int> writer[2];
  std::vector<std::vector<int>> readers[2];
  for (int i = 0; i < 2; ++i) { writer[i].assign((size_t)N * s.tiles(i), -1); readers[i].assign(N, {}); }
  std::vector<int> f_first((size_t)nc * N, -1);         // first F node of (call, pair); the 8 heads are consecutive
  std::vector<int> x_first((size_t)nc * N, -1);         // first X node of (call, pair)
  nodes.reserve(s.n_items());
  for (int c = 0; c < nc; ++c) {
    int layer, xi, si;
    s.call(c, layer, xi, si);
    const int gs = s.tiles(si), gx = s.tiles(xi);
    // the call whose X items computed this call's K / V partials in their tails (none: K items)
    int producer = -1;
    for (int q = 0; q < c && producer < 0; ++q) { int f[2]; s.folds(q, f); if (f[0] == c || f[1] == c) producer = q; }
    int fold[2];
    const int nfold = s.folds(c, fold);
    for (int p = 0; p < N; ++p) {
      const int k0 = (int)nodes.size();
      if (s.standalone_k(c)) {
        for (int t = 0; t < gs; ++t) {
          Node n{};
          n.what = PCT_K | (uint32_t)c << 4 | (uint32_t)p << 12 | (uint32_t)t << 20; n.signal = s.kcnt(c, p); n.dur = tK;
          for (auto& d : n.dep) d = PCT_NODEP;
          n.sig2[0] = n.sig2[1] = PCT_NODEP;
          const int w = writer[si][(size_t)p * gs + t];
          if (w >= 0) { n.dep[0] = nodes[w].signal | 1u << 20; n.pred.push_back(w); }
          nodes.push_back(n);
        }
        readers[si][p].push_back(c);
      }
      f_first[(size_t)c * N + p] = (int)nodes.size();
      for (int h = 0; h < 8; ++h) {
        Node n{};
        n.what = PCT_F | (uint32_t)c << 4 | (uint32_t)p << 12 | (uint32_t)h << 20; n.signal = s.fcnt(c, p); n.dur = tF;
        for (auto& d : n.dep) d = PCT_NODEP;
        n.sig2[0] = n.sig2[1] = PCT_NODEP;
        n.dep[0] = s.kcnt(c, p) | (uint32_t)gs << 20;     // gs partials: from the K items, or from the producer call's X items
        if (producer < 0) for (int t = 0; t < gs; ++t) n.pred.push_back(k0 + t);
        else for (int t = 0; t < gs; ++t) n.pred.push_back(x_first[(size_t)producer * N + p] + t);
        nodes.push_back(n);
      }
      // X items: their own F items, the previous writer of the tile, and (write-after-read) the F items of every OTHER call whose K
      // items read the content they overwrite
      std::vector<int> war;
      for (int rc : readers[xi][p]) if (rc != c) war.push_back(rc);
      x_first[(size_t)c * N + p] = (int)nodes.size();
      for (int g = 0; g < gx; ++g) {
        Node n{};
        n.what = PCT_X | (uint32_t)c << 4 | (uint32_t)p << 12 | (uint32_t)g << 20; n.signal = s.xflag(c, p, g); n.dur = tX + nfold * tKf;
        for (auto& d : n.dep) d = PCT_NODEP;
        for (int i = 0; i < 2; ++i) n.sig2[i] = fold[i] >= 0 ? s.kcnt(fold[i], p) : PCT_NODEP;
        int nd = 0;
        n.dep[nd++] = s.fcnt(c, p) | 8u << 20;
        for (int h = 0; h < 8; ++h) n.pred.push_back(f_first[(size_t)c * N + p] + h);
        const int w = writer[xi][(size_t)p * gx + g];
        if (w >= 0) { n.dep[nd++] = nodes[w].signal | 1u << 20; n.pred.push_back(w); }
        for (size_t k = 0; k < war.size() && nd < 4; ++k) {
          n.dep[nd++] = s.fcnt(war[k], p) | 8u << 20;
          for (int h = 0; h < 8; ++h) n.pred.push_back(f_first[(size_t)war[k] * N + p] + h);
        }
        nodes.push_back(n);
      }
      for (int g = 0; g < gx; ++g) writer[xi][(size_t)p * gx + g] = (int)nodes.size() - gx + g;
      readers[xi][p].clear();
      if (xi == si) readers[xi][p].clear();
    }
  }
}

// start order of a greedy list schedule on `workers` workers, longest remaining path first
std::vector<int> list_schedule(const std::vector<Node>& nodes, int workers) {
  const int n = (int)nodes.size();
  std::vector<std::vector<int>> succ(n);
  std::vector<int> ndep(n, 0);
  for (int i = 0; i < n; ++i) { ndep[i] = (int)nodes[i].pred.size(); for (int q : nodes[i].pred) succ[q].push_back(i); }
  std::vector<float> cp(n, 0.f);
  for (int i = n - 1; i >= 0; --i) {                     // creation order is topological
    float m = 0.f;
    for (int q : succ[i]) m = std::max(m, cp[q]);
    cp[i] = nodes[i].dur + m;
  }
  typedef std::pair<float, int> PI;
  std::priority_queue<PI> ready;                         // (priority, -index): ties broken towards the earlier item
  for (int i = 0; i < n; ++i) if (!ndep[i]) ready.push(PI(cp[i], -i));
  std::priority_queue<PI, std::vector<PI>, std::greater<PI>> running;   // (finish time, index)
  std::vector<int> order; order.reserve(n);
  float t = 0.f; int free_w = workers;
  while (!ready.empty() || !running.empty()) {
    while (free_w > 0 && !ready.empty()) {
      const int i = -ready.top().second; ready.pop();
      order.push_back(i); --free_w;
      running.push(PI(t + nodes[i].dur, i));
    }
    if (running.empty()) break;
    t = running.top().first;
    const int i = running.top().second; running.pop(); ++free_w;
    for (int q : succ[i]) if (--ndep[q] == 0) ready.push(PI(cp[q], -q));
  }
  return order;
}
}  // namespace

static bool plan_shape(const int* layer_is_cross, int n_layers, int N, int L, int S, PctShape& s) {
  s = PctShape{n_layers, N, {L, S}};
  if (!layer_is_cross || !s.ok()) return false;
  for (int i = 0; i < n_layers; ++i) if ((layer_is_cross[i] != 0) != ((i & 1) != 0)) return false;      // [self, cross] * P only
  return true;
}

extern "C" size_t loftr_coarse_plan_bytes(const int* layer_is_cross, int n_layers, int N, int L, int S) {
  PctShape s;
  return plan_shape(layer_is_cross, n_layers, N, L, S, s) ? pct_plan_bytes(s) : 0;
}

extern "C" int loftr_coarse_plan_build(const int* layer_is_cross, int n_layers, int N, int L, int S, int order, void* plan, size_t plan_bytes,
                                       void* stream) {
  LOFTR_CHECK_ARG(plan && (order == 0 || order == 1));
  PctShape s;
  if (!plan_shape(layer_is_cross, n_layers, N, L, S, s)) return LOFTR_ERR_UNSUPPORTED;
  if (plan_bytes < pct_plan_bytes(s)) return LOFTR_ERR_WORKSPACE;
  std::vector<Node> nodes;
  // estimated durations [us] of an X / K / F item with every CU busy (profiles/r06_pct_trace.txt)
  build_graph(s, nodes, 112.f, 40.f, 8.f, 30.f);
  if (nodes.size() != s.n_items()) return LOFTR_ERR_BAD_ARG;
  std::vector<int> ord;
  if (order == 0) ord = list_schedule(nodes, 256);
  else {
    // call order: per call all K items (tile-major over the pairs), then all F, then all X
    ord.reserve(nodes.size());
    size_t i = 0;
    for (int c = 0; c < s.n_calls(); ++c) {
      const size_t i0 = i;
      while (i < nodes.size() && (int)((nodes[i].what >> 4) & 255) == c) ++i;
      for (int type : {PCT_K, PCT_F, PCT_X})
        for (size_t k = i0; k < i; ++k) if ((int)(nodes[k].what & 15) == type) ord.push_back((int)k);
    }
  }
  if (ord.size() != nodes.size()) return LOFTR_ERR_BAD_ARG;
  std::vector<PctItem> items(1 + nodes.size());
  items[0] = PctItem{s.signature(order), (uint32_t)nodes.size(), {PCT_NODEP, PCT_NODEP, PCT_NODEP, PCT_NODEP}, {0, 0}};
  for (size_t k = 0; k < ord.size(); ++k) {
    const Node& n = nodes[ord[k]];
    items[1 + k] = PctItem{n.what, n.signal, {n.dep[0], n.dep[1], n.dep[2], n.dep[3]}, {n.sig2[0], n.sig2[1]}};
  }
  // a set-up call like loftr_transformer_prepare: synchronous with respect to the host vector
  if (hipMemcpyAsync(plan, items.data(), items.size() * sizeof(PctItem), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return LOFTR_ERR_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return LOFTR_ERR_LAUNCH;
  return LOFTR_OK;
}

extern "C" unsigned loftr_coarse_plan_signature(int n_layers, int N, int L, int S, int order) {
  return PctShape{n_layers, N, {L, S}}.signature(order);
}
