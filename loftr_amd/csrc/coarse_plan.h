// The persistent coarse transformer's work queue (encoder_fused.hip: coarse_persistent_kernel): item encoding, counter layout and
// the host-side launch description.  Shared by the planner (coarse_plan.hip), the launcher (encoder_fused.hip) and transformer.hip.
#pragma once
#include "gemm.h"

struct PctItem { uint32_t what; uint32_t signal; uint32_t dep[4]; uint32_t sig2[2]; };  // what: type | call << 4 | pair << 12 | index << 20; sig2: further counters an X item bumps (PCT_NODEP: none)
constexpr uint32_t PCT_NODEP = 0xffffffffu;            // dep: counter index (20 bits) | target << 20
constexpr int PCT_X = 0, PCT_K = 1, PCT_F = 2;
constexpr int PCT_MAX_LAYERS = 8, PCT_MAX_CALLS = 2 * PCT_MAX_LAYERS, PCT_TILE = 128, PCT_CNT0 = 4;
constexpr int PCT_MAX_PAIRS = 256, PCT_MAX_TILES = 4095;

// counters (uint32, zeroed before every launch): [0] queue head, [1] error word, then per X item a done flag, per (call, pair) the
// number of finished K items and of finished F items
struct PctShape {
  int n_layers, N, T[2];
  int n_calls() const { return 2 * n_layers; }
  int tiles(int img) const { return (T[img] + PCT_TILE - 1) / PCT_TILE; }
  int gmax() const { return tiles(0) > tiles(1) ? tiles(0) : tiles(1); }
  uint32_t xflag(int c, int p, int g) const { return (uint32_t)(PCT_CNT0 + (c * N + p) * gmax() + g); }
  uint32_t kcnt(int c, int p) const { return (uint32_t)(PCT_CNT0 + n_calls() * N * gmax() + c * N + p); }
  uint32_t fcnt(int c, int p) const { return kcnt(c, p) + (uint32_t)(n_calls() * N); }
  size_t n_counters() const { return (size_t)PCT_CNT0 + (size_t)n_calls() * N * (gmax() + 2); }
  // calls of layer pair i: A (self 0), B (self 1), C (0 <- 1), D (1 <- updated 0)          transformer.py:91-99
  void call(int c, int& layer, int& x_img, int& s_img) const {
    const int i = c / 4, k = c % 4;
    layer = 2 * i + (k >= 2); x_img = k & 1; s_img = k < 2 ? (k & 1) : 1 - (k & 1);
  }
  // K / V of a call's source: calls 0 and 1 (the first self layer) read the transformer's input and run K items; every later call's
  // source tile is the OUTPUT tile of an earlier X item, which computes that call's K / V partial in its tail while the tile is still in
  // registers ("fold"): X items of call B_i fold C_i; of C_i fold D_i and A_(i+1); of D_i fold B_(i+1)          transformer.py:91-99
  bool standalone_k(int c) const { return c < 2; }
  int folds(int c, int (&out)[2]) const {
    const int k = c % 4;
    int n = 0;
    if (k == 1) out[n++] = c + 1;
    if (k == 2) { out[n++] = c + 1; if (c + 2 < n_calls()) out[n++] = c + 2; }
    if (k == 3 && c + 2 < n_calls()) out[n++] = c + 2;
    for (int i = n; i < 2; ++i) out[i] = -1;
    return n;
  }
  size_t n_items() const {
    size_t n = 0;
    for (int c = 0; c < n_calls(); ++c) { int l, x, s; call(c, l, x, s); n += (size_t)N * ((standalone_k(c) ? tiles(s) : 0) + 8 + tiles(x)); }
    return n;
  }
  uint32_t signature(int order) const {
    uint32_t h = 0x9e3779b9u ^ 7u;                     // (7: plan format)
    for (uint32_t v : {(uint32_t)n_layers, (uint32_t)N, (uint32_t)T[0], (uint32_t)T[1], (uint32_t)order}) h = (h ^ v) * 0x01000193u + 0x7ed55d16u;
    return h | 1u;
  }
  bool ok() const {
    return n_layers >= 2 && n_layers % 2 == 0 && n_layers <= PCT_MAX_LAYERS && N >= 1 && N <= PCT_MAX_PAIRS && T[0] >= 1 && T[1] >= 1 &&
           gmax() <= PCT_MAX_TILES && n_counters() < (1u << 20);
  }
};
inline size_t pct_plan_bytes(const PctShape& s) { return (1 + s.n_items()) * sizeof(PctItem); }   // a header item, then the queue
// workspace of a launch: counters, then per (call, pair) the row-tile partials [8][gmax][33][32], KV [8][33][32] and P [256][256] SP
inline size_t pct_ws_bytes(const PctShape& s) {
  const size_t cp = (size_t)s.n_calls() * s.N;
  return align_up(s.n_counters() * 4, 256) + align_up(cp * 8 * s.gmax() * 33 * 32 * 4, 256) + align_up(cp * 8 * 33 * 32 * 4, 256) +
         align_up(cp * 65536 * 4, 256) + 1024;
}

struct PctLayerPtrs {
  const sp_t *wq, *w0, *w2, *wkv;
  const float *wq_s, *w0_s, *w2_s, *wkv_s, *merge_f32, *g1, *b1, *g2, *b2;
};
struct PctLaunch {
  PctShape shape;
  float* f32[2]; sp_t* sp[2]; const uint8_t* mask[2];
  PctLayerPtrs layer[PCT_MAX_LAYERS];
  const void* plan; size_t plan_bytes; unsigned plan_signature;
  void* ws; size_t ws_bytes;
  unsigned* status;                                      // null, or a device word that receives the error word (0 = ok)
  unsigned long long* trace;                             // null, or 4 x u64 per item (popped, ready, done [wall_clock64, 100 MHz], workgroup)
};
int launch_coarse_persistent(const PctLaunch& p, hipStream_t st);
