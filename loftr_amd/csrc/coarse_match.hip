// CoarseMatching (src/loftr/utils/coarse_matching.py) without ever running the reference's ~14
// elementwise passes over the [N, L, S] score volume.
//
// dual-softmax  (coarse_matching.py:105-119 + get_coarse_match :150-196,238-261)
//   pass A  score_stats : sim tile = <f0,f1>/(C*T) on the matrix cores (split-fp16 core, gemm.h); per-wave online
//                         (max, sum exp) of every row and column of the tile -> partials
//   merge   stats       : partials -> (max, 1/sum) per row and per column
//   pass B  score_conf  : recompute the tile (bitwise the same sim), conf = softmax_row * softmax_col,
//                         write conf_matrix ONCE (optional), per-wave row (max, first argmax) and
//                         column max partials of conf
//   merge   colmax      : column maxima of conf
//   select              : per row: global (max, first argmax), threshold / border / mutual-NN test,
//                         block-local exclusive scan of the survivors
//   scan + scatter      : ordered compaction -> b_ids, i_ids, j_ids, mconf, mkpts*_c (ascending (b,i))
//
// sinkhorn  (coarse_matching.py:121-143 + SuperGlue log_optimal_transport)
//   score_store         : Z = <f0,f1>/C written once into conf_out (it is the returned buffer anyway)
//   3 x (row LSE, col LSE) in the log domain with the dustbin row / column handled analytically
//   ot_finalize         : conf = exp(Z + u + v - norm) in place, optional [L+1, S+1] assignment
//                         matrix, dustbin prefilter, the same row/col max partials as pass B
//   then the same select / scan / scatter.
#include "gemm.h"

namespace {

using Cfg = GemmCfg<128, 128, 2, 2>;
constexpr float SENTINEL = -3.0e38f;          // marks out-of-range tile entries (never a real score)

struct Geometry {
  int N, L, S, C;
  int h0c, w0c, h1c, w1c;
  int PJ, PI;                                   // partials per row (col tiles * WN) / per col
};

__device__ __forceinline__ bool in_range(float v) { return v > -1.0e38f; }

// XCD-aware order of the (pair, row tile, col tile) space of the score GEMMs.  The unit of locality
// is an 8 x 8 super-tile of 128 x 128 tiles of one pair: its 64 workgroups are exactly what one XCD
// (32 CUs x 2) holds at a time and they share 8 + 8 descriptor panels (2 MB < the XCD's 4 MB L2).
// Units are dealt round-robin to the XCDs with the pair index fastest, so with N = 8 pairs every
// XCD works on its own pair.   launch: 1-D grid of score_grid(g) workgroups.
constexpr int ST = 8;
__host__ __device__ inline int score_units(const Geometry& g) {
  return g.N * ceil_div(ceil_div(g.L, Cfg::BM), ST) * ceil_div(ceil_div(g.S, Cfg::BN), ST);
}
inline unsigned score_grid(const Geometry& g) { return (unsigned)(NUM_XCD * ceil_div(score_units(g), NUM_XCD) * ST * ST); }
__device__ __forceinline__ bool score_tile(const Geometry& g, int& n, int& ti, int& tj) {
  const int id = blockIdx.x, xcd = id % NUM_XCD, slot = id / NUM_XCD;
  const int u = (slot / (ST * ST)) * NUM_XCD + xcd, within = slot % (ST * ST);
  if (u >= score_units(g)) return false;
  const int tiles_m = ceil_div(g.L, Cfg::BM), tiles_n = ceil_div(g.S, Cfg::BN);
  const int nst_j = ceil_div(tiles_n, ST);
  n = u % g.N;
  const int st = u / g.N;
  ti = (st / nst_j) * ST + within / ST;
  tj = (st % nst_j) * ST + within % ST;
  return ti < tiles_m && tj < tiles_n;
}

// exp for the softmax terms: v_exp_f32 on x * log2(e).  Arguments are <= 0 and the terms that
// matter have |x| small; worst-case relative error ~|x| * 1e-7, far inside the 1e-4 budget on conf.
__device__ __forceinline__ float fexp(float x) { return __expf(x); }

// acc -> sim in place: scale, padding mask (-1e9), out-of-range -> SENTINEL.
//   FULL: the tile lies inside [L, S] (block-uniform) -> no range tests.
template <bool HAS_MASK, bool FULL>
__device__ __forceinline__ void acc_to_sim(f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0, int L, int S,
                                           float scale, const uint8_t* __restrict__ mask0,
                                           const uint8_t* __restrict__ mask1) {
  const EpiLane<Cfg> e;
  bool rm[Cfg::TM][16];
  if (HAS_MASK) {
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + e.lrow + e.rr(i, r);
        rm[i][r] = (FULL || row < L) ? mask0[row] != 0 : false;
      }
  }
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + e.lcol + j * 32;
    const bool cok = FULL || col < S;
    bool cm = true;
    if (HAS_MASK) cm = cok ? mask1[col] != 0 : false;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] * scale;
        if (HAS_MASK) { if (!(rm[i][r] && cm)) v = LOFTR_NEG_INF; }   // masked_fill_(~(m0 x m1), -INF)  :115-118
        if (!FULL) { if (!(cok && m0 + e.lrow + e.rr(i, r) < L)) v = SENTINEL; }
        acc[i][j][r] = v;
      }
  }
}

// ------------------------------------------------------------------------------------------
// pass A epilogue: per-wave online (max, sum exp) of every row and column of the tile
template <bool FULL>
__device__ __forceinline__ void stats_epilogue(f32x16 (&acc)[Cfg::TM][Cfg::TN], const Geometry& g, int n, int ti,
                                               int tj, float2* __restrict__ rowpart, float2* __restrict__ colpart) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % Cfg::WM, wn = wave / Cfg::WM;
  const EpiLane<Cfg> e;
  const int m0 = ti * Cfg::BM, n0 = tj * Cfg::BN;
  // rows: reduce over the TN tiles of the lane and the 32 lanes of the half-wave
  const int pj = tj * Cfg::WN + wn;
  float2* rp = rowpart + ((long)n * g.PJ + pj) * g.L + m0;      // partials are strip-major: [n][strip][row]
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i) {
    f32x16 m = acc[i][0];
#pragma unroll
    for (int j = 1; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) m[r] = fmaxf(m[r], acc[i][j][r]);
    half_max16(m);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = 0.f;
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const float t = fexp(acc[i][j][r] - m[r]);
        s[r] += (FULL || in_range(acc[i][j][r])) ? t : 0.f;
      }
    }
    half_sum16(s);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trow = e.lrow + e.rr(i, r);
      if ((lane & 31) == 0 && (FULL || m0 + trow < g.L)) rp[trow] = make_float2(m[r], s[r]);
    }
  }
  // columns: reduce over the TM*16 rows of the lane and the other half-wave
  const int pi = ti * Cfg::WM + wm;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    float m = SENTINEL;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
    m = fmaxf(m, swap32(m));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = fexp(acc[i][j][r] - m);
        s += (FULL || in_range(acc[i][j][r])) ? t : 0.f;
      }
    s += swap32(s);
    const int col = n0 + e.lcol + j * 32;
    if (lane < 32 && (FULL || col < g.S)) colpart[((long)n * g.PI + pi) * g.S + col] = make_float2(m, s);
  }
}

template <bool HAS_MASK>
__global__ __launch_bounds__(Cfg::THREADS, 2) void score_stats_kernel(
    const sp_t* __restrict__ f0, const sp_t* __restrict__ f1, Geometry g, float scale,
    const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
    float2* __restrict__ rowpart, float2* __restrict__ colpart) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  int n, ti, tj;
  if (!score_tile(g, n, ti, tj)) return;
  const int m0 = ti * Cfg::BM, n0 = tj * Cfg::BN;
  const sp_t* a = f0 + (long)n * g.L * g.C;
  const sp_t* b = f1 + (long)n * g.S * g.C;
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(asrc_plain(a, g.C), b, g.C, g.L, g.S, g.C, m0, n0, lds, acc);
  const uint8_t* mk0 = HAS_MASK ? mask0 + (long)n * g.L : nullptr;
  const uint8_t* mk1 = HAS_MASK ? mask1 + (long)n * g.S : nullptr;
  if (m0 + Cfg::BM <= g.L && n0 + Cfg::BN <= g.S) {
    acc_to_sim<HAS_MASK, true>(acc, m0, n0, g.L, g.S, scale, mk0, mk1);
    stats_epilogue<true>(acc, g, n, ti, tj, rowpart, colpart);
  } else {
    acc_to_sim<HAS_MASK, false>(acc, m0, n0, g.L, g.S, scale, mk0, mk1);
    stats_epilogue<false>(acc, g, n, ti, tj, rowpart, colpart);
  }
}

// (max, sum) partials -> (max, 1/sum).   one thread per row (or column)
//   part [N][P][len] (strip-major: consecutive threads read consecutive addresses), stat [N][len]
__global__ void merge_stats_kernel(const float2* __restrict__ part, float2* __restrict__ stat, long rows, int P, int len) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const long n = i / len;
  const float2* p = part + (n * P) * len + (i - n * len);
  float m = SENTINEL;
  // eight independent loads in flight per thread (the grid is only rows / 256 workgroups); the clamped tail re-reads
  // the last partial, which is harmless for the max and masked out of the sum
  for (int k0 = 0; k0 < P; k0 += 8) {
    float2 e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = p[(long)min(k0 + u, P - 1) * len];
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(m, e[u].x);
  }
  float s = 0.f;
  for (int k0 = 0; k0 < P; k0 += 8) {
    float2 e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = p[(long)min(k0 + u, P - 1) * len];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (k0 + u < P && in_range(e[u].x)) ? e[u].y * fexp(e[u].x - m) : 0.f;
  }
  stat[i] = make_float2(m, 1.f / s);
}

// Row (max, first argmax) and column max partials of a tile of conf held in acc.
template <bool FULL>
__device__ __forceinline__ void conf_partials(f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0, int n,
                                              const Geometry& g, int bx, int by,
                                              float2* __restrict__ rowmax_part,
                                              float* __restrict__ colmax_part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % Cfg::WM, wn = wave / Cfg::WM;
  const int pj = bx * Cfg::WN + wn, pi = by * Cfg::WM + wm;
  const EpiLane<Cfg> e;
  float2* rp = rowmax_part + ((long)n * g.PJ + pj) * g.L + m0;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i) {
    f32x16 bv = acc[i][0];
#pragma unroll
    for (int j = 1; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[r] = fmaxf(bv[r], acc[i][j][r]);
    half_max16(bv);                                  // maximum over the wave's 64-column strip
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int bc = 0x7fffffff, bl = 0x7fffffff;          // first column that attains it / minus the last one
#pragma unroll
      for (int j = Cfg::TN - 1; j >= 0; --j)
        if (acc[i][j][r] == bv[r]) bc = n0 + e.lcol + j * 32;
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
        if (acc[i][j][r] == bv[r]) bl = -(n0 + e.lcol + j * 32);
      bc = half_min_i32(bc);
      bl = half_min_i32(bl);
      const int trow = e.lrow + e.rr(i, r);
      if ((lane & 31) == 0 && (FULL || m0 + trow < g.L))      // attained at two different columns: TIE_BIT (select_kernel)
        rp[trow] = make_float2(bv[r], __int_as_float(bc | (bc != -bl ? (1 << 30) : 0)));
    }
  }
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    float m = -1.f;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
    m = fmaxf(m, swap32(m));
    const int col = n0 + e.lcol + j * 32;
    if (lane < 32 && (FULL || col < g.S)) colmax_part[((long)n * g.PI + pi) * g.S + col] = m;
  }
}

// ------------------------------------------------------------------------------------------
// pass B epilogue: conf = softmax_row * softmax_col, written once
template <bool FULL>
__device__ __forceinline__ void conf_epilogue(f32x16 (&acc)[Cfg::TM][Cfg::TN], const Geometry& g, int n, int m0, int n0,
                                              const float2* __restrict__ rowstat, const float2* __restrict__ colstat,
                                              float* __restrict__ conf_out) {
  const EpiLane<Cfg> e;
  float2 cs[Cfg::TN];
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + e.lcol + j * 32;
    cs[j] = colstat[(long)n * g.S + (FULL ? col : min(col, g.S - 1))];
  }
  float* co = conf_out ? conf_out + ((long)n * g.L + m0) * g.S + n0 : nullptr;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trow = e.lrow + e.rr(i, r);
      const float2 rs = rowstat[(long)n * g.L + (FULL ? m0 + trow : min(m0 + trow, g.L - 1))];
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const float v = acc[i][j][r];
        // softmax(sim, dim=1) * softmax(sim, dim=2)          coarse_matching.py:119
        float c = (fexp(v - cs[j].x) * cs[j].y) * (fexp(v - rs.x) * rs.y);
        if (!FULL) { if (!in_range(v)) c = -1.f; }             // out of range: below any confidence
        if (co && (FULL || c >= 0.f)) co[(unsigned)(trow * g.S + e.lcol + j * 32)] = c;
        acc[i][j][r] = c;
      }
    }
}

template <bool HAS_MASK>
__global__ __launch_bounds__(Cfg::THREADS, 2) void score_conf_kernel(
    const sp_t* __restrict__ f0, const sp_t* __restrict__ f1, Geometry g, float scale,
    const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
    const float2* __restrict__ rowstat, const float2* __restrict__ colstat,
    float* __restrict__ conf_out, float2* __restrict__ rowmax_part, float* __restrict__ colmax_part) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  int n, ti, tj;
  if (!score_tile(g, n, ti, tj)) return;
  const int m0 = ti * Cfg::BM, n0 = tj * Cfg::BN;
  const sp_t* a = f0 + (long)n * g.L * g.C;
  const sp_t* b = f1 + (long)n * g.S * g.C;
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(asrc_plain(a, g.C), b, g.C, g.L, g.S, g.C, m0, n0, lds, acc);
  const uint8_t* mk0 = HAS_MASK ? mask0 + (long)n * g.L : nullptr;
  const uint8_t* mk1 = HAS_MASK ? mask1 + (long)n * g.S : nullptr;
  if (m0 + Cfg::BM <= g.L && n0 + Cfg::BN <= g.S) {
    acc_to_sim<HAS_MASK, true>(acc, m0, n0, g.L, g.S, scale, mk0, mk1);
    conf_epilogue<true>(acc, g, n, m0, n0, rowstat, colstat, conf_out);
    conf_partials<true>(acc, m0, n0, n, g, tj, ti, rowmax_part, colmax_part);
  } else {
    acc_to_sim<HAS_MASK, false>(acc, m0, n0, g.L, g.S, scale, mk0, mk1);
    conf_epilogue<false>(acc, g, n, m0, n0, rowstat, colstat, conf_out);
    conf_partials<false>(acc, m0, n0, n, g, tj, ti, rowmax_part, colmax_part);
  }
}

#include "score_sweep.h"      // namespace sweep: the stationary-operand sweep kernels (dual-softmax passes A / B, Sinkhorn score store)


__global__ void merge_colmax_kernel(const float* __restrict__ part, float* __restrict__ colmax, long cols, int P, int len) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cols) return;
  float m = -1.f;
  const long n = i / len;
  const float* p = part + (n * P) * len + (i - n * len);
  for (int k0 = 0; k0 < P; k0 += 8) {
    float e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = p[(long)min(k0 + u, P - 1) * len];
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(m, e[u]);
  }
  colmax[i] = m;
}

// ------------------------------------------------------------------------------------------
// valid[n] = (h0, w0, h1, w1) of the top-left-anchored valid rectangles of the padding masks,
// recovered like coarse_matching.py:37-38: p_m.sum(1).max(-1), p_m.sum(-1).max(-1).
__global__ void valid_hw_kernel(const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                int h0c, int w0c, int h1c, int w1c, int* __restrict__ valid) {
  const int n = blockIdx.x;
  __shared__ int res[4];
  if (threadIdx.x < 4) res[threadIdx.x] = 0;
  __syncthreads();
  for (int which = 0; which < 2; ++which) {
    const uint8_t* m = which ? mask1 + (long)n * h1c * w1c : mask0 + (long)n * h0c * w0c;
    const int h = which ? h1c : h0c, w = which ? w1c : w0c;
    for (int x = threadIdx.x; x < w; x += blockDim.x) {        // column sums -> valid height
      int s = 0;
      for (int y = 0; y < h; ++y) s += m[y * w + x] != 0;
      atomicMax(&res[which * 2 + 0], s);
    }
    for (int y = threadIdx.x; y < h; y += blockDim.x) {        // row sums -> valid width
      int s = 0;
      for (int x = 0; x < w; ++x) s += m[y * w + x] != 0;
      atomicMax(&res[which * 2 + 1], s);
    }
  }
  __syncthreads();
  if (threadIdx.x < 4) valid[n * 4 + threadIdx.x] = res[threadIdx.x];
}

struct SelectParams {
  Geometry g;
  float thr; int border;
  const int* valid;                 // [N,4] or null
  int panel_mode;                   // row partials carry the 32-column PANEL of the maximum, not its column (sweep pass B)
};

// python slice semantics of `m[b, lim:] = False` with lim = hv - bd possibly negative
__device__ __forceinline__ int upper_limit(int hv, int bd, int hc) {
  int lim = hv - bd;
  if (lim < 0) lim = max(lim + hc, 0);
  return lim;
}

// The three tests of get_coarse_match on one (row, column) candidate whose confidence is the row maximum `bv`:
//   1. confidence threshold (:172)  2. borders (:176-183)  3. mutual nearest neighbour (:187-189)
__device__ __forceinline__ bool candidate_ok(const SelectParams& sp, const float* __restrict__ colmax, int n, int i, int j, float bv) {
  const Geometry& g = sp.g;
  if (!(bv > sp.thr)) return false;
  if (sp.border > 0) {
    const int y0 = i / g.w0c, x0 = i % g.w0c, y1 = j / g.w1c, x1 = j % g.w1c;
    int l_h0, l_w0, l_h1, l_w1;
    if (sp.valid) {
      const int* v = sp.valid + n * 4;
      l_h0 = upper_limit(v[0], sp.border, g.h0c); l_w0 = upper_limit(v[1], sp.border, g.w0c);
      l_h1 = upper_limit(v[2], sp.border, g.h1c); l_w1 = upper_limit(v[3], sp.border, g.w1c);
    } else {
      l_h0 = g.h0c - sp.border; l_w0 = g.w0c - sp.border; l_h1 = g.h1c - sp.border; l_w1 = g.w1c - sp.border;
    }
    const int b = sp.border;
    if (!(y0 >= b && x0 >= b && y1 >= b && x1 >= b && y0 < l_h0 && x0 < l_w0 && y1 < l_h1 && x1 < l_w1)) return false;
  }
  return bv == colmax[(long)n * g.S + j];
}

// the part of the border / padding test that depends on the ROW alone
__device__ __forceinline__ bool row_ok(const SelectParams& sp, int n, int i) {
  const Geometry& g = sp.g;
  if (sp.border <= 0) return true;
  const int y0 = i / g.w0c, x0 = i % g.w0c;
  int l_h0 = g.h0c - sp.border, l_w0 = g.w0c - sp.border;
  if (sp.valid) { l_h0 = upper_limit(sp.valid[n * 4], sp.border, g.h0c); l_w0 = upper_limit(sp.valid[n * 4 + 1], sp.border, g.w0c); }
  return y0 >= sp.border && x0 >= sp.border && y0 < l_h0 && x0 < l_w0;
}

// one thread per row of the flattened [N*L] rows; 256 rows per block
//
// Exact ties.  The reference ANDs threshold, border and mutual-maximum masks over the whole row and takes the FIRST
// surviving column (`mask.max(dim=2)`, coarse_matching.py:187-193): when the row maximum is attained more than once
// and its first occurrence fails a test, a later tied column is still emitted.  The partials carry an "attained
// twice" flag (TIE_BIT); only for such rows, and only if the first candidate fails, the thread walks the row of
// conf_matrix for the first tied column that passes (needs the materialised conf_matrix; without it the row is
// dropped like any row whose arg-max fails).
__global__ __launch_bounds__(256) void select_kernel(SelectParams sp, const float2* __restrict__ rowmax_part,
                                                     const float* __restrict__ colmax, const float* __restrict__ conf,
                                                     int* __restrict__ cand_j, float* __restrict__ cand_conf,
                                                     int* __restrict__ cand_rank, int* __restrict__ block_count,
                                                     int* __restrict__ counts) {
  const Geometry& g = sp.g;
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  const long rows = (long)g.N * g.L;
  bool flag = false, walk = false;
  int bj = 0; float bv = 0.f; int n = 0;
  if (row < rows) {
    n = (int)(row / g.L);
    const int i = (int)(row - (long)n * g.L);
    const float2* p = rowmax_part + ((long)n * g.PJ) * g.L + i;
    bv = -1.f; bj = 0;
    bool tie = false;
    // eight partials in flight per thread (written as one dependent load -> compare chain the strips cost one DRAM
    // round trip each: the grid is only rows / 256 workgroups).  The tail re-reads the last strip (k0 + u >= PJ: skipped).
    for (int k0 = 0; k0 < g.PJ; k0 += 8) {
      float2 e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = p[(long)min(k0 + u, g.PJ - 1) * g.L];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                    // ascending column chunks: > keeps the first
        if (k0 + u >= g.PJ) continue;
        const int w = __float_as_int(e[u].y);
        if (e[u].x > bv) { bv = e[u].x; bj = w & ~sweep::TIE_BIT; tie = (w & sweep::TIE_BIT) != 0; }
        else if (e[u].x == bv) tie = true;
      }
    }
    if (sp.panel_mode) {        // the sweep tracked the first panel that attains the maximum: its first column, and whether
      const float* cr = conf + ((long)n * g.L + i) * g.S;      // the maximum occurs again, come from those 32 entries of conf
      const int c0 = bj * 32, c1 = min(g.S, c0 + 32);
      int first = -1;
      for (int j = c1 - 1; j >= c0; --j)
        if (cr[j] == bv) { tie = tie || first >= 0; first = j; }
      bj = first >= 0 ? first : c0;
    }
    flag = candidate_ok(sp, colmax, n, i, bj, bv);
    // rare: exact tie at the row maximum and the first one failed.  Not for a row that fails the border / padding test on its
    // OWN coordinates: no column can pass for it (every padded row of a MegaDepth-style batch is such a row, and its uniform
    // confidences tie everywhere: walking them was 3.3 ms of the 840 x 840 configuration, profiles/r03_kernel_stats_outdoor.txt)
    walk = !flag && tie && conf && bv > sp.thr && row_ok(sp, n, i);
  }
  // The walk itself is done by the whole wave for one row at a time (64 columns per step, coalesced) instead of by the row's
  // thread alone (one dependent load per column: 0.5 ms for a single 4800-column row).
  {
    unsigned long long todo = __ballot(walk);
    while (todo) {                                     // wave-uniform
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const long r_s = __shfl((int)(row >> 31), src) * (1L << 31) + (long)(__shfl((int)(row & 0x7fffffff), src));
      const int n_s = __shfl(n, src), bj_s = __shfl(bj, src);
      const float bv_s = __shfl(bv, src);
      const int i_s = (int)(r_s - (long)n_s * g.L);
      const float* cr = conf + r_s * g.S;
      const int lane_ = threadIdx.x & 63;
      int found = -1;
      for (int base = bj_s + 1; base < g.S && found < 0; base += 64) {
        const int j = base + lane_;
        const bool hit = j < g.S && cr[j] == bv_s && candidate_ok(sp, colmax, n_s, i_s, j, bv_s);
        const unsigned long long hb = __ballot(hit);
        if (hb) found = base + (__ffsll((long long)hb) - 1);
      }
      if (lane_ == src && found >= 0) { bj = found; flag = true; }
    }
  }
  // block-local exclusive scan of the flags (ballot per wave + wave offsets through LDS)
  __shared__ int wave_tot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(flag);
  const int within = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[wave] = __popcll(bal);
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += wave_tot[w];
  if (row < rows) {
    cand_j[row] = bj;
    cand_conf[row] = bv;
    cand_rank[row] = flag ? off + within : -1;
  }
  // per-pair match counts: one atomic per (wave, pair) instead of one per match (with a low threshold the per-match
  // atomics on N addresses were the whole kernel time).  A wave's 64 consecutive rows span one or two pairs when
  // L >= 64 and up to 64 when the coarse grid is tiny: peel one pair per iteration.
  {
    unsigned long long rest = bal;
    while (rest) {                                     // wave-uniform
      const int src = __ffsll((long long)rest) - 1;
      const int n_k = __shfl(n, src);
      const unsigned long long same = __ballot(flag && n == n_k);
      if (lane == 0) atomicAdd(&counts[1 + n_k], __popcll(same));
      rest &= ~same;
    }
  }
  if (threadIdx.x == 0) block_count[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// exclusive scan of the per-block counts (single block) + total
__global__ __launch_bounds__(1024) void scan_blocks_kernel(const int* __restrict__ block_count,
                                                           int* __restrict__ block_off, int nblk,
                                                           int* __restrict__ counts) {
  __shared__ int buf[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblk ? block_count[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {              // Hillis-Steele inclusive scan
      int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblk) block_off[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[0] = carry;
}

struct ScatterParams {
  Geometry g;
  float scale; const float* scale0; const float* scale1;
  loftr_match_out out;
};

__global__ __launch_bounds__(256) void scatter_kernel(ScatterParams sp, const int* __restrict__ cand_j,
                                                      const float* __restrict__ cand_conf,
                                                      const int* __restrict__ cand_rank,
                                                      const int* __restrict__ block_off) {
  const Geometry& g = sp.g;
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= (long)g.N * g.L) return;
  const int rk = cand_rank[row];
  if (rk < 0) return;
  const long dst = block_off[blockIdx.x] + rk;
  const int n = (int)(row / g.L);
  const int i = (int)(row - (long)n * g.L);
  const int j = cand_j[row];
  sp.out.b_ids[dst] = n;
  sp.out.i_ids[dst] = i;
  sp.out.j_ids[dst] = j;
  sp.out.mconf[dst] = cand_conf[row];
  // mkpts = stack([id % w, id // w]) * (scale * scale{0,1}[b])       coarse_matching.py:242-250
  float s0x = sp.scale, s0y = sp.scale, s1x = sp.scale, s1y = sp.scale;
  if (sp.scale0) { s0x = sp.scale * sp.scale0[n * 2]; s0y = sp.scale * sp.scale0[n * 2 + 1]; }
  if (sp.scale1) { s1x = sp.scale * sp.scale1[n * 2]; s1y = sp.scale * sp.scale1[n * 2 + 1]; }
  sp.out.mkpts0_c[dst * 2 + 0] = (float)(i % g.w0c) * s0x;
  sp.out.mkpts0_c[dst * 2 + 1] = (float)(i / g.w0c) * s0y;
  sp.out.mkpts1_c[dst * 2 + 0] = (float)(j % g.w1c) * s1x;
  sp.out.mkpts1_c[dst * 2 + 1] = (float)(j / g.w1c) * s1y;
}

#include "sinkhorn.h"         // Sinkhorn: iteration / finalize kernels (row-streaming and round-1 forms), prefilter, dustbins
#include "dual_softmax_bwd.h" // dual-softmax backward: the streaming passes after the recomputed statistics / scores
#include "sinkhorn_bwd.h"     // Sinkhorn backward: reverse mode through the unrolled iterations

// ------------------------------------------------------------------------------------------
// upper bound of the number of sweep work units (pair x column chunk x 256-row block)
inline size_t sweep_flag_count(int N, int L, int S) { return (size_t)N * (ceil_div(S, 32 * 16) + 1) * ceil_div(L, 256); }

struct MatchWs {
  float2 *rowpart, *colpart, *rowstat, *colstat, *rowmax_part;
  float *colmax_part, *colmax, *cand_conf;
  int *cand_j, *cand_rank, *block_count, *block_off, *valid, *sweep_flags;
  float *ot_u, *ot_v; float2* ot_part; uint8_t *rowkill, *colkill;
  sp_t *f0sp, *f1sp;             // SP copies of the descriptors (GEMM operands, gemm.h)
  bool ok;
};

Geometry make_geometry(const loftr_coarse_params& p) {
  Geometry g;
  g.N = p.N; g.C = p.C; g.h0c = p.h0c; g.w0c = p.w0c; g.h1c = p.h1c; g.w1c = p.w1c;
  g.L = p.h0c * p.w0c; g.S = p.h1c * p.w1c;
  g.PJ = ceil_div(g.S, Cfg::BN) * Cfg::WN;
  g.PI = ceil_div(g.L, Cfg::BM) * Cfg::WM;
  return g;
}

MatchWs carve(void* ws, size_t bytes, const Geometry& g) {
  WsAlloc wa(ws, bytes);
  MatchWs m;
  const size_t NL = (size_t)g.N * g.L, NS = (size_t)g.N * g.S;
  m.rowpart = wa.take<float2>(NL * g.PJ);
  const size_t PIw = (size_t)ceil_div(g.L, 256) * 8;  // column partials per column: the sweep kernels write one per 32-row wave of every 256-row block
  m.colpart = wa.take<float2>(NS * (PIw > (size_t)g.PI ? PIw : (size_t)g.PI));
  m.rowstat = wa.take<float2>(NL);
  m.colstat = wa.take<float2>(NS);
  m.rowmax_part = wa.take<float2>(NL * g.PJ);
  m.colmax_part = wa.take<float>(NS * (PIw > (size_t)g.PI ? PIw : (size_t)g.PI));
  m.colmax = wa.take<float>(NS);
  m.cand_conf = wa.take<float>(NL);
  m.cand_j = wa.take<int>(NL);
  m.cand_rank = wa.take<int>(NL);
  const size_t nblk = (NL + 255) / 256;
  m.block_count = wa.take<int>(nblk);
  m.block_off = wa.take<int>(nblk);
  m.valid = wa.take<int>((size_t)g.N * 4);
  m.sweep_flags = wa.take<int>(sweep_flag_count(g.N, g.L, g.S));
  m.ot_u = wa.take<float>((size_t)g.N * (g.L + 1));
  m.ot_v = wa.take<float>((size_t)g.N * (g.S + 1));
  m.ot_part = wa.take<float2>((size_t)g.N * (g.S + 1) * OT_RCH);
  m.rowkill = wa.take<uint8_t>(NL);
  m.colkill = wa.take<uint8_t>(NS);
  m.f0sp = wa.take<sp_t>(NL * g.C);
  m.f1sp = wa.take<sp_t>(NS * g.C);
  m.ok = wa.ok();
  return m;
}

size_t match_ws_bytes(int N, int L, int S, int C) {
  const size_t PJ = (size_t)ceil_div(S, Cfg::BN) * Cfg::WN;
  const size_t PIw = (size_t)ceil_div(L, 256) * 8, PIt = (size_t)ceil_div(L, Cfg::BM) * Cfg::WM;
  const size_t PI = PIw > PIt ? PIw : PIt;
  const size_t NL = (size_t)N * L, NS = (size_t)N * S;
  size_t b = 0;
  b += NL * PJ * 8 * 2 + NS * PI * 8 + NS * PI * 4;
  b += NL * 8 + NS * 8 + NS * 4 + NL * 4 * 3 + ((NL + 255) / 256) * 8 + (size_t)N * 16;
  b += (size_t)N * (L + 1) * 4 + (size_t)N * (S + 1) * 4 + (size_t)N * (S + 1) * OT_RCH * 8 + NL + NS;
  b += (NL + NS) * (size_t)C * 4;
  b += sweep_flag_count(N, L, S) * 4;
  return b + 36 * 256;     // alignment slack of the bump allocator
}

bool params_ok(const loftr_coarse_params* p, const loftr_match_out* o) {
  if (!p || !o) return false;
  if (p->N < 0 || p->h0c <= 0 || p->w0c <= 0 || p->h1c <= 0 || p->w1c <= 0) return false;
  if ((p->mask0 == nullptr) != (p->mask1 == nullptr)) return false;
  return o->b_ids && o->i_ids && o->j_ids && o->mconf && o->mkpts0_c && o->mkpts1_c && o->counts;
}

// select -> scan -> scatter on the row/col max partials of conf
int select_and_compact(const Geometry& g, const loftr_coarse_params& p, const loftr_match_out& out,
                       const MatchWs& w, const float* conf, hipStream_t st, bool panel_mode = false) {
  const long NL = (long)g.N * g.L, NS = (long)g.N * g.S;
  hipLaunchKernelGGL(merge_colmax_kernel, dim3(ceil_div((int)NS, 256)), dim3(256), 0, st, w.colmax_part, w.colmax, NS, g.PI, g.S);
  const int* valid = nullptr;
  if (p.mask0) {
    hipLaunchKernelGGL(valid_hw_kernel, dim3(g.N), dim3(128), 0, st, p.mask0, p.mask1, g.h0c, g.w0c, g.h1c, g.w1c, w.valid);
    valid = w.valid;
  }
  (void)hipMemsetAsync(out.counts, 0, sizeof(int32_t) * (1 + g.N), st);
  const int nblk = (int)((NL + 255) / 256);
  SelectParams sp{g, p.thr, p.border_rm, valid, panel_mode ? 1 : 0};
  hipLaunchKernelGGL(select_kernel, dim3(nblk), dim3(256), 0, st, sp, w.rowmax_part, w.colmax, conf, w.cand_j, w.cand_conf,
                     w.cand_rank, w.block_count, out.counts);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, w.block_count, w.block_off, nblk, out.counts);
  ScatterParams sc{g, p.scale, p.scale0, p.scale1, out};
  hipLaunchKernelGGL(scatter_kernel, dim3(nblk), dim3(256), 0, st, sc, w.cand_j, w.cand_conf, w.cand_rank, w.block_off);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

}  // namespace

extern "C" size_t loftr_coarse_match_workspace_bytes(int N, int L, int S, int C) {
  if (N <= 0 || L <= 0 || S <= 0 || C <= 0) return 0;
  return match_ws_bytes(N, L, S, C);
}

namespace {
// Work decomposition of the sweep kernels: RB 256-row blocks x NCH column chunks of PPC 32-column panels per pair.
// A function of (L, S) only -- never of N -- so that a pair's partial sums are merged in the same order whatever
// batch it is in (bitwise batch invariance, tests/test_hip_parity.py::test_batch_consistency_full_size).
void sweep_plan(const Geometry& g, sweep::Args& a) {
  a.N = g.N; a.L = g.L; a.S = g.S;
  a.RB = ceil_div(g.L, sweep::BR);
  a.NP = ceil_div(g.S, sweep::PC);
  a.NCH = ceil_div(a.NP, 30);                 // <= 30 panels per chunk (LDS tables hold 32); S = 4800: 5 chunks of 30,
  a.PPC = ceil_div(a.NP, a.NCH);              //   8 pairs x 19 row blocks x 5 = 760 workgroups = 2.97 rounds of the 256 CUs
  a.NCH = ceil_div(a.NP, a.PPC);
}

// descriptors -> SP (both images in one launch)
int stage_descriptors(const float* f0, const float* f1, const Geometry& g, const MatchWs& w, hipStream_t st) {
  SpJobs j; j.n = 2;
  j.src[0] = f0; j.dst[0] = w.f0sp; j.rows[0] = g.N * g.L; j.K[0] = g.C; j.ld[0] = g.C;
  j.src[1] = f1; j.dst[1] = w.f1sp; j.rows[1] = g.N * g.S; j.K[1] = g.C; j.ld[1] = g.C;
  return launch_sp_convert(j, st);
}
}  // namespace

extern "C" int loftr_coarse_match_dual_softmax(const float* feat_c0, const float* feat_c1,
                                               const loftr_coarse_params* p, float temperature,
                                               float* conf_out, const loftr_match_out* out, void* ws,
                                               size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(feat_c0 && feat_c1 && params_ok(p, out) && temperature > 0.f);
  if (p->C % 32 != 0) return LOFTR_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (p->N == 0) { (void)hipMemsetAsync(out->counts, 0, sizeof(int32_t), st); return LOFTR_OK; }
  LOFTR_CHECK_ARG(ws != nullptr);
  const Geometry g = make_geometry(*p);
  MatchWs w = carve(ws, ws_bytes, g);
  if (!w.ok) return LOFTR_ERR_WORKSPACE;
  { int rc = stage_descriptors(feat_c0, feat_c1, g, w, st); if (rc) return rc; }
  // feat / sqrt(C) on both sides, then / temperature                 coarse_matching.py:108-114
  const float scale = 1.f / ((float)g.C * temperature);
  const long NL = (long)g.N * g.L, NS = (long)g.N * g.S;
  if (g.C == 256) {
    // stationary-operand sweep (sweep:: above).  Partials: one per (row, column chunk) and per (column, row block).
    Geometry gs = g;
    sweep::Args a{};
    sweep_plan(g, a);
    gs.PJ = a.NCH; gs.PI = a.RB * sweep::W;
    a.f0 = w.f0sp; a.f1 = w.f1sp; a.scale = scale; a.mask0 = p->mask0; a.mask1 = p->mask1;
    a.rowpart = w.rowpart; a.colpart = w.colpart; a.rowstat = w.rowstat; a.colstat = w.colstat;
    a.conf = conf_out; a.rowmax_part = w.rowmax_part; a.colmax_part = w.colmax_part;
    const dim3 grid(NUM_XCD * ceil_div(g.N * a.NCH, NUM_XCD) * a.RB), block(512);
    {
      TimedLaunch tl(LOFTR_T_SCORE_STATS, st);
      if (p->mask0 || !SWEEP_PROBE_FAST) {
        a.exact_flags = nullptr;                       // padding masks (-1e9 fills): per-row / per-column references throughout
        if (p->mask0) hipLaunchKernelGGL((sweep::score_sweep_kernel<0, true, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((sweep::score_sweep_kernel<0, false, false>), grid, block, 0, st, a);
      } else {
        a.exact_flags = w.sweep_flags;                 // shared-reference variant first, exact variant for the units it gave up on
        (void)hipMemsetAsync(w.sweep_flags, 0, sizeof(int) * (size_t)g.N * a.NCH * a.RB, st);
        hipLaunchKernelGGL((sweep::score_sweep_kernel<0, false, true>), grid, block, 0, st, a);
        hipLaunchKernelGGL((sweep::score_sweep_kernel<0, false, false>), grid, block, 0, st, a);
      }
    }
    hipLaunchKernelGGL(merge_stats_kernel, dim3(ceil_div((int)NL, 256)), dim3(256), 0, st, w.rowpart, w.rowstat, NL, gs.PJ, g.L);
    hipLaunchKernelGGL(merge_stats_kernel, dim3(ceil_div((int)NS, 256)), dim3(256), 0, st, w.colpart, w.colstat, NS, gs.PI, g.S);
    {
      TimedLaunch tl(LOFTR_T_SCORE_CONF, st);
      a.exact_flags = nullptr;
      if (p->mask0) hipLaunchKernelGGL((sweep::score_sweep_kernel<1, true, false, true>), grid, block, 0, st, a);
      else if (conf_out) hipLaunchKernelGGL((sweep::score_sweep_kernel<1, false, false, false>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((sweep::score_sweep_kernel<1, false, false, true>), grid, block, 0, st, a);
    }
    LOFTR_CHECK_LAUNCH();
    return select_and_compact(gs, *p, *out, w, conf_out, st, conf_out != nullptr && !p->mask0);
  }
  // other descriptor widths: the tiled two-pass kernels
  const dim3 sgrid(score_grid(g)), block(Cfg::THREADS);
  {
    TimedLaunch tl(LOFTR_T_SCORE_STATS, st);
    if (p->mask0)
      hipLaunchKernelGGL((score_stats_kernel<true>), sgrid, block, 0, st, w.f0sp, w.f1sp, g, scale, p->mask0, p->mask1, w.rowpart, w.colpart);
    else
      hipLaunchKernelGGL((score_stats_kernel<false>), sgrid, block, 0, st, w.f0sp, w.f1sp, g, scale, p->mask0, p->mask1, w.rowpart, w.colpart);
  }
  hipLaunchKernelGGL(merge_stats_kernel, dim3(ceil_div((int)NL, 256)), dim3(256), 0, st, w.rowpart, w.rowstat, NL, g.PJ, g.L);
  hipLaunchKernelGGL(merge_stats_kernel, dim3(ceil_div((int)NS, 256)), dim3(256), 0, st, w.colpart, w.colstat, NS, g.PI, g.S);
  {
    TimedLaunch tl(LOFTR_T_SCORE_CONF, st);
    if (p->mask0)
      hipLaunchKernelGGL((score_conf_kernel<true>), sgrid, block, 0, st, w.f0sp, w.f1sp, g, scale, p->mask0, p->mask1, w.rowstat, w.colstat, conf_out, w.rowmax_part, w.colmax_part);
    else
      hipLaunchKernelGGL((score_conf_kernel<false>), sgrid, block, 0, st, w.f0sp, w.f1sp, g, scale, p->mask0, p->mask1, w.rowstat, w.colstat, conf_out, w.rowmax_part, w.colmax_part);
  }
  LOFTR_CHECK_LAUNCH();
  return select_and_compact(g, *p, *out, w, conf_out, st);
}


namespace {
// Work decomposition of the Sinkhorn iteration passes (a function of the geometry only) and the iteration loop itself, shared
// by the forward and by the backward's re-creation of (u_t, v_t).
// wide rows (up to 12 x 1024 columns): ONE workgroup per CU, one row per round, the next row prefetched.  Measured at 2 x 11025^2
// (936 MB per pass): iteration 272 us with 512 threads, 321 with 1024; last pass (reads + writes) 786 / 575 us.  Round 6: two rows per
// round (twice the bytes in flight) 370 us with 512 threads, 429 with 1024 -- a round is bound by its two block reductions and the
// exponentials between them, not by the loads (tools/gpu/r6_ot_wide.sh).
#ifndef OT_WIDE_NT
#define OT_WIDE_NT 512
#endif
#ifndef OT_WIDE_NT_FINAL
#define OT_WIDE_NT_FINAL 1024
#endif
struct OtPlan { bool rowstream, fused, wide, aligned; int wgs, rpws, wgp, rpw, cpt; };
OtPlan ot_plan(const Geometry& g) {
  OtPlan p{};
  // row-streaming passes (otp::ot_pass_kernel): at most 5 x 1024 columns incl. the dustbin with 256 threads (indoor), 6 x 2048 with
  // 512 (outdoor 105 x 105); rows of any alignment
  p.rowstream = g.S + 1 <= 4 * 512 * 6;
  p.wide = g.S + 1 > 4 * 256 * 5;
  p.aligned = (g.S & 3) == 0;
  if (p.rowstream) {
    const int R = p.wide ? 1 : 2;
    const int capP = [&] { const int a = ceil_div(g.L, 256) * 8; const int c = a > g.PI ? a : g.PI; return c < OT_RCH ? c : OT_RCH; }();
    int wgs = (p.wide ? 256 : 768) / g.N;                  // ~3 workgroups of 4 waves per CU over the batch (wide: one)
    wgs = wgs < 1 ? 1 : (wgs > capP ? capP : wgs);
    if (wgs > ceil_div(g.L, R)) wgs = ceil_div(g.L, R);
    p.rpws = ceil_div(ceil_div(g.L, wgs), R) * R;
    p.wgs = ceil_div(g.L, p.rpws);
  }
  // fused iteration (one pass over Z): up to 19 x 256 (indoor) / 44 x 256 (outdoor 840 x 840) columns incl. the dustbin
  p.cpt = ceil_div(g.S + 1, 256);
  p.fused = p.cpt <= 44;
  int wgp = 512 / (g.N > 0 ? g.N : 1);                     // ~2 workgroups per CU over the batch
  wgp = wgp < 1 ? 1 : (wgp > OT_RCH ? OT_RCH : wgp);       // the partial buffer holds OT_RCH rows per column
  if (wgp > ceil_div(g.L, 4)) wgp = ceil_div(g.L, 4);
  int rpw = ceil_div(g.L, wgp);
  p.rpw = ceil_div(rpw, 4) * 4;
  p.wgp = ceil_div(g.L, p.rpw);
  return p;
}
// one row-streaming pass in the variant the plan names
template <bool FINAL>
void ot_pass_launch(const OtPlan& pl, const Geometry& g, hipStream_t st, float* z, float bin_score, float norm, const float* v, float* u,
                    float2* part, const uint8_t* rk, const uint8_t* ck, float* assign, float2* rowmax_part, float* colmax_part) {
  const dim3 grid(pl.wgs, g.N);
  constexpr int WNT = FINAL ? OT_WIDE_NT_FINAL : OT_WIDE_NT;
#define OT_PASS(NT_, G4_, R_, AL_, PF_)                                                                                        \
  hipLaunchKernelGGL((otp::ot_pass_kernel<NT_, G4_, R_, FINAL, AL_, PF_>), grid, dim3(NT_), 0, st, z, g, bin_score, norm, v, u, part, \
                     pl.rpws, rk, ck, assign, rowmax_part, colmax_part)
  if (!pl.wide) { if (pl.aligned) OT_PASS(256, 5, 2, true, false); else OT_PASS(256, 5, 2, false, false); }
  else { if (pl.aligned) OT_PASS(WNT, 12 * 256 / WNT, 1, true, true); else OT_PASS(WNT, 12 * 256 / WNT, 1, false, true); }
#undef OT_PASS
}
// u = v = 0, then `iters` iterations on z (the scaled, mask-filled scores) in w.ot_u / w.ot_v.  save_u [iters][N (L+1)] /
// save_v [iters + 1][N (S+1)] (or null): the potentials after every iteration (save_v[0] = 0), for the backward.
void ot_iterate(const Geometry& g, const OtPlan& pl, float* z, const MatchWs& w, float bin_score, float norm, int iters, hipStream_t st,
                float* save_u, float* save_v) {
  const size_t ub = sizeof(float) * g.N * (g.L + 1), vb = sizeof(float) * g.N * (g.S + 1);
  (void)hipMemsetAsync(w.ot_u, 0, ub, st);
  (void)hipMemsetAsync(w.ot_v, 0, vb, st);
  if (save_v) (void)hipMemsetAsync(save_v, 0, vb, st);
  const long cols = (long)g.N * (g.S + 1);
  for (int it = 0; it < iters; ++it) {
    if (pl.rowstream) {
      ot_pass_launch<false>(pl, g, st, z, bin_score, norm, w.ot_v, w.ot_u, w.ot_part, nullptr, nullptr, nullptr, nullptr, nullptr);
      hipLaunchKernelGGL(ot_col_merge2_kernel, dim3(ceil_div((int)cols, 256)), dim3(256), 0, st, w.ot_part, g, bin_score, norm, pl.wgs, w.ot_u, w.ot_v);
    } else if (pl.fused) {
      if (pl.cpt <= 19) hipLaunchKernelGGL((ot_iter_kernel<19, 4>), dim3(pl.wgp, g.N), dim3(256), 0, st, z, g, bin_score, norm, w.ot_v, w.ot_u, w.ot_part, pl.rpw);
      else hipLaunchKernelGGL((ot_iter_kernel<44, 2>), dim3(pl.wgp, g.N), dim3(256), 0, st, z, g, bin_score, norm, w.ot_v, w.ot_u, w.ot_part, pl.rpw);
      hipLaunchKernelGGL(ot_col_merge2_kernel, dim3(ceil_div((int)cols, 256)), dim3(256), 0, st, w.ot_part, g, bin_score, norm, pl.wgp, w.ot_u, w.ot_v);
    } else {
      hipLaunchKernelGGL(ot_row_lse_kernel, dim3(ceil_div(g.L + 1, 4), g.N), dim3(256), 0, st, z, g, bin_score, norm, w.ot_v, w.ot_u);
      hipLaunchKernelGGL(ot_col_part_kernel, dim3(ceil_div(g.S + 1, 64), OT_RCH, g.N), dim3(256), 0, st, z, g, bin_score, w.ot_u, w.ot_part);
      hipLaunchKernelGGL(ot_col_merge_kernel, dim3(ceil_div((int)cols, 256)), dim3(256), 0, st, w.ot_part, g, norm, w.ot_v);
    }
    if (save_u) (void)hipMemcpyAsync(save_u + (size_t)it * g.N * (g.L + 1), w.ot_u, ub, hipMemcpyDeviceToDevice, st);
    if (save_v) (void)hipMemcpyAsync(save_v + (size_t)(it + 1) * g.N * (g.S + 1), w.ot_v, vb, hipMemcpyDeviceToDevice, st);
  }
}
}  // namespace

extern "C" int loftr_coarse_match_sinkhorn(const float* feat_c0, const float* feat_c1,
                                           const loftr_coarse_params* p, float bin_score, int iters,
                                           int prefilter, float* conf_out, float* assign_out,
                                           const loftr_match_out* out, void* ws, size_t ws_bytes,
                                           void* stream) {
  LOFTR_CHECK_ARG(feat_c0 && feat_c1 && params_ok(p, out) && conf_out && iters >= 0);
  if (p->C % 32 != 0) return LOFTR_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (p->N == 0) { (void)hipMemsetAsync(out->counts, 0, sizeof(int32_t), st); return LOFTR_OK; }
  LOFTR_CHECK_ARG(ws != nullptr);
  const Geometry g = make_geometry(*p);
  MatchWs w = carve(ws, ws_bytes, g);
  if (!w.ok) return LOFTR_ERR_WORKSPACE;
  { int rc = stage_descriptors(feat_c0, feat_c1, g, w, st); if (rc) return rc; }
  const float scale = 1.f / (float)g.C;                    // no temperature   coarse_matching.py:123
  const float norm = -logf((float)(g.L + g.S));            // SuperGlue: norm = -log(m + n)
  const dim3 grid(ceil_div(g.S, Cfg::BN), ceil_div(g.L, Cfg::BM), g.N), sgrid(score_grid(g)), block(Cfg::THREADS);
  { TimedLaunch tl(LOFTR_T_OT_STORE, st);
    if (g.C == 256) {                                      // the stationary-operand sweep with a store-only epilogue
      sweep::Args a{};
      sweep_plan(g, a);
      a.f0 = w.f0sp; a.f1 = w.f1sp; a.scale = scale; a.mask0 = p->mask0; a.mask1 = p->mask1; a.conf = conf_out;
      const dim3 swgrid(NUM_XCD * ceil_div(g.N * a.NCH, NUM_XCD) * a.RB);
      if (p->mask0) hipLaunchKernelGGL((sweep::score_sweep_kernel<2, true>), swgrid, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((sweep::score_sweep_kernel<2, false>), swgrid, dim3(512), 0, st, a);
    } else {
      hipLaunchKernelGGL(score_store_kernel, sgrid, block, 0, st, w.f0sp, w.f1sp, g, scale, p->mask0, p->mask1, conf_out);
    }
  }
  const OtPlan plan = ot_plan(g);
  ot_iterate(g, plan, conf_out, w, bin_score, norm, iters, st, nullptr, nullptr);
  const bool rowstream = plan.rowstream;
  const int wgs = plan.wgs;
  const uint8_t *rk = nullptr, *ck = nullptr;
  if (prefilter) {
    hipLaunchKernelGGL(ot_rowkill_kernel, dim3(ceil_div(g.L, 4), g.N), dim3(256), 0, st, conf_out, g, bin_score, w.ot_u, w.ot_v, w.rowkill);
    hipLaunchKernelGGL(ot_colkill_kernel, dim3(ceil_div(g.S, 256), g.N), dim3(256), 0, st, conf_out, g, bin_score, w.ot_u, w.ot_v, w.colkill);
    rk = w.rowkill; ck = w.colkill;
  }
  if (assign_out)
    hipLaunchKernelGGL(ot_assign_bins_kernel, dim3(ceil_div((g.L > g.S ? g.L : g.S) + 1, 256), g.N), dim3(256), 0, st, g, bin_score, norm, w.ot_u, w.ot_v, assign_out);
  if (rowstream) {
    ot_pass_launch<true>(plan, g, st, conf_out, bin_score, norm, w.ot_v, w.ot_u, nullptr, rk, ck, assign_out, w.rowmax_part, w.colmax_part);
    LOFTR_CHECK_LAUNCH();
    Geometry gs = g;
    gs.PJ = 1; gs.PI = wgs;                                // one row partial per row, one column-maximum partial per workgroup
    return select_and_compact(gs, *p, *out, w, conf_out, st);
  }
  hipLaunchKernelGGL(ot_finalize_kernel, grid, block, 0, st, conf_out, g, norm, w.ot_u, w.ot_v, rk, ck, assign_out, w.rowmax_part, w.colmax_part);
  LOFTR_CHECK_LAUNCH();
  return select_and_compact(g, *p, *out, w, conf_out, st);
}

// ---- backward of the dual-softmax confidence ---------------------------------------------------------------------------
// dsim [N, L, S] <- dL/d sim_matrix given grad_conf = dL/d conf_matrix (dual_softmax_bwd.h).  The caller finishes with
// the two plain GEMMs  dL/dfeat_c0 = dsim feat_c1 / (C T),  dL/dfeat_c1 = dsim^T feat_c0 / (C T)  (library GEMMs).
// Workspace: loftr_coarse_match_workspace_bytes (the forward's).
extern "C" int loftr_dual_softmax_bwd(const float* feat_c0, const float* feat_c1, const loftr_coarse_params* p, float temperature,
                                      const float* grad_conf, float* dsim, void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(feat_c0 && feat_c1 && p && grad_conf && dsim && temperature > 0.f);
  LOFTR_CHECK_ARG(p->N >= 0 && p->h0c > 0 && p->w0c > 0 && p->h1c > 0 && p->w1c > 0 && (p->mask0 == nullptr) == (p->mask1 == nullptr));
  if (p->C % 32 != 0) return LOFTR_ERR_UNSUPPORTED;
  if (p->N == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(ws != nullptr);
  hipStream_t st = (hipStream_t)stream;
  const Geometry g = make_geometry(*p);
  MatchWs w = carve(ws, ws_bytes, g);
  if (!w.ok) return LOFTR_ERR_WORKSPACE;
  { int rc = stage_descriptors(feat_c0, feat_c1, g, w, st); if (rc) return rc; }
  const float scale = 1.f / ((float)g.C * temperature);
  const long NL = (long)g.N * g.L, NS = (long)g.N * g.S;
  if (g.C == 256) {                                        // the forward's sweeps: statistics, then the store-only pass
    sweep::Args a{};
    sweep_plan(g, a);
    a.f0 = w.f0sp; a.f1 = w.f1sp; a.scale = scale; a.mask0 = p->mask0; a.mask1 = p->mask1;
    a.rowpart = w.rowpart; a.colpart = w.colpart; a.rowstat = w.rowstat; a.colstat = w.colstat;
    a.conf = dsim; a.exact_flags = nullptr;
    const dim3 grid(NUM_XCD * ceil_div(g.N * a.NCH, NUM_XCD) * a.RB), block(512);
    if (p->mask0) hipLaunchKernelGGL((sweep::score_sweep_kernel<0, true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((sweep::score_sweep_kernel<0, false, false>), grid, block, 0, st, a);
    hipLaunchKernelGGL(merge_stats_kernel, dim3(ceil_div((int)NL, 256)), dim3(256), 0, st, w.rowpart, w.rowstat, NL, a.NCH, g.L);
    hipLaunchKernelGGL(merge_stats_kernel, dim3(ceil_div((int)NS, 256)), dim3(256), 0, st, w.colpart, w.colstat, NS, a.RB * sweep::W, g.S);
    if (p->mask0) hipLaunchKernelGGL((sweep::score_sweep_kernel<2, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((sweep::score_sweep_kernel<2, false>), grid, block, 0, st, a);
  } else {
    const dim3 sgrid(score_grid(g)), block(Cfg::THREADS);
    if (p->mask0)
      hipLaunchKernelGGL((score_stats_kernel<true>), sgrid, block, 0, st, w.f0sp, w.f1sp, g, scale, p->mask0, p->mask1, w.rowpart, w.colpart);
    else
      hipLaunchKernelGGL((score_stats_kernel<false>), sgrid, block, 0, st, w.f0sp, w.f1sp, g, scale, p->mask0, p->mask1, w.rowpart, w.colpart);
    hipLaunchKernelGGL(merge_stats_kernel, dim3(ceil_div((int)NL, 256)), dim3(256), 0, st, w.rowpart, w.rowstat, NL, g.PJ, g.L);
    hipLaunchKernelGGL(merge_stats_kernel, dim3(ceil_div((int)NS, 256)), dim3(256), 0, st, w.colpart, w.colstat, NS, g.PI, g.S);
    hipLaunchKernelGGL(score_store_kernel, sgrid, block, 0, st, w.f0sp, w.f1sp, g, scale, p->mask0, p->mask1, dsim);
  }
  float* r = w.ot_u;                                       // Sinkhorn's buffers are idle on this path: [N (L+1)], [N (S+1)],
  float* c = w.ot_v;                                       //   [N (S+1) OT_RCH] float2 >= [N RCH S] floats
  float* part = reinterpret_cast<float*>(w.ot_part);
  static_assert(dsb::RCH <= 2 * OT_RCH, "column partials must fit Sinkhorn's partial buffer");
  hipLaunchKernelGGL(dsb::row_dot_kernel, dim3((unsigned)((NL + 3) / 4)), dim3(256), 0, st, dsim, grad_conf, g, w.rowstat, w.colstat, r);
  hipLaunchKernelGGL(dsb::col_dot_part_kernel, dim3(ceil_div(g.S, 256), dsb::RCH, g.N), dim3(256), 0, st, dsim, grad_conf, g, w.rowstat,
                     w.colstat, part);
  hipLaunchKernelGGL(dsb::col_dot_merge_kernel, dim3(ceil_div((int)NS, 256)), dim3(256), 0, st, part, g, c);
  hipLaunchKernelGGL(dsb::dsim_kernel, dim3((unsigned)((NL + 3) / 4)), dim3(256), 0, st, dsim, grad_conf, g, w.rowstat, w.colstat, r, c,
                     p->mask0, p->mask1);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ---- backward of the Sinkhorn head (sinkhorn_bwd.h) ---------------------------------------------------------------------------
// dZ [N, L+1, S+1] <- dL/d couplings (the padded score matrix) given grad_assign = dL/d conf_matrix_with_bin; *dbin <- dL/d bin_score.
// z_scratch [N, L, S]: the re-created scores.  The caller slices dZ[:, :L, :S] (zeroing mask-filled entries) and finishes with
// dL/dfeat_c0 = dsim feat_c1 / C, dL/dfeat_c1 = dsim^T feat_c0 / C.   Workspace: loftr_sinkhorn_bwd_workspace_bytes.
extern "C" size_t loftr_sinkhorn_bwd_workspace_bytes(int N, int L, int S, int C, int iters) {
  if (N <= 0 || L <= 0 || S <= 0 || C <= 0 || iters < 0) return 0;
  const size_t un = (size_t)N * (L + 1), vn = (size_t)N * (S + 1);
  return match_ws_bytes(N, L, S, C) + align_up(un * 4 * (iters > 0 ? iters : 1), 256) + align_up(vn * 4 * (iters + 1), 256) +
         align_up(un * 4, 256) + align_up(vn * 4, 256) + align_up(vn * 4 * otb::RCH, 256) + 4096;
}

extern "C" int loftr_sinkhorn_bwd(const float* feat_c0, const float* feat_c1, const loftr_coarse_params* p, float bin_score, int iters,
                                  const float* grad_assign, float* z_scratch, float* dZ, float* dbin, void* ws, size_t ws_bytes,
                                  void* stream) {
  LOFTR_CHECK_ARG(feat_c0 && feat_c1 && p && grad_assign && z_scratch && dZ && dbin && iters >= 0);
  LOFTR_CHECK_ARG(p->N >= 0 && p->h0c > 0 && p->w0c > 0 && p->h1c > 0 && p->w1c > 0 && (p->mask0 == nullptr) == (p->mask1 == nullptr));
  if (p->C % 32 != 0) return LOFTR_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (p->N == 0) { (void)hipMemsetAsync(dbin, 0, sizeof(float), st); return LOFTR_OK; }
  LOFTR_CHECK_ARG(ws != nullptr);
  const Geometry g = make_geometry(*p);
  const size_t base = match_ws_bytes(g.N, g.L, g.S, g.C);
  if (ws_bytes < base) return LOFTR_ERR_WORKSPACE;
  MatchWs w = carve(ws, base, g);
  if (!w.ok) return LOFTR_ERR_WORKSPACE;
  WsAlloc wa(static_cast<char*>(ws) + base, ws_bytes - base);
  const size_t un = (size_t)g.N * (g.L + 1), vn = (size_t)g.N * (g.S + 1);
  float* save_u = wa.take<float>(un * (iters > 0 ? iters : 1));
  float* save_v = wa.take<float>(vn * (iters + 1));
  float* du = wa.take<float>(un);
  float* dv = wa.take<float>(vn);
  float* part = wa.take<float>(vn * otb::RCH);
  if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
  { int rc = stage_descriptors(feat_c0, feat_c1, g, w, st); if (rc) return rc; }
  const float scale = 1.f / (float)g.C, norm = -logf((float)(g.L + g.S));
  if (g.C == 256) {
    sweep::Args a{};
    sweep_plan(g, a);
    a.f0 = w.f0sp; a.f1 = w.f1sp; a.scale = scale; a.mask0 = p->mask0; a.mask1 = p->mask1; a.conf = z_scratch;
    const dim3 swgrid(NUM_XCD * ceil_div(g.N * a.NCH, NUM_XCD) * a.RB);
    if (p->mask0) hipLaunchKernelGGL((sweep::score_sweep_kernel<2, true>), swgrid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((sweep::score_sweep_kernel<2, false>), swgrid, dim3(512), 0, st, a);
  } else {
    hipLaunchKernelGGL(score_store_kernel, dim3(score_grid(g)), dim3(Cfg::THREADS), 0, st, w.f0sp, w.f1sp, g, scale, p->mask0, p->mask1, z_scratch);
  }
  if (iters == 0) (void)hipMemsetAsync(save_u, 0, un * sizeof(float), st);
  ot_iterate(g, ot_plan(g), z_scratch, w, bin_score, norm, iters, st, save_u, save_v);
  const otb::Pad pad{z_scratch, bin_score, norm, logf((float)g.S) + norm, logf((float)g.L) + norm, g.N, g.L, g.S};
  const dim3 rgrid((unsigned)((un + 3) / 4)), cgrid(ceil_div(g.S + 1, 256), otb::RCH, g.N), mgrid(ceil_div((int)vn, 256));
  const float* uT = save_u + (iters > 0 ? (size_t)(iters - 1) * un : 0);
  const float* vT = save_v + (size_t)iters * vn;
  hipLaunchKernelGGL((otb::row_step_kernel<0>), rgrid, dim3(256), 0, st, pad, grad_assign, dZ, uT, vT, nullptr, du, 0);
  hipLaunchKernelGGL((otb::col_step_kernel<0>), cgrid, dim3(256), 0, st, pad, dZ, nullptr, nullptr, nullptr, part);
  hipLaunchKernelGGL(otb::col_merge_kernel, mgrid, dim3(256), 0, st, part, g.N, g.S + 1, dv);
  for (int t = iters; t >= 1; --t) {
    const float* ut = save_u + (size_t)(t - 1) * un;
    hipLaunchKernelGGL((otb::row_step_kernel<1>), rgrid, dim3(256), 0, st, pad, nullptr, dZ, ut, save_v + (size_t)t * vn, dv, du, t == iters ? 1 : 0);
    hipLaunchKernelGGL((otb::col_step_kernel<1>), cgrid, dim3(256), 0, st, pad, dZ, ut, save_v + (size_t)(t - 1) * vn, du, part);
    hipLaunchKernelGGL(otb::col_merge_kernel, mgrid, dim3(256), 0, st, part, g.N, g.S + 1, dv);
  }
  hipLaunchKernelGGL(otb::dbin_kernel, dim3(1), dim3(1024), 0, st, dZ, g.N, g.L, g.S, dbin);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
