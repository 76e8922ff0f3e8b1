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

// ==========================================================================================
// Score-volume kernels, second generation (C == 256): a STATIONARY-OPERAND SWEEP instead of a tiled GEMM.
//
// A wave keeps the fp16 (hi, lo) MFMA fragments of 32 descriptors of image 0 for the whole K = 256 in
// registers (128 VGPRs) and sweeps them over 32-column panels of image-1 descriptors that the workgroup's
// EIGHT waves (256 rows, one workgroup per CU) share through LDS: 32 KB per panel, a four-stage ring filled by
// global_load_lds two panels ahead.  Per panel and wave: 48 MFMAs from 32 ds_read_b128 -- no B-operand staging,
// ONE s_barrier per panel (per 48 MFMAs; the tiled loop needs one per 24) and 85 B of DMA per MFMA (341 B
// there): the global -> LDS path (~6.5 TB/s chip-wide) was what bounded a 4-wave version of this kernel.
// The MFMA is issued with the image-1 panel as the A (row) operand and the image-0 fragments as the B (column)
// operand, so in the accumulator layout a LANE owns one row i of the score matrix (lane & 31) and its 16
// registers are 16 columns j = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of the panel:
//   * row statistics (pass A: online max / sum exp; pass B: running max + first argmax of conf) are lane-private
//     running values over the whole sweep -- no cross-lane reduction per tile, one half-wave exchange at the end;
//   * each lane holds 4 consecutive columns per register quad -> conf_matrix leaves as 16-byte stores;
//   * column statistics are a 32-lane DPP reduction per panel and wave, written as per-wave partials.
// The two waves that share a SIMD (w and w + 4) run HALF A PERIOD APART: waves 0-3 do {MFMAs of panel p, epilogue
// of panel p} between two barriers, waves 4-7 do {epilogue of panel p-1, MFMAs of panel p}, so one wave's VALU-only
// epilogue (exp, DPP reductions, stores) always runs under its partner's MFMAs instead of next to its epilogue.
// No LDS store and no VGPR-destination global load is issued inside the panel loop: either makes hipcc wait
// vmcnt(0) and would drain the DMA ring every iteration.
// Work unit = (pair, 256-row block, chunk of panels); the chunking depends on S only, so a pair's results do not
// depend on the batch it is in.  Row partials: one per (row, chunk); column partials: one per (column, 32-row wave).
#ifndef SWEEP_PROBE_EPI
#define SWEEP_PROBE_EPI 1        // 0: skip the epilogues (timing probe only; wrong results)
#endif
#ifndef SWEEP_PROBE_LSE
#define SWEEP_PROBE_LSE 1        // 0: pass B on (max, 1/sum) statistics instead of log-sum-exp biases
#endif
#ifndef SWEEP_PROBE_FAST
#define SWEEP_PROBE_FAST 1       // 0: pass A exact variant only
#endif
#ifndef SWEEP_PROBE_ACC1
#define SWEEP_PROBE_ACC1 0       // 1: ONE accumulator chain (no acc0 + acc1 in the epilogue): pass A -2.5 %, pass B +1.5 %, net 0.
                                 // MUST be the same in both passes: conf = exp2(2 v - LSE) near 1 relies on pass B reproducing
                                 // pass A's v bit for bit (a different summation order costs 1e-4 at logits of a few hundred)
#endif
#ifndef SWEEP_PROBE_PRIO
#define SWEEP_PROBE_PRIO 0       // 1: s_setprio 1 around the MFMA block (probe)
#endif
#ifndef SWEEP_PIPE
#define SWEEP_PIPE 0             // 0: compiler-visible ds_reads in the panel loop, order and waits left to hipcc (A/B)
#endif
#ifndef PIPE_DEP
#define PIPE_DEP 1               // phases of fragment prefetch
#endif
#define PIPE_NBUF (PIPE_DEP + 1)
#ifndef SWEEP_PROBE_NOSTORE
#define SWEEP_PROBE_NOSTORE 0    // 1: pass B without the conf_matrix stores (timing probe)
#endif
#ifndef SWEEP_PROBE_COAL
#define SWEEP_PROBE_COAL 0       // 1: pass B stores in the pattern of a lane = column layout: 16 x 4 B, 128 B contiguous per half-wave
#endif
#ifndef SWEEP_PROBE_NODMA
#define SWEEP_PROBE_NODMA 0      // 1: no LDS-DMA (timing probes only; wrong results)
#endif
#ifndef SWEEP_PROBE_NOLDS
#define SWEEP_PROBE_NOLDS 0      // 1: no panel fragment reads
#endif
#ifndef SWEEP_PROBE_NOBAR
#define SWEEP_PROBE_NOBAR 0      // 1: no per-panel barrier
#endif
#ifndef SWEEP_PROBE_SKEW
#define SWEEP_PROBE_SKEW 1       // 0: all eight waves in phase
#endif
namespace sweep {
constexpr int W = 8, BR = 32 * W, PC = 32, KS = 16, STAGE = PC * 1024, NST = 4, MAXP = 32;
constexpr int OFF_CSTAT = NST * STAGE;                    // float2 [MAXP * PC] column (max, 1/sum) of the chunk (pass B)
constexpr int OFF_MASK = OFF_CSTAT + MAXP * PC * 8;       // uint8  [MAXP * PC] mask1 of the chunk
constexpr int LDS_BYTES = OFF_MASK + MAXP * PC;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
constexpr int DMA_PER_WAVE = PC * 8 / 8 / W;              // global_load_lds instructions per wave per panel (4)
constexpr int TIE_BIT = 1 << 30;                          // set in the argmax word of a row partial: the maximum is attained twice

struct Args {
  const sp_t* f0; const sp_t* f1;
  int N, L, S;
  int RB, NCH, PPC, NP;             // row blocks, column chunks, panels per chunk, panels in total
  float scale;
  const uint8_t* mask0; const uint8_t* mask1;
  float2* rowpart; float2* colpart;                 // pass A out: [N][NCH][L], [N][RB * W][S]
  const float2* rowstat; const float2* colstat;     // pass B in
  float* conf;                                      // pass B out or null
  float2* rowmax_part; float* colmax_part;          // pass B out: [N][NCH][L] (max, argmax | TIE_BIT), [N][RB * W][S]
  int* exact_flags;                                 // pass A: [N * NCH * RB] units the exact variant has to (re)do, or null = all
};

__device__ __forceinline__ int jr(int r, int g) { return 8 * (r >> 2) + 4 * g + (r & 3); }
// the value of register (lane & 15) of a 16-register vector: lane l of a half-wave then owns column jr(l & 15, g)
__device__ __forceinline__ float pick16(const f32x16& v, int sel) {
  float x = v[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) x = sel == r ? v[r] : x;
  return x;
}

// ---- transposed reductions over the 32 lanes of a half-wave --------------------------------------------------
// A plain reduction of 16 registers across 32 lanes costs 16 x 5 cross-lane steps and leaves every lane with all 16
// results.  Here every step HALVES the registers a lane carries (it keeps the half selected by one of its lane bits and
// hands the other half to its partner, who keeps exactly that one): 8 + 4 + 2 + 1 + 1 steps, and lane l ends up with
// the full reduction of register tr_reg(l) only -- which is all the column partials need (one lane stores one column).
//   step partners: l ^ 7 (row_half_mirror), l ^ 1, l ^ 2 (quad_perm), l ^ 8 (row_ror:8), l ^ 16 (v_permlane16_swap);
//   register kept by lane l:  8 * bit2(l) + 4 * bit0(l) + 2 * bit1(l) + bit3(l).
__device__ __forceinline__ int tr_reg(int l) { return 8 * ((l >> 2) & 1) + 4 * (l & 1) + 2 * ((l >> 1) & 1) + ((l >> 3) & 1); }
#define SWEEP_DPPF(v_, c_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), c_, 0xF, 0xF, true))
#define SWEEP_DPPI(v_, c_) __builtin_amdgcn_update_dpp(0, v_, c_, 0xF, 0xF, true)
// lane-dependent choice between two registers as ONE v_bfi_b32 on a precomputed all-ones / all-zeros lane mask
// (a bool select costs a v_cmp + hazard nops + v_cndmask each time: hipcc re-materialises the comparison)
__device__ __forceinline__ int bsel(int m, int a1, int a0) { return (a1 & m) | (a0 & ~m); }      // m ? a1 : a0
__device__ __forceinline__ float bself(int m, float a1, float a0) { return __int_as_float(bsel(m, __float_as_int(a1), __float_as_int(a0))); }
struct TrMasks { int m0, m1, m2, m3; };            // lane bit k set -> all ones
__device__ __forceinline__ TrMasks tr_masks(int lane) { return TrMasks{-(lane & 1), -((lane >> 1) & 1), -((lane >> 2) & 1), -((lane >> 3) & 1)}; }
__device__ __forceinline__ float treduce_add16(const f32x16& v, const TrMasks& t) {
  float x8[8], x4[4], x2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) x8[k] = bself(t.m2, v[8 + k], v[k]) + SWEEP_DPPF(bself(t.m2, v[k], v[8 + k]), 0x141);
#pragma unroll
  for (int k = 0; k < 4; ++k) x4[k] = bself(t.m0, x8[4 + k], x8[k]) + SWEEP_DPPF(bself(t.m0, x8[k], x8[4 + k]), 0xB1);
#pragma unroll
  for (int k = 0; k < 2; ++k) x2[k] = bself(t.m1, x4[2 + k], x4[k]) + SWEEP_DPPF(bself(t.m1, x4[k], x4[2 + k]), 0x4E);
  const float x1 = bself(t.m3, x2[1], x2[0]) + SWEEP_DPPF(bself(t.m3, x2[0], x2[1]), 0x128);
  const auto q = __builtin_amdgcn_permlane16_swap(__float_as_int(x1), __float_as_int(x1), false, false);
  return __int_as_float(q[0]) + __int_as_float(q[1]);
}
// the same with a signed-integer maximum: on the bit patterns of NON-NEGATIVE floats it is the float maximum (and any
// negative float, the "invalid" marker -1, loses) -- without the canonicalisation fmaxf costs on cross-lane values
__device__ __forceinline__ int treduce_imax16(const int (&v)[16], const TrMasks& t) {
  int x8[8], x4[4], x2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) x8[k] = max(bsel(t.m2, v[8 + k], v[k]), SWEEP_DPPI(bsel(t.m2, v[k], v[8 + k]), 0x141));
#pragma unroll
  for (int k = 0; k < 4; ++k) x4[k] = max(bsel(t.m0, x8[4 + k], x8[k]), SWEEP_DPPI(bsel(t.m0, x8[k], x8[4 + k]), 0xB1));
#pragma unroll
  for (int k = 0; k < 2; ++k) x2[k] = max(bsel(t.m1, x4[2 + k], x4[k]), SWEEP_DPPI(bsel(t.m1, x4[k], x4[2 + k]), 0x4E));
  const int x1 = max(bsel(t.m3, x2[1], x2[0]), SWEEP_DPPI(bsel(t.m3, x2[0], x2[1]), 0x128));
  const auto q = __builtin_amdgcn_permlane16_swap(x1, x1, false, false);
  return max((int)q[0], (int)q[1]);
}
constexpr float LOG2E = 1.4426950408889634f;
constexpr float FAST_SPREAD = 64.f;    // max - min of a wave's 32 x 32 score tile up to which one shared exp reference is exact enough

// FASTA (pass A only): the lean epilogue with ONE shared exp reference per 32 x 32 wave tile.  It is exact only while
// the tile's values span less than FAST_SPREAD and the tile is full, so this variant gives up on a unit the moment a
// tile fails the test (or at once for units with a partial last panel): it raises the unit's flag and the
// exact variant -- launched right behind it on the same grid, returning immediately for unflagged units -- redoes
// that unit with per-row / per-column references.  Two kernels instead of one two-path kernel: together the paths
// exceed the 256-VGPR budget of a 512-thread workgroup (30 spills measured).
// TRACKJ (pass B): per-ELEMENT tracking of the first row arg-max (5 VALU per element).  Without it the sweep only tracks,
// per lane, the maximum and the PANEL it first occurred in (11 VALU per panel) and select_kernel finds the column --
// and any second occurrence -- by reading those 32 entries of conf_matrix back; that needs the materialised matrix.
// The sweep is issue-bound (about five non-MFMA instructions fit under one 32-cycle MFMA), so this matters.
template <int PASS, bool HAS_MASK, bool FASTA = false, bool TRACKJ = true>
__global__ __launch_bounds__(512, 2) void score_sweep_kernel(Args a) {
  static_assert(!(FASTA && (PASS != 0 || HAS_MASK)), "the shared-reference path is pass A without masks");
  static_assert(TRACKJ || (PASS == 1 && !HAS_MASK), "panel-level tracking is the unmasked pass B");
  constexpr bool LSE = SWEEP_PROBE_LSE && PASS == 1 && !HAS_MASK;      // pass B on log-sum-exp biases (one exp2(fma) per element)
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  // ---- unit: groups (chunk, pair) are dealt to the XCDs; the row blocks of a group run back to back on it
  const int id = blockIdx.x, xcd = id % NUM_XCD, slot = id / NUM_XCD;
  const int grp = (slot / a.RB) * NUM_XCD + xcd, rb = slot % a.RB;
  if (grp >= a.N * a.NCH) return;
  const int n = grp % a.N, cc = grp / a.N;
  const int p0 = cc * a.PPC, np = min(a.PPC, a.NP - p0);      // panels of this chunk (>= 1 by construction)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  const bool late = SWEEP_PROBE_SKEW && wave >= W / 2;         // this wave runs its epilogue half a period later
  const int L = a.L, S = a.S;
  const int row = rb * BR + wave * 32 + li;
  const bool row_ok = row < L;
  const bool rows_full = FASTA || rb * BR + BR <= L;           // block-uniform (the fast variant only keeps full units)
  int* const unit_flag = a.exact_flags ? a.exact_flags + (grp * a.RB + rb) : nullptr;
  const bool rows_part = rb * BR + BR > L;                     // block-uniform: the last row block of the pair is partial
  if (PASS == 0 && FASTA) {                                    // partial last panel: the exact variant's job
    if ((p0 + np) * PC > S) { if (threadIdx.x == 0) *unit_flag = 1; return; }
  }
  if (PASS == 0 && !FASTA && unit_flag && *unit_flag == 0) return;   // the fast variant has done this unit
  const sp_t* f0n = a.f0 + (long)n * L * 256;
  const sp_t* f1n = a.f1 + (long)n * S * 256;

  // ---- stationary operand: this lane's 16-byte MFMA fragments of its row, all 16 k-steps, hi and lo
  h16x8 bh[KS], bl[KS];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(f0n + (long)min(row, L - 1) * 256);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = (ks >> 1) * 8 + 2 * (ks & 1) + g;
      bh[ks] = __builtin_bit_cast(h16x8, src[c]);
      bl[ks] = __builtin_bit_cast(h16x8, src[c + 4]);
    }
  }
  // ---- chunk tables -> LDS (ordinary loads and LDS stores happen only here, before any DMA is in flight)
  float2* cstat_s = reinterpret_cast<float2*>(lds + OFF_CSTAT);
  uint8_t* mask_s = reinterpret_cast<uint8_t*>(lds + OFF_MASK);
  for (int t = threadIdx.x; t < np * PC; t += 512) {
    const int col = min(p0 * PC + t, S - 1);
    if (PASS == 1) {
      const float2 cs = a.colstat[(long)n * S + col];
      // LSE form: -log2 sum_i exp(v_ij) = -(max * log2 e) + log2(1 / sum): independent of which reference "max" was
      if (LSE) reinterpret_cast<float*>(cstat_s)[t] = fmaf(-cs.x, LOG2E, __builtin_amdgcn_logf(cs.y));
      else cstat_s[t] = cs;
    }
    if (HAS_MASK) mask_s[t] = a.mask1[(long)n * S + col];
  }
  float rm = 0.f, rs = 0.f;                        // pass B: row (max, 1/sum); LSE form: rm = -log2 sum_j exp(v_ij)
  if (PASS == 1) {
    const float2 t = a.rowstat[(long)n * L + min(row, L - 1)];
    rm = LSE ? fmaf(-t.x, LOG2E, __builtin_amdgcn_logf(t.y)) : t.x; rs = t.y;
  }
  const float k2 = 2.f * a.scale * LOG2E;          // LSE form: conf = exp2(k2 * dot + rm + cb_j)
  const bool mrow = HAS_MASK ? (a.mask0[(long)n * L + min(row, L - 1)] != 0) : true;

  // ---- DMA of one panel: 32 rows x 8 k-groups x 128 B = 32 instructions, 4 per wave (k-group = wave)
  // dword offset of this lane's 16 B inside a panel row set, per row octet: recomputed per issue (a few VALU per panel)
  // rather than held in registers across the loop
#define SWEEP_DOFF(oct_) ((oct_) * 8 * 256 + (lane >> 3) * 256 + (((lane & 7) ^ (((oct_) * 4 + (lane >> 4)) & 7)) << 2) + wave * 32)
#define SWEEP_ISSUE(p_)                                                                                  \
  {                                                                                                      \
    const int col0__ = (p0 + (p_)) * PC;                                                                 \
    char* st__ = lds + ((p_) & (NST - 1)) * STAGE + wave * 4096;                                         \
    if (col0__ + PC <= S) {                                                                              \
      const sp_t* base__ = f1n + (long)col0__ * 256;                                                     \
      _Pragma("unroll") for (int oct__ = 0; oct__ < 4; ++oct__)                                          \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base__ + SWEEP_DOFF(oct__)), (lds_ptr_t)(st__ + oct__ * 1024), 16, 0, 0); \
    } else {                  /* last panel of the matrix: rows beyond S re-read row S-1 (masked later) */ \
      _Pragma("unroll") for (int oct__ = 0; oct__ < 4; ++oct__) {                                        \
        const int r__ = oct__ * 8 + (lane >> 3);                                                         \
        const int gc__ = min(col0__ + r__, S - 1);                                                       \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(f1n + (long)gc__ * 256 + (SWEEP_DOFF(oct__) - r__ * 256)), \
                                         (lds_ptr_t)(st__ + oct__ * 1024), 16, 0, 0);                    \
      }                                                                                                  \
    }                                                                                                    \
  }

  // ---- running row state (lane private)
  // pass A: sum of exp(v - ref_run) over the columns seen so far; ref_run >= every v seen is a running REFERENCE, not
  // necessarily the maximum -- the merge kernels and pass B only ever use max-reference + log(sum) combinations
  float ref_run = SENTINEL, s_run = 0.f;
  float best = -1.f; int bestj = 0; bool tie = false;   // pass B
  const TrMasks trm = tr_masks(lane);
  const int trcol = jr(tr_reg(lane & 15), g);       // panel column whose transposed reduction ends in this lane
  [[maybe_unused]] const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;     // SWEEP_PIPE: LDS address of the panel ring
  const int a_off = lds_chunk_off(li, g);          // hi chunk of the even k-step; odd k-step: ^ 32, lo: ^ 64 (chunk + 2 / + 4)
  const int sel = lane & 15;
  const long part_row = ((long)n * a.RB * W + rb * W + wave) * S;          // this wave's row of the column partials

  // Epilogue of panel p_ on the accumulators: statistics / conf_matrix.  A macro, not a lambda (captures of the register
  // arrays by reference end up in scratch); expanded twice (early and late waves).
#define SWEEP_ACC(r_) (SWEEP_PROBE_ACC1 ? acc0[r_] : acc0[r_] + acc1[r_])
#define SWEEP_EPILOGUE(p_)                                                                               \
  {                                                                                                      \
    const int col0 = (p0 + (p_)) * PC;                                                                   \
    const bool fullp = FASTA || col0 + PC <= S;    /* panel-uniform */                                   \
    f32x16 v;                                                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) v[r] = SWEEP_ACC(r) * a.scale;                        \
    if (HAS_MASK) {                                                                                      \
      _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                     \
        if (!(mrow && mask_s[(p_) * PC + jr(r, g)])) v[r] = LOFTR_NEG_INF;   /* masked_fill_(~(m0 x m1), -INF)  :115-118 */ \
    }                                                                                                    \
    if (!fullp) {                                                                                        \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) if (col0 + jr(r, g) >= S) v[r] = SENTINEL;          \
    }                                                                                                    \
    const int mycol = col0 + jr(sel, g);           /* the column this lane stores a partial for */       \
    if (PASS == 2) {                               /* Sinkhorn: the scaled, mask-filled score itself (coarse_matching.py:123-126) */ \
      if (rows_full || row_ok) {                                                                         \
        float* co = a.conf + ((long)n * L + row) * S + col0 + 4 * g;                                     \
        if (fullp && (S & 3) == 0) {                                                                     \
          _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
            *reinterpret_cast<f32x4*>(co + 8 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; \
        } else {                                                                                         \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) if (col0 + jr(r, g) < S) co[8 * (r >> 2) + (r & 3)] = v[r]; \
        }                                                                                                \
      }                                                                                                  \
    } else if (PASS == 0) {                                                                                     \
      float tm = FASTA ? SWEEP_ACC(0) : v[0], tn = tm;      /* FASTA: extrema of the RAW dot products (scale > 0) */ \
      _Pragma("unroll") for (int r = 1; r < 16; r += 2) {                                                \
        const float x0 = FASTA ? SWEEP_ACC(r) : v[r], x1 = r + 1 < 16 ? (FASTA ? SWEEP_ACC(r + 1) : v[r + 1]) : x0; \
        tm = fmaxf(fmaxf(tm, x0), x1); tn = fminf(fminf(tn, x0), x1);            /* v_max3 / v_min3 */  \
      }                                                                                                  \
      if (FASTA) {                                                                                       \
        float R = half_max(tm); R = fmaxf(R, swap32(R));           /* maximum / minimum of the wave's 32 x 32 tile */ \
        float mnw = -half_max(-tn); mnw = fminf(mnw, swap32(mnw));                                       \
        R *= a.scale; mnw *= a.scale;                                                                    \
        if (!(R - mnw <= FAST_SPREAD) && lane == 0) *unit_flag = 1;      /* (also catches NaN) -> redone exactly */ \
        /* ONE exponential per element, relative to the tile maximum R, serves the row AND the column sums: every   \
           element is within FAST_SPREAD of R, so nothing that matters to any row or column underflows.  Packed     \
           fp32 arithmetic (v_pk_fma_f32 / v_pk_add_f32: two elements per instruction) on the raw accumulators,     \
           exponent = dot * (scale log2 e) - R log2 e with the same scale log2 e = k2 / 2 pass B uses */            \
        const f32x2 sl2 = {0.5f * k2, 0.5f * k2}, nrk2 = {-R * LOG2E, -R * LOG2E};                       \
        f32x16 e;                                                                                        \
        f32x2 ss2 = {0.f, 0.f};                                                                          \
        _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                              \
          const f32x2 x2 = __builtin_elementwise_fma(f32x2{SWEEP_ACC(r), SWEEP_ACC(r + 1)}, sl2, nrk2);  \
          e[r] = __builtin_amdgcn_exp2f(x2.x); e[r + 1] = __builtin_amdgcn_exp2f(x2.y);                  \
          ss2 += f32x2{e[r], e[r + 1]};                                                                  \
        }                                                                                                \
        const float ssum = ss2.x + ss2.y;                                                                \
        const float Rn = fmaxf(ref_run, R);                                                              \
        s_run = s_run * fexp(ref_run - Rn) + ssum * fexp(R - Rn);                                        \
        ref_run = Rn;                                                                                    \
        /* columns: sum of e over the wave's 32 rows, one column per lane, as (reference, sum); rows beyond L are      \
           clamped copies of row L-1 (harmless for the tile extrema) and drop out here */                \
        if (rows_part) { _Pragma("unroll") for (int r = 0; r < 16; ++r) e[r] = row_ok ? e[r] : 0.f; }    \
        const float csum = treduce_add16(e, trm);                                                        \
        if (li < 16) a.colpart[part_row + col0 + trcol] = make_float2(R, csum);                          \
      } else {                                                                                           \
        /* exact path: per-row reference = the running row maximum, per-column reference = the column maximum */    \
        const float mn = fmaxf(ref_run, tm);                                                             \
        float ssum = 0.f;                                                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) ssum += fexp(v[r] - mn);    /* exp(SENTINEL - x) == 0 */ \
        s_run = s_run * fexp(ref_run - mn) + ssum;                                                       \
        ref_run = mn;                                                                                    \
        f32x16 cm;                                                                                       \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) cm[r] = (rows_full || row_ok) ? v[r] : SENTINEL;  \
        f32x16 e = cm;                                                                                   \
        half_max16(cm);                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) e[r] = fexp(e[r] - cm[r]);                        \
        half_sum16(e);                                                                                   \
        const float2 mine = make_float2(pick16(cm, sel), pick16(e, sel));                                \
        if (li < 16 && (fullp || mycol < S)) a.colpart[part_row + mycol] = mine;                         \
      }                                                                                                  \
    } else {                                                                                             \
      /* conf = softmax(sim, dim=1) * softmax(sim, dim=2) = exp((v - rowmax) + (v - colmax)) / (rowsum * colsum)   :119 */ \
      f32x16 c;                                                                                          \
      if (LSE) {                                                                                         \
        /* = exp2(2 v log2e - LSE_row - LSE_col): the (acc0 + acc1) * scale above folds into the fma */  \
        const f32x2 k22 = {k2, k2}, rm2 = {rm, rm};                                                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                  \
          const f32x4 cb = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(cstat_s) + (p_) * PC + 8 * q + 4 * g); \
          _Pragma("unroll") for (int e = 0; e < 4; e += 2) {            /* two elements per v_pk_add / v_pk_fma */ \
            const f32x2 x2 = __builtin_elementwise_fma(f32x2{SWEEP_ACC(4 * q + e), SWEEP_ACC(4 * q + e + 1)}, k22, \
                                                       rm2 + f32x2{cb[e], cb[e + 1]});                   \
            c[4 * q + e] = __builtin_amdgcn_exp2f(x2.x); c[4 * q + e + 1] = __builtin_amdgcn_exp2f(x2.y); \
          }                                                                                              \
        }                                                                                                \
      } else {                                                                                           \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                    \
        const f32x4* cs4 = reinterpret_cast<const f32x4*>(cstat_s + (p_) * PC + 8 * q + 4 * g);   /* (max, 1/sum) x 4 columns */ \
        const f32x4 c01 = cs4[0], c23 = cs4[1];                                                          \
        const float cmx[4] = {c01.x, c01.z, c23.x, c23.z}, cis[4] = {c01.y, c01.w, c23.y, c23.w};        \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
          const float x = v[4 * q + e];                                                                  \
          c[4 * q + e] = fexp((x - rm) + (x - cmx[e])) * (rs * cis[e]);                                  \
        }                                                                                                \
      }                                                                                                  \
      }                                                                                                  \
      if (!fullp) {                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) if (col0 + jr(r, g) >= S) c[r] = -1.f;            \
      }                                                                                                  \
      if (a.conf && (rows_full || row_ok)) {                                                             \
        float* co = a.conf + ((long)n * L + row) * S + col0 + 4 * g;                                     \
        if (SWEEP_PROBE_NOSTORE) {                                                                       \
        } else if (SWEEP_PROBE_COAL) {      /* timing probe: the store pattern of a lane = column layout (wrong data) */ \
          float* cq = a.conf + ((long)n * L + rb * BR + wave * 32) * S + col0 + li;                      \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) cq[(long)jr(r, g) * S] = c[r];                  \
        } else if (fullp && (S & 3) == 0) {                                                              \
          _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
            *reinterpret_cast<f32x4*>(co + 8 * q) = f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]}; \
        } else {                                                                                         \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) if (col0 + jr(r, g) < S) co[8 * (r >> 2) + (r & 3)] = c[r]; \
        }                                                                                                \
      }                                                                                                  \
      if (TRACKJ) {                                                                                      \
        /* row: running (max, FIRST argmax, attained-twice flag); registers ascend in column order for this half */ \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
          const bool gt = c[r] > best;                                                                   \
          tie = gt ? false : (tie || c[r] == best);                                                      \
          bestj = gt ? col0 + jr(r, g) : bestj;                                                          \
          best = gt ? c[r] : best;                                                                       \
        }                                                                                                \
      } else {                                                                                           \
        /* row: running maximum and the FIRST panel that attains it (select_kernel finds the column) */  \
        float pm = c[0];                                                                                 \
        _Pragma("unroll") for (int r = 1; r < 16; r += 2) pm = fmaxf(fmaxf(pm, c[r]), r + 1 < 16 ? c[r + 1] : c[r]);   /* v_max3 */ \
        const bool gt = pm > best;                                                                       \
        tie = gt ? false : (tie || pm == best);                                                          \
        bestj = gt ? (p0 + (p_)) : bestj;                                                                \
        best = gt ? pm : best;                                                                           \
      }                                                                                                  \
      /* columns: max over the wave's rows, one column per lane (conf >= 0: integer maximum of the bit patterns) */ \
      int cb[16];                                                                                        \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) cb[r] = __float_as_int((rows_full || row_ok) ? c[r] : -1.f); \
      const float cmine = __int_as_float(treduce_imax16(cb, trm));                            \
      if (li < 16 && (fullp || col0 + trcol < S)) a.colmax_part[part_row + col0 + trcol] = cmine;        \
    }                                                                                                    \
  }

  // Every ordinary load above must be COMPLETE before the first DMA is issued: hipcc waits vmcnt(0) at the first use
  // of a VGPR-destination load, and a first use inside the panel loop would drain the in-flight DMA every iteration.
  LOFTR_WAITCNT_VM(0);
  __syncthreads();                                 // tables visible; no DMA in flight yet
  SWEEP_ISSUE(0);
  if (np > 1) SWEEP_ISSUE(1);
  // Stores a wave issues between two DMA issues (they sit between the DMA of panel p+1 and the barrier of panel p+1
  // in the in-order VMEM queue): pass A one partial store; pass B four conf stores + one partial store.  Panels that
  // take the scalar-store tail path are followed by a full drain instead.
  constexpr int ST = PASS == 0 ? 1 : PASS == 2 ? 4 : (SWEEP_PROBE_NOSTORE ? 1 : SWEEP_PROBE_COAL ? 17 : 5);
  f32x16 acc0, acc1;
  bool drain = false;                              // block-uniform: the previous period issued an unknown number of stores
  for (int p = 0; p < np; ++p) {
    // panel p has landed once at most {DMA of panel p+1, the epilogue stores issued after it} are outstanding (VMEM
    // operations retire in order).  The late waves have not stored anything before period 2, so the count only
    // includes the stores from there on (conservative for the early waves at p = 1).
    if (drain || p + 1 >= np) LOFTR_WAITCNT_VM(0);
    else if (p < 2) LOFTR_WAITCNT_VM(DMA_PER_WAVE);
    else LOFTR_WAITCNT_VM(DMA_PER_WAVE + ST);
    if (!SWEEP_PROBE_NOBAR) __builtin_amdgcn_s_barrier();   // ... for every wave; and every wave is past the MFMAs of panel p-2
    if (p + 2 < np && !SWEEP_PROBE_NODMA) SWEEP_ISSUE(p + 2);
    drain = !((p0 + p) * PC + PC <= S && (S & 3) == 0) && PASS >= 1;
    const char* st = lds + (p & (NST - 1)) * STAGE;
#if SWEEP_PIPE
    // ---- 48 MFMAs in eight phases of two k-steps (one 4 KB k-group of the panel: hi / lo fragments of an even and an
    // odd k-step = four ds_read_b128), the fragments of phase ph + PIPE_DEP in flight while phase ph multiplies.
    // The reads are inline asm with COUNTED lgkmcnt waits: while LDS-DMA is in flight hipcc's wait insertion degrades
    // every LDS dependency to lgkmcnt(0) (the DMA counts as a pending flat access), so with compiler-visible reads a
    // wave exposes the full LDS latency once per ds_read group and cannot keep the matrix pipe busy on its own --
    // which is what the half-period skew of the two waves of a SIMD relies on.  LDS reads return in order, so
    // "phase ph has landed" is lgkmcnt(4 PIPE_DEP) right after the reads of phase ph + PIPE_DEP were issued; the wait
    // statement names the fragments it releases ("+v"), which is what keeps their MFMAs below it.
    h16x8 fr[PIPE_NBUF][4];
    const unsigned stb = lds_base + (p & (NST - 1)) * STAGE;
    const unsigned ad0 = stb + a_off, ad1 = stb + (a_off ^ 64), ad2 = stb + (a_off ^ 32), ad3 = stb + (a_off ^ 96);
#define SWEEP_LOADPH(ph_)                                                                                \
    asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\t"                  \
                 "ds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8"                       \
                 : "=&v"(fr[(ph_) % PIPE_NBUF][0]), "=&v"(fr[(ph_) % PIPE_NBUF][1]), "=&v"(fr[(ph_) % PIPE_NBUF][2]),  \
                   "=&v"(fr[(ph_) % PIPE_NBUF][3])                                                       \
                 : "v"(ad0), "v"(ad1), "v"(ad2), "v"(ad3), "i"((ph_) * 4096));
#define SWEEP_WAITPH(ph_, n_)                                                                            \
    asm volatile("s_waitcnt lgkmcnt(%4)"                                                                 \
                 : "+v"(fr[(ph_) % PIPE_NBUF][0]), "+v"(fr[(ph_) % PIPE_NBUF][1]), "+v"(fr[(ph_) % PIPE_NBUF][2]),    \
                   "+v"(fr[(ph_) % PIPE_NBUF][3])                                                        \
                 : "i"(n_));
#define SWEEP_PHASE(ph_)                                                                                 \
    if ((ph_) + PIPE_DEP < 8) SWEEP_LOADPH((ph_) + PIPE_DEP)                                             \
    SWEEP_WAITPH(ph_, 4 * ((ph_) + PIPE_DEP < 8 ? PIPE_DEP : 7 - (ph_)))                                 \
    {                                                                                                    \
      const h16x8 ah0 = fr[(ph_) % PIPE_NBUF][0], al0 = fr[(ph_) % PIPE_NBUF][1];                        \
      const h16x8 ah1 = fr[(ph_) % PIPE_NBUF][2], al1 = fr[(ph_) % PIPE_NBUF][3];                        \
      if (SWEEP_PROBE_ACC1) {                                                                            \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[2 * (ph_)], acc0, 0, 0, 0);                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[2 * (ph_)], acc0, 0, 0, 0);                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[2 * (ph_)], acc0, 0, 0, 0);                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh[2 * (ph_) + 1], acc0, 0, 0, 0);            \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl[2 * (ph_) + 1], acc0, 0, 0, 0);            \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh[2 * (ph_) + 1], acc0, 0, 0, 0);            \
      } else {                                                                                           \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[2 * (ph_)], acc0, 0, 0, 0);                \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[2 * (ph_)], acc1, 0, 0, 0);                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[2 * (ph_)], acc0, 0, 0, 0);                \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh[2 * (ph_) + 1], acc1, 0, 0, 0);            \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl[2 * (ph_) + 1], acc0, 0, 0, 0);            \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh[2 * (ph_) + 1], acc1, 0, 0, 0);            \
      }                                                                                                  \
    }                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);
#if SWEEP_PROBE_NOLDS
#undef SWEEP_LOADPH
#undef SWEEP_WAITPH
#define SWEEP_LOADPH(ph_)
#define SWEEP_WAITPH(ph_, n_)
#pragma unroll
    for (int b_ = 0; b_ < PIPE_NBUF; ++b_) { fr[b_][0] = bh[b_]; fr[b_][1] = bl[b_]; fr[b_][2] = bh[b_ + 4]; fr[b_][3] = bl[b_ + 4]; }
#endif
    SWEEP_LOADPH(0)
    if (PIPE_DEP > 1) SWEEP_LOADPH(1)
    if (SWEEP_PROBE_EPI && late && p > 0) SWEEP_EPILOGUE(p - 1);      // (its VALU work covers the latency of the first fragments)
    if (SWEEP_PROBE_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    SWEEP_PHASE(0) SWEEP_PHASE(1) SWEEP_PHASE(2) SWEEP_PHASE(3) SWEEP_PHASE(4) SWEEP_PHASE(5) SWEEP_PHASE(6) SWEEP_PHASE(7)
#undef SWEEP_LOADPH
#undef SWEEP_WAITPH
#undef SWEEP_PHASE
#else
    if (SWEEP_PROBE_EPI && late && p > 0) SWEEP_EPILOGUE(p - 1);
    // ---- 48 MFMAs: two accumulators alternate so that no MFMA depends on its predecessor
    if (SWEEP_PROBE_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const char* sk = st + (ks >> 1) * 4096;
      const h16x8 ah = *reinterpret_cast<const h16x8*>(sk + (a_off ^ ((ks & 1) ? 32 : 0)));
      const h16x8 al = *reinterpret_cast<const h16x8*>(sk + (a_off ^ ((ks & 1) ? 96 : 64)));
      if (SWEEP_PROBE_ACC1) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc0, 0, 0, 0);
      } else if (ks & 1) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc0, 0, 0, 0);
      }
    }
#endif
    if (SWEEP_PROBE_PRIO) __builtin_amdgcn_s_setprio(0);
    if (SWEEP_PROBE_EPI && !late) SWEEP_EPILOGUE(p);
    if (!SWEEP_PROBE_EPI) { s_run += acc0[0] + (SWEEP_PROBE_ACC1 ? 0.f : acc1[5]); best += acc0[3] + (SWEEP_PROBE_ACC1 ? 0.f : acc1[7]); }     // keep the MFMAs alive
  }
  if (SWEEP_PROBE_EPI && late) SWEEP_EPILOGUE(np - 1);
#undef SWEEP_ISSUE
#undef SWEEP_DOFF
#undef SWEEP_EPILOGUE
#undef SWEEP_ACC
#undef SWEEP_DPPF
#undef SWEEP_DPPI
  // ---- row partials of this chunk: combine the two half-waves (they hold disjoint columns of the same row)
  if (PASS == 2) return;
  float2* rp = (PASS == 0 ? a.rowpart : a.rowmax_part) + ((long)n * a.NCH + cc) * L;
  if (PASS == 0) {
    const float ro = swap32(ref_run), so = swap32(s_run);
    const float M = fmaxf(ref_run, ro);
    const float Ssum = s_run * fexp(ref_run - M) + so * fexp(ro - M);
    if (g == 0 && row_ok) rp[row] = make_float2(M, Ssum);
  } else {
    const float bo = swap32(best);
    const int jo = __float_as_int(swap32(__int_as_float(bestj)));
    const bool to = swap32(tie ? 1.f : 0.f) != 0.f;
    const bool other = bo > best || (bo == best && jo < bestj);
    const bool t = (bo == best) || (bo > best ? to : (bo < best ? tie : false));
    const float B = other ? bo : best;
    const int J = other ? jo : bestj;
    if (g == 0 && row_ok) rp[row] = make_float2(B, __int_as_float(J | (t ? TIE_BIT : 0)));
  }
}
}  // namespace sweep

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
  bool flag = false;
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
    if (!flag && tie && conf && bv > sp.thr) {          // rare: exact tie at the row maximum and the first one failed
      const float* cr = conf + ((long)n * g.L + i) * g.S;
      for (int j = bj + 1; j < g.S; ++j)
        if (cr[j] == bv && candidate_ok(sp, colmax, n, i, j, bv)) { bj = j; flag = true; break; }
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

// ------------------------------------------------------------------------------------------
// Sinkhorn pieces
__global__ __launch_bounds__(Cfg::THREADS, 2) void score_store_kernel(const sp_t* __restrict__ f0,
                                                                   const sp_t* __restrict__ f1, Geometry g,
                                                                   float scale, const uint8_t* __restrict__ mask0,
                                                                   const uint8_t* __restrict__ mask1,
                                                                   float* __restrict__ z) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  int n, ti, tj;
  if (!score_tile(g, n, ti, tj)) return;
  const int m0 = ti * Cfg::BM, n0 = tj * Cfg::BN;
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(asrc_plain(f0 + (long)n * g.L * g.C, g.C), f1 + (long)n * g.S * g.C, g.C, g.L, g.S, g.C,
                     m0, n0, lds, acc);
  if (mask0) acc_to_sim<true, false>(acc, m0, n0, g.L, g.S, scale, mask0 + (long)n * g.L, mask1 + (long)n * g.S);
  else acc_to_sim<false, false>(acc, m0, n0, g.L, g.S, scale, nullptr, nullptr);
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[i][j][r];
        if (in_range(v)) z[((long)n * g.L + acc_row<Cfg>(m0, i, r)) * g.S + acc_col<Cfg>(n0, j)] = v;
      }
}

// u[n][i] = log_mu[i] - logsumexp_j(Zfull[i][j] + v[j]),  i in [0, L]  (row L = dustbin row),
// j over the S real columns plus the dustbin column (value alpha).  One wave per row.
//   grid (ceil((L+1)/4), N), 256 threads.
__global__ __launch_bounds__(256) void ot_row_lse_kernel(const float* __restrict__ z, Geometry g, float alpha,
                                                         float norm, const float* __restrict__ v,
                                                         float* __restrict__ u) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y, i = blockIdx.x * 4 + wave;
  if (i > g.L) return;
  const float* vn = v + (long)n * (g.S + 1);
  const float* zr = z + ((long)n * g.L + min(i, g.L - 1)) * g.S;
  const bool bin_row = i == g.L;
  float m = SENTINEL;
  for (int j = lane; j <= g.S; j += 64) {
    const float x = ((bin_row || j == g.S) ? alpha : zr[j]) + vn[j];
    m = fmaxf(m, x);
  }
  m = wave_max(m);
  float s = 0.f;
  for (int j = lane; j <= g.S; j += 64) {
    const float x = ((bin_row || j == g.S) ? alpha : zr[j]) + vn[j];
    s += expf(x - m);
  }
  s = wave_sum(s);
  const float log_mu = bin_row ? logf((float)g.S) + norm : norm;
  if (lane == 0) u[(long)n * (g.L + 1) + i] = log_mu - (m + logf(s));
}

// column partial (max, sum exp) of Zfull[i][j] + u[i] over a chunk of rows.
//   grid (ceil((S+1)/64), RCH, N), 256 threads = 64 columns x 4 row lanes
constexpr int OT_RCH = 128;      // rows of the column partial buffer per pair (>= workgroups per pair of the fused passes)
__global__ __launch_bounds__(256) void ot_col_part_kernel(const float* __restrict__ z, Geometry g, float alpha,
                                                          const float* __restrict__ u,
                                                          float2* __restrict__ part) {
  __shared__ float2 red[4][64];
  const int n = blockIdx.z, j = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int rows = g.L + 1;
  const int per = ceil_div(rows, OT_RCH);
  const int r0 = blockIdx.y * per, r1 = min(r0 + per, rows);
  const float* un = u + (long)n * rows;
  float m = SENTINEL, s = 0.f;
  if (j <= g.S) {
    const bool bin_col = j == g.S;
    for (int i = r0 + rl; i < r1; i += 4) {
      const float x = ((bin_col || i == g.L) ? alpha : z[((long)n * g.L + i) * g.S + j]) + un[i];
      if (x > m) { s = s * expf(m - x) + 1.f; m = x; } else { s += expf(x - m); }
    }
  }
  red[rl][threadIdx.x & 63] = make_float2(m, s);
  __syncthreads();
  if (rl == 0 && j <= g.S) {
    float M = SENTINEL;
    for (int k = 0; k < 4; ++k) M = fmaxf(M, red[k][threadIdx.x].x);
    float Ssum = 0.f;
    for (int k = 0; k < 4; ++k) Ssum += in_range(red[k][threadIdx.x].x) ? red[k][threadIdx.x].y * expf(red[k][threadIdx.x].x - M) : 0.f;
    part[((long)n * (g.S + 1) + j) * OT_RCH + blockIdx.y] = make_float2(M, Ssum);
  }
}

__global__ void ot_col_merge_kernel(const float2* __restrict__ part, Geometry g, float norm,
                                    float* __restrict__ v) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long cols = (long)g.N * (g.S + 1);
  if (idx >= cols) return;
  const int j = (int)(idx % (g.S + 1));
  const float2* p = part + idx * OT_RCH;
  float m = SENTINEL;
  for (int k = 0; k < OT_RCH; ++k) m = fmaxf(m, p[k].x);
  float s = 0.f;
  for (int k = 0; k < OT_RCH; ++k) s += in_range(p[k].x) ? p[k].y * expf(p[k].x - m) : 0.f;
  const float log_nu = j == g.S ? logf((float)g.L) + norm : norm;
  v[idx] = log_nu - (m + logf(s));
}

// ---- one Sinkhorn iteration in ONE pass over Z ------------------------------------------------------------------
// u = log_mu - LSE_j(Z + v) needs whole rows, v' = log_nu - LSE_i(Z + u) needs whole columns: two sweeps over the
// 92 MB-per-pair volume per iteration when done as separate kernels (plus 32 column-partial rows).  Here a workgroup
// owns a contiguous range of rows and thread t owns the columns {t, t + 256, ...} for the whole kernel: it loads its
// CPT entries of a row ONCE (coalesced: the block reads 1 KB per instruction), keeps them in registers through the
// block-wide row reduction (-> u_i) and then folds them, now with u_i, into its private running column statistics --
// every element of Z crosses HBM once per iteration.  R rows are processed per round so that one pair of block
// reductions (max, sum) serves R rows.  The dustbin column (j = S) is an extra lane-private term of every row; the
// dustbin row (i = L, constant alpha) only needs u_L = log(S) + norm - LSE_j(alpha + v_j), computed by the first
// workgroup of the pair and added analytically by the merge kernel.
//   grid (WGP, N), 256 threads;  part [N][WGP][S + 1] (max, sum exp) of Z[i][j] + u[i] over the workgroup's rows.
template <int CPT, int R>
__global__ __launch_bounds__(256) void ot_iter_kernel(const float* __restrict__ z, Geometry g, float alpha, float norm,
                                                      const float* __restrict__ v, float* __restrict__ u,
                                                      float2* __restrict__ part, int rows_per_wg) {
  __shared__ float red[R][4];
  __shared__ float bc[R];
  const int n = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int S = g.S, L = g.L;
  const float* vn = v + (long)n * (S + 1);
  float vk[CPT], cm[CPT], cs[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int j = t + 256 * k;
    vk[k] = j <= S ? vn[j] : 0.f;
    cm[k] = SENTINEL; cs[k] = 0.f;
  }
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, L);
  for (int rb = r0; rb < r1; rb += R) {
    float zz[R][CPT], tm[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = min(rb + r, L - 1);
      const float* zr = z + ((long)n * L + i) * S;
      tm[r] = SENTINEL;
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const int j = t + 256 * k;
        zz[r][k] = j < S ? zr[j] : (j == S ? alpha : SENTINEL);       // dustbin column; beyond it: never contributes
        tm[r] = fmaxf(tm[r], zz[r][k] + vk[k]);
      }
    }
    // block-wide row maxima, then sums of exp
#pragma unroll
    for (int r = 0; r < R; ++r) { const float m = wave_max(tm[r]); if (lane == 0) red[r][wave] = m; }
    __syncthreads();
    float rmax[R], ts[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      rmax[r] = fmaxf(fmaxf(red[r][0], red[r][1]), fmaxf(red[r][2], red[r][3]));
      ts[r] = 0.f;
#pragma unroll
      for (int k = 0; k < CPT; ++k) ts[r] += expf(zz[r][k] + vk[k] - rmax[r]);        // exp(SENTINEL - x) == 0
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) { const float sm = wave_sum(ts[r]); if (lane == 0) red[r][wave] = sm; }
    __syncthreads();
    if (t < R) {
      const float ssum = (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
      const float ui = norm - (rmax[t] + logf(ssum));                  // log_mu = norm for the real rows
      bc[t] = ui;
      if (rb + t < r1) u[(long)n * (L + 1) + rb + t] = ui;
    }
    __syncthreads();
    // fold the rows, now with their u, into the thread's column statistics
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (rb + r >= r1) break;                                         // block-uniform
      const float ui = bc[r];
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const float y = zz[r][k] + ui;
        const float mn = fmaxf(cm[k], y);
        cs[k] = cs[k] * expf(cm[k] - mn) + expf(y - mn);
        cm[k] = mn;
      }
    }
    __syncthreads();                                                   // red / bc are reused by the next round
  }
  float2* pn = part + ((long)n * gridDim.x + blockIdx.x) * (S + 1);
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int j = t + 256 * k;
    if (j <= S) pn[j] = make_float2(cm[k], cs[k]);
  }
  if (blockIdx.x == 0) {               // u of the dustbin row: log(S) + norm - LSE_j(alpha + v_j), j = 0 .. S
    float m = SENTINEL;
#pragma unroll
    for (int k = 0; k < CPT; ++k) if (t + 256 * k <= S) m = fmaxf(m, alpha + vk[k]);
    m = wave_max(m);
    if (lane == 0) red[0][wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) if (t + 256 * k <= S) sm += expf(alpha + vk[k] - m);
    __syncthreads();
    sm = wave_sum(sm);
    if (lane == 0) red[0][wave] = sm;
    __syncthreads();
    if (t == 0) u[(long)n * (L + 1) + L] = logf((float)S) + norm - (m + logf((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])));
  }
}

// v[n][j] = log_nu[j] - LSE over {the P workgroup partials of column j, the dustbin-row term alpha + u[n][L]}
__global__ void ot_col_merge2_kernel(const float2* __restrict__ part, Geometry g, float alpha, float norm, int P,
                                     const float* __restrict__ u, float* __restrict__ v) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long cols = (long)g.N * (g.S + 1);
  if (idx >= cols) return;
  const int n = (int)(idx / (g.S + 1)), j = (int)(idx - (long)n * (g.S + 1));
  const float2* p = part + (long)n * P * (g.S + 1) + j;
  const float bin = alpha + u[(long)n * (g.L + 1) + g.L];
  // eight partials in flight per thread (as a dependent load -> compare chain the P partial rows cost one DRAM round
  // trip each: 48 us at P = 96); same reference and summation order as the plain loops
  float m = bin;
  for (int k0 = 0; k0 < P; k0 += 8) {
    float e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) e[q] = p[(long)min(k0 + q, P - 1) * (g.S + 1)].x;
#pragma unroll
    for (int q = 0; q < 8; ++q) m = fmaxf(m, e[q]);
  }
  float s = expf(bin - m);
  for (int k0 = 0; k0 < P; k0 += 8) {
    float2 e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) e[q] = p[(long)min(k0 + q, P - 1) * (g.S + 1)];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += (k0 + q < P && in_range(e[q].x)) ? e[q].y * expf(e[q].x - m) : 0.f;
  }
  const float log_nu = j == g.S ? logf((float)g.L) + norm : norm;
  v[idx] = log_nu - (m + logf(s));
}

// dustbin prefilter (coarse_matching.py:136-140): row i is dropped when the argmax of its
// assignment row (dustbin column included) is the dustbin; same for columns.
//   rowkill[n][i], colkill[n][j].  Ties resolve to the first index like torch.max.
__global__ __launch_bounds__(256) void ot_rowkill_kernel(const float* __restrict__ z, Geometry g, float alpha,
                                                         const float* __restrict__ u, const float* __restrict__ v,
                                                         uint8_t* __restrict__ rowkill) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y, i = blockIdx.x * 4 + wave;
  if (i >= g.L) return;
  const float* vn = v + (long)n * (g.S + 1);
  const float* zr = z + ((long)n * g.L + i) * g.S;
  float m = SENTINEL;
  for (int j = lane; j < g.S; j += 64) m = fmaxf(m, zr[j] + vn[j]);
  m = wave_max(m);
  // assignment = exp(z + u + v - norm): monotone in (z + v) along a row; bin wins only if strictly larger
  if (lane == 0) rowkill[(long)n * g.L + i] = (alpha + vn[g.S]) > m;
}
__global__ __launch_bounds__(256) void ot_colkill_kernel(const float* __restrict__ z, Geometry g, float alpha,
                                                         const float* __restrict__ u, const float* __restrict__ v,
                                                         uint8_t* __restrict__ colkill) {
  const int n = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= g.S) return;
  const float* un = u + (long)n * (g.L + 1);
  float m = SENTINEL;
  for (int i = 0; i < g.L; ++i) m = fmaxf(m, z[((long)n * g.L + i) * g.S + j] + un[i]);
  colkill[(long)n * g.S + j] = (alpha + un[g.L]) > m;
}

// conf = exp(z + u + v - norm) in place (+ full assignment matrix, + prefilter) and the row/col
// max partials of conf.  Tile = 128 x 128 like the GEMM kernels so that conf_partials applies.
__global__ __launch_bounds__(Cfg::THREADS, 2) void ot_finalize_kernel(float* __restrict__ z, Geometry g, float norm,
                                                                   const float* __restrict__ u,
                                                                   const float* __restrict__ v,
                                                                   const uint8_t* __restrict__ rowkill,
                                                                   const uint8_t* __restrict__ colkill,
                                                                   float* __restrict__ assign,
                                                                   float2* __restrict__ rowmax_part,
                                                                   float* __restrict__ colmax_part) {
  const int n = blockIdx.z, m0 = blockIdx.y * Cfg::BM, n0 = blockIdx.x * Cfg::BN;
  const float* un = u + (long)n * (g.L + 1);
  const float* vn = v + (long)n * (g.S + 1);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = acc_col<Cfg>(n0, j);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row<Cfg>(m0, i, r);
        float c = -1.f;
        if (row < g.L && col < g.S) {
          const long o = ((long)n * g.L + row) * g.S + col;
          c = expf(z[o] + un[row] + vn[col] - norm);
          if (rowkill && (rowkill[(long)n * g.L + row] || colkill[(long)n * g.S + col])) c = 0.f;
          // conf_matrix is a VIEW of assign_matrix in the reference (:133), so the prefilter
          // zeroing (:139-140) is visible in conf_matrix_with_bin (:143) as well
          if (assign) assign[((long)n * (g.L + 1) + row) * (g.S + 1) + col] = c;
          z[o] = c;
        }
        acc[i][j][r] = c;
      }
    }
  conf_partials<false>(acc, m0, n0, n, g, blockIdx.x, blockIdx.y, rowmax_part, colmax_part);
}

// ---- round 2: the Sinkhorn passes as ONE row-streaming kernel -------------------------------------------------------
// ot_iter_kernel above issues CPT scalar loads per row and consumes them at once (no load is in flight while the block
// reduces and exponentiates: 1.35 TB/s measured), pays three expf per element, and its grid was capped at 32 workgroups
// per pair (4 waves per CU at N = 8).  ot_pass_kernel keeps the ownership scheme (a workgroup owns a contiguous range of
// rows, a thread owns columns for the whole kernel) and changes the rest:
//   * a thread owns G4 groups of FOUR consecutive columns: one 16-byte load per group and row (S % 4 == 0: rows aligned);
//   * the rows of round k + 1 are loaded into a second register set before round k is processed;
//   * exponentials are v_exp_f32(x log2e); the running column statistics take ONE reference update
//     per column and round (R + 1 exponentials per R elements instead of 2 R); every thread derives u_i itself from the
//     block sums (no broadcast round trip) and the reduction buffers alternate by round parity: two barriers per round;
//   * FINAL = true is the last pass (ot_finalize_kernel's job) on the same skeleton: conf = exp(Z + u + v - norm) written
//     over Z (and into assign_matrix), per-row (max, FIRST arg-max, attained-twice flag) by a block reduction -- one
//     partial per row, PJ = 1 -- and per-workgroup column maxima (P = workgroups per pair partial rows).
//   grid (WGP, N), 256 threads.
namespace otp {
constexpr float L2E = 1.4426950408889634f;

// exp(x) as v_exp_f32(x log2 e).  The DIFFERENCE is formed first, never folded into an fma with a prescaled offset: with
// padding masks the potentials u, v of masked rows / columns are ~ +-1e9 (they cancel the -1e9 fill), and
// fma(y, log2e, -m log2e) would carry the rounding error of the 1.4e9-sized offset (+-64) into the exponent, where
// y - m is exact.  For the same reason conf is evaluated in the reference's order ((Z + u) + v) - norm: on masked
// entries the result IS rounding noise of that order, and the mutual-nearest test sees it.
__device__ __forceinline__ float ex(float x) { return __builtin_amdgcn_exp2f(x * L2E); }

// (value, first index | TIE) pairs: the better of two; equal values keep the smaller index and raise the flag
__device__ __forceinline__ void best_merge(float& b, int& w, float ob, int ow) {
  const int jb = w & ~sweep::TIE_BIT, jo = ow & ~sweep::TIE_BIT;
  const bool take = ob > b || (ob == b && jo < jb);
  const int tie = ob == b ? sweep::TIE_BIT : (take ? (ow & sweep::TIE_BIT) : (w & sweep::TIE_BIT));
  b = take ? ob : b;
  w = (take ? jo : jb) | tie;
}

#ifndef OTP_PREFETCH
#define OTP_PREFETCH 0           // 1: second register set for the next round's rows (256 VGPRs, 2 workgroups / SIMD set) -- A/B
#endif
template <int G4, int R, bool FINAL>
__global__ __launch_bounds__(256, OTP_PREFETCH ? 2 : 3) void ot_pass_kernel(float* __restrict__ z, Geometry g, float alpha, float norm,
                                                      const float* __restrict__ v, float* __restrict__ u,
                                                      float2* __restrict__ part, int rows_per_wg,
                                                      const uint8_t* __restrict__ rowkill, const uint8_t* __restrict__ colkill,
                                                      float* __restrict__ assign, float2* __restrict__ rowmax_part,
                                                      float* __restrict__ colmax_part) {
  __shared__ float red_a[2][R][4], red_b[2][R][4];
  __shared__ int red_w[2][R][4];
  const int n = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int S = g.S, L = g.L, S4 = S >> 2;
  const float* vn = v + (long)n * (S + 1);
  f32x4 vk[G4], ca[G4], cb[G4];         // column constants; ITER: running (reference, sum);  FINAL: ca = running column maximum
  unsigned kill = 0;                    // FINAL: bit 4 k + e set: the prefilter zeroes this column
#pragma unroll
  for (int k = 0; k < G4; ++k) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * (t + 256 * k) + e;
      const float x = j <= S ? vn[j] : 0.f;
      vk[k][e] = x;
      if (FINAL && colkill && j < S && colkill[(long)n * S + j]) kill |= 1u << (4 * k + e);
      ca[k][e] = FINAL ? -1.f : SENTINEL; cb[k][e] = 0.f;
    }
  }
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, L);
  if (r0 >= r1) return;                 // (never: the host sizes the grid to the rows)
  f32x4 zc[R][G4];
#if OTP_PREFETCH
  f32x4 zn[R][G4];
#endif
#define OTP_LOAD(dst_, rb_)                                                                              \
  _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                        \
    const float* zr__ = z + ((long)n * L + min((rb_) + r, L - 1)) * S;                                   \
    _Pragma("unroll") for (int k = 0; k < G4; ++k) {                                                     \
      const int q__ = t + 256 * k;                                                                       \
      dst_[r][k] = q__ < S4 ? *reinterpret_cast<const f32x4*>(zr__ + 4 * q__)                            \
                            : (q__ == S4 ? f32x4{alpha, SENTINEL, SENTINEL, SENTINEL} : f32x4{SENTINEL, SENTINEL, SENTINEL, SENTINEL}); \
    }                                                                                                    \
  }
#if OTP_PREFETCH
  OTP_LOAD(zc, r0)
#endif
  int par = 0;
  for (int rb = r0; rb < r1; rb += R, par ^= 1) {
#if OTP_PREFETCH
    const bool more = rb + R < r1;                       // block-uniform
    if (more) OTP_LOAD(zn, rb + R)
#else
    OTP_LOAD(zc, rb)                                     // latency is hidden by the other workgroups of the CU (3 x 4 waves)
#endif
    if (!FINAL) {
      // ---- u_i = log_mu - LSE_j(Z_ij + v_j): block maximum, then block sum of exponentials
      float rmx[R], ui[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float m = SENTINEL;
#pragma unroll
        for (int k = 0; k < G4; ++k) {
          const f32x4 a = zc[r][k] + vk[k];
          m = fmaxf(fmaxf(m, a[0]), a[1]); m = fmaxf(fmaxf(m, a[2]), a[3]);
        }
        m = wave_max(m);
        if (lane == 0) red_a[par][r][wave] = m;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        rmx[r] = fmaxf(fmaxf(red_a[par][r][0], red_a[par][r][1]), fmaxf(red_a[par][r][2], red_a[par][r][3]));
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < G4; ++k) {
          const f32x4 a = zc[r][k] + vk[k];
#pragma unroll
          for (int e = 0; e < 4; ++e) sm += ex(a[e] - rmx[r]);                       // exp(-huge) == 0 for the padding
        }
        sm = wave_sum(sm);
        if (lane == 0) red_b[par][r][wave] = sm;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float ssum = (red_b[par][r][0] + red_b[par][r][1]) + (red_b[par][r][2] + red_b[par][r][3]);
        ui[r] = norm - (rmx[r] + logf(ssum));            // log_mu = norm for the real rows
        if (t == 0 && rb + r < r1) u[(long)n * (L + 1) + rb + r] = ui[r];
        if (rb + r >= r1) ui[r] = SENTINEL;              // rows beyond the range: y = SENTINEL below, contribute nothing
      }
      // ---- fold the R rows, now with their u, into the thread's column statistics: one reference update per round
#pragma unroll
      for (int k = 0; k < G4; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float y[R], m = ca[k][e];
#pragma unroll
          for (int r = 0; r < R; ++r) { y[r] = ui[r] > SENTINEL ? zc[r][k][e] + ui[r] : SENTINEL; m = fmaxf(m, y[r]); }
          float acc = cb[k][e] * ex(ca[k][e] - m);
#pragma unroll
          for (int r = 0; r < R; ++r) acc += ex(y[r] - m);
          ca[k][e] = m; cb[k][e] = acc;
        }
      }
    } else {
      // ---- conf_ij = exp(((Z_ij + u_i) + v_j) - norm), in the reference's association (see ex() above)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = rb + r;
        const bool valid = i < r1;                       // block-uniform
        const int ic = min(i, L - 1);
        const float ub = u[(long)n * (L + 1) + ic];
        const bool rk = rowkill && rowkill[(long)n * L + ic];
        float* zr = z + ((long)n * L + ic) * S;
        float* ar = assign ? assign + ((long)n * (L + 1) + ic) * (S + 1) : nullptr;
        float best = -1.f; int bw = 0;
#pragma unroll
        for (int k = 0; k < G4; ++k) {
          const int q = t + 256 * k;
          f32x4 c;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = ex(((zc[r][k][e] + ub) + vk[k][e]) - norm);
            if (rk || ((kill >> (4 * k + e)) & 1u)) x = 0.f;       // skh_prefilter: coarse_matching.py:136-140
            c[e] = (q < S4 && valid) ? x : -1.f;
          }
          if (q < S4 && valid) {
            *reinterpret_cast<f32x4*>(zr + 4 * q) = c;
            // conf_matrix is a VIEW of assign_matrix in the reference (:133): the prefilter zeroing is visible there too
            // (row pitch S + 1: only 4-byte aligned.  Scalar stores: one unaligned dwordx4 per group measured 40 % slower)
            if (ar) { ar[4 * q] = c[0]; ar[4 * q + 1] = c[1]; ar[4 * q + 2] = c[2]; ar[4 * q + 3] = c[3]; }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {                  // this thread's columns ascend with (k, e): > keeps the first
            if (c[e] > best) { best = c[e]; bw = 4 * q + e; }
            else if (c[e] == best) bw |= sweep::TIE_BIT;
            ca[k][e] = fmaxf(ca[k][e], c[e]);
          }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          const float ob = __shfl_xor(best, o, 64);
          const int ow = __shfl_xor(bw, o, 64);
          best_merge(best, bw, ob, ow);
        }
        if (lane == 0) { red_a[par][r][wave] = best; red_w[par][r][wave] = bw; }
      }
      __syncthreads();
      if (t < R && rb + t < r1) {
        float b = red_a[par][t][0]; int w = red_w[par][t][0];
#pragma unroll
        for (int k = 1; k < 4; ++k) best_merge(b, w, red_a[par][t][k], red_w[par][t][k]);
        rowmax_part[(long)n * L + rb + t] = make_float2(b, __int_as_float(w));       // PJ = 1
      }
    }
#if OTP_PREFETCH
    if (more) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < G4; ++k) zc[r][k] = zn[r][k];
    }
#endif
  }
#undef OTP_LOAD
  if (FINAL) {
    float* cp = colmax_part + ((long)n * gridDim.x + blockIdx.x) * S;
#pragma unroll
    for (int k = 0; k < G4; ++k) if (t + 256 * k < S4) *reinterpret_cast<f32x4*>(cp + 4 * (t + 256 * k)) = ca[k];
    return;
  }
  float2* pn = part + ((long)n * gridDim.x + blockIdx.x) * (S + 1);
#pragma unroll
  for (int k = 0; k < G4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * (t + 256 * k) + e;
      if (j <= S) pn[j] = make_float2(ca[k][e], cb[k][e]);
    }
  if (blockIdx.x == 0) {               // u of the dustbin row: log(S) + norm - LSE_j(alpha + v_j), j = 0 .. S
    __syncthreads();
    float m = SENTINEL;
#pragma unroll
    for (int k = 0; k < G4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) if (4 * (t + 256 * k) + e <= S) m = fmaxf(m, alpha + vk[k][e]);
    m = wave_max(m);
    if (lane == 0) red_a[0][0][wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red_a[0][0][0], red_a[0][0][1]), fmaxf(red_a[0][0][2], red_a[0][0][3]));
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < G4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) if (4 * (t + 256 * k) + e <= S) sm += expf(alpha + vk[k][e] - m);
    sm = wave_sum(sm);
    if (lane == 0) red_b[0][0][wave] = sm;
    __syncthreads();
    if (t == 0)
      u[(long)n * (L + 1) + L] = logf((float)S) + norm - (m + logf((red_b[0][0][0] + red_b[0][0][1]) + (red_b[0][0][2] + red_b[0][0][3])));
  }
}
}  // namespace otp

// dustbin column / row / corner of the assignment matrix
__global__ void ot_assign_bins_kernel(Geometry g, float alpha, float norm, const float* __restrict__ u,
                                      const float* __restrict__ v, float* __restrict__ assign) {
  const int n = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const float* un = u + (long)n * (g.L + 1);
  const float* vn = v + (long)n * (g.S + 1);
  float* an = assign + (long)n * (g.L + 1) * (g.S + 1);
  if (t < g.L) an[(long)t * (g.S + 1) + g.S] = expf(alpha + un[t] + vn[g.S] - norm);
  if (t <= g.S) an[(long)g.L * (g.S + 1) + t] = expf(alpha + un[g.L] + vn[t] - norm);
}

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
  (void)hipMemsetAsync(w.ot_u, 0, sizeof(float) * g.N * (g.L + 1), st);
  (void)hipMemsetAsync(w.ot_v, 0, sizeof(float) * g.N * (g.S + 1), st);
  const long cols = (long)g.N * (g.S + 1);
  // row-streaming passes (otp::ot_pass_kernel): aligned rows and at most 5 x 1024 columns incl. the dustbin (indoor)
  const bool rowstream = (g.S & 3) == 0 && g.S + 1 <= 4 * 256 * 5;
  int wgs = 0;                                             // workgroups per pair of the row-streaming passes
  int rpws = 0;
  if (rowstream) {
    constexpr int R = 2;
    const int capP = [&] { const int a = ceil_div(g.L, 256) * 8; const int c = a > g.PI ? a : g.PI; return c < OT_RCH ? c : OT_RCH; }();
    wgs = 768 / g.N;                                       // ~3 workgroups (of 4 waves) per CU over the batch
    wgs = wgs < 1 ? 1 : (wgs > capP ? capP : wgs);
    if (wgs > ceil_div(g.L, R)) wgs = ceil_div(g.L, R);
    rpws = ceil_div(ceil_div(g.L, wgs), R) * R;
    wgs = ceil_div(g.L, rpws);
  }
  // fused iteration (one pass over Z): up to 19 x 256 (indoor) / 44 x 256 (outdoor 840 x 840) columns incl. the dustbin
  const int cpt = ceil_div(g.S + 1, 256);
  const bool fused = cpt <= 44;
  int wgp = 512 / (g.N > 0 ? g.N : 1);                     // ~2 workgroups per CU over the batch
  wgp = wgp < 1 ? 1 : (wgp > OT_RCH ? OT_RCH : wgp);       // the partial buffer holds OT_RCH rows per column
  if (wgp > ceil_div(g.L, 4)) wgp = ceil_div(g.L, 4);
  int rpw = ceil_div(g.L, wgp);
  rpw = ceil_div(rpw, 4) * 4;
  wgp = ceil_div(g.L, rpw);
  for (int it = 0; it < iters; ++it) {
    if (rowstream) {
      hipLaunchKernelGGL((otp::ot_pass_kernel<5, 2, false>), dim3(wgs, g.N), dim3(256), 0, st, conf_out, g, bin_score, norm, w.ot_v, w.ot_u,
                         w.ot_part, rpws, nullptr, nullptr, nullptr, nullptr, nullptr);
      hipLaunchKernelGGL(ot_col_merge2_kernel, dim3(ceil_div((int)cols, 256)), dim3(256), 0, st, w.ot_part, g, bin_score, norm, wgs, w.ot_u, w.ot_v);
      continue;
    }
    if (fused) {
      if (cpt <= 19) hipLaunchKernelGGL((ot_iter_kernel<19, 4>), dim3(wgp, g.N), dim3(256), 0, st, conf_out, g, bin_score, norm, w.ot_v, w.ot_u, w.ot_part, rpw);
      else hipLaunchKernelGGL((ot_iter_kernel<44, 2>), dim3(wgp, g.N), dim3(256), 0, st, conf_out, g, bin_score, norm, w.ot_v, w.ot_u, w.ot_part, rpw);
      hipLaunchKernelGGL(ot_col_merge2_kernel, dim3(ceil_div((int)cols, 256)), dim3(256), 0, st, w.ot_part, g, bin_score, norm, wgp, w.ot_u, w.ot_v);
      continue;
    }
    hipLaunchKernelGGL(ot_row_lse_kernel, dim3(ceil_div(g.L + 1, 4), g.N), dim3(256), 0, st, conf_out, g, bin_score, norm, w.ot_v, w.ot_u);
    hipLaunchKernelGGL(ot_col_part_kernel, dim3(ceil_div(g.S + 1, 64), OT_RCH, g.N), dim3(256), 0, st, conf_out, g, bin_score, w.ot_u, w.ot_part);
    hipLaunchKernelGGL(ot_col_merge_kernel, dim3(ceil_div((int)cols, 256)), dim3(256), 0, st, w.ot_part, g, norm, w.ot_v);
  }
  const uint8_t *rk = nullptr, *ck = nullptr;
  if (prefilter) {
    hipLaunchKernelGGL(ot_rowkill_kernel, dim3(ceil_div(g.L, 4), g.N), dim3(256), 0, st, conf_out, g, bin_score, w.ot_u, w.ot_v, w.rowkill);
    hipLaunchKernelGGL(ot_colkill_kernel, dim3(ceil_div(g.S, 256), g.N), dim3(256), 0, st, conf_out, g, bin_score, w.ot_u, w.ot_v, w.colkill);
    rk = w.rowkill; ck = w.colkill;
  }
  if (assign_out)
    hipLaunchKernelGGL(ot_assign_bins_kernel, dim3(ceil_div((g.L > g.S ? g.L : g.S) + 1, 256), g.N), dim3(256), 0, st, g, bin_score, norm, w.ot_u, w.ot_v, assign_out);
  if (rowstream) {
    hipLaunchKernelGGL((otp::ot_pass_kernel<5, 2, true>), dim3(wgs, g.N), dim3(256), 0, st, conf_out, g, bin_score, norm, w.ot_v, w.ot_u,
                       nullptr, rpws, rk, ck, assign_out, w.rowmax_part, w.colmax_part);
    LOFTR_CHECK_LAUNCH();
    Geometry gs = g;
    gs.PJ = 1; gs.PI = wgs;                                // one row partial per row, one column-maximum partial per workgroup
    return select_and_compact(gs, *p, *out, w, conf_out, st);
  }
  hipLaunchKernelGGL(ot_finalize_kernel, grid, block, 0, st, conf_out, g, norm, w.ot_u, w.ot_v, rk, ck, assign_out, w.rowmax_part, w.colmax_part);
  LOFTR_CHECK_LAUNCH();
  return select_and_compact(g, *p, *out, w, conf_out, st);
}
