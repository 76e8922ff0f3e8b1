// Sinkhorn kernels of coarse_match.hip (coarse_matching.py:121-143 + SuperGlue log_optimal_transport); included INSIDE its anonymous
// namespace, after score_sweep.h (one translation unit).  Split out for readability only.
#pragma once

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

// NT threads (NW = NT / 64 waves), G4 column groups of four per thread: 4 NT G4 >= S + 1.  AL: rows are 16-byte aligned
// (S % 4 == 0); otherwise the groups are loaded / stored as 4-byte aligned dwordx4 (sweep::F4U) and the ragged last group
// (S % 4 columns, then the dustbin) element by element, so that nothing is read beyond the volume.  PF: a second register
// set holds the next round's rows while this round is processed (one workgroup per CU: no neighbour hides the latency).
//   <256, 5, 2, ., ., false>, 3 workgroups / CU: S + 1 <= 5120 (indoor 60 x 80);  <512, 6, 2, ., ., true>, 1 / CU: S + 1 <= 12288
//   (outdoor 105 x 105).
template <int NT, int G4, int R, bool FINAL, bool AL, bool PF>
__global__ __launch_bounds__(NT, PF ? 1 : (NT == 256 ? 3 : 2)) void ot_pass_kernel(float* __restrict__ z, Geometry g, float alpha, float norm,
                                                      const float* __restrict__ v, float* __restrict__ u,
                                                      float2* __restrict__ part, int rows_per_wg,
                                                      const uint8_t* __restrict__ rowkill, const uint8_t* __restrict__ colkill,
                                                      float* __restrict__ assign, float2* __restrict__ rowmax_part,
                                                      float* __restrict__ colmax_part) {
  constexpr int NW = NT / 64;
  __shared__ float red_a[2][R][NW], red_b[2][R][NW];
  __shared__ int red_w[2][R][NW];
  const int n = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int S = g.S, L = g.L, S4 = S >> 2, rem = AL ? 0 : (S & 3);
  const float* vn = v + (long)n * (S + 1);
  f32x4 vk[G4], ca[G4], cb[G4];         // column constants; ITER: running (reference, sum);  FINAL: ca = running column maximum
  unsigned kill = 0;                    // FINAL: bit 4 k + e set: the prefilter zeroes this column
#pragma unroll
  for (int k = 0; k < G4; ++k) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * (t + NT * k) + e;
      const float x = j <= S ? vn[j] : 0.f;
      vk[k][e] = x;
      if (FINAL && colkill && j < S && colkill[(long)n * S + j]) kill |= 1u << (4 * k + e);
      ca[k][e] = FINAL ? -1.f : SENTINEL; cb[k][e] = 0.f;
    }
  }
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, L);
  if (r0 >= r1) return;                 // (never: the host sizes the grid to the rows)
  f32x4 zc[R][G4];
  f32x4 zn[PF ? R : 1][PF ? G4 : 1];
  // the group that holds the dustbin: S % 4 scores (read one by one), alpha, padding
  auto tail_group = [&](const float* zr) {
    f32x4 x{SENTINEL, SENTINEL, SENTINEL, SENTINEL};
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = e < rem ? zr[4 * S4 + e] : (e == rem ? alpha : SENTINEL);
    return x;
  };
#define OTP_LOAD(dst_, rb_)                                                                              \
  _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                        \
    const float* zr__ = z + ((long)n * L + min((rb_) + r, L - 1)) * S;                                   \
    _Pragma("unroll") for (int k = 0; k < G4; ++k) {                                                     \
      const int q__ = t + NT * k;                                                                        \
      if (q__ < S4) dst_[r][k] = AL ? *reinterpret_cast<const f32x4*>(zr__ + 4 * q__) : reinterpret_cast<const sweep::F4U*>(zr__ + 4 * q__)->v; \
      else if (q__ == S4) dst_[r][k] = tail_group(zr__);                                                 \
      else dst_[r][k] = f32x4{SENTINEL, SENTINEL, SENTINEL, SENTINEL};                                   \
    }                                                                                                    \
  }
  if constexpr (PF) { OTP_LOAD(zc, r0) }
  int par = 0;
  for (int rb = r0; rb < r1; rb += R, par ^= 1) {
    const bool more = rb + R < r1;                       // block-uniform
    if constexpr (PF) { if (more) { OTP_LOAD(zn, rb + R) } }
    else { OTP_LOAD(zc, rb) }                            // latency is hidden by the other workgroups of the CU (3 x 4 waves)
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
        float m = red_a[par][r][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, red_a[par][r][w]);
        rmx[r] = m;
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
        float ssum;
        if (NW == 4) ssum = (red_b[par][r][0] + red_b[par][r][1]) + (red_b[par][r][2] + red_b[par][r][3]);
        else { ssum = 0.f;
#pragma unroll
          for (int w = 0; w < NW; w += 2) ssum += red_b[par][r][w] + red_b[par][r][w + 1]; }
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
          const int q = t + NT * k;
          f32x4 c;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = ex(((zc[r][k][e] + ub) + vk[k][e]) - norm);
            if (rk || ((kill >> (4 * k + e)) & 1u)) x = 0.f;       // skh_prefilter: coarse_matching.py:136-140
            c[e] = ((AL ? q < S4 : 4 * q + e < S) && valid) ? x : -1.f;
          }
          if (q < S4 && valid) {
            if (AL) *reinterpret_cast<f32x4*>(zr + 4 * q) = c;
            else reinterpret_cast<sweep::F4U*>(zr + 4 * q)->v = c;
            // conf_matrix is a VIEW of assign_matrix in the reference (:133): the prefilter zeroing is visible there too
            // (row pitch S + 1: only 4-byte aligned.  Scalar stores: +147 us for the 737 MB at N = 8, i.e. 5 TB/s -- already the HBM
            //  write rate; one unaligned dwordx4 per group measured 40 % slower, a separate aligned fill kernel re-reading conf 120 us slower)
            if (ar) { ar[4 * q] = c[0]; ar[4 * q + 1] = c[1]; ar[4 * q + 2] = c[2]; ar[4 * q + 3] = c[3]; }
          } else if (!AL && q == S4 && valid) {
#pragma unroll
            for (int e = 0; e < 3; ++e)
              if (e < rem) { zr[4 * q + e] = c[e]; if (ar) ar[4 * q + e] = c[e]; }
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
        for (int k = 1; k < NW; ++k) best_merge(b, w, red_a[par][t][k], red_w[par][t][k]);
        rowmax_part[(long)n * L + rb + t] = make_float2(b, __int_as_float(w));       // PJ = 1
      }
    }
    if constexpr (PF) {
      if (more) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int k = 0; k < G4; ++k) zc[r][k] = zn[r][k];
      }
    }
  }
#undef OTP_LOAD
  if (FINAL) {
    float* cp = colmax_part + ((long)n * gridDim.x + blockIdx.x) * S;
#pragma unroll
    for (int k = 0; k < G4; ++k) {
      const int q = t + NT * k;
      if (AL) { if (q < S4) *reinterpret_cast<f32x4*>(cp + 4 * q) = ca[k]; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (4 * q + e < S) cp[4 * q + e] = ca[k][e];
      }
    }
    return;
  }
  float2* pn = part + ((long)n * gridDim.x + blockIdx.x) * (S + 1);
#pragma unroll
  for (int k = 0; k < G4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * (t + NT * k) + e;
      if (j <= S) pn[j] = make_float2(ca[k][e], cb[k][e]);
    }
  if (blockIdx.x == 0) {               // u of the dustbin row: log(S) + norm - LSE_j(alpha + v_j), j = 0 .. S
    __syncthreads();
    float m = SENTINEL;
#pragma unroll
    for (int k = 0; k < G4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) if (4 * (t + NT * k) + e <= S) m = fmaxf(m, alpha + vk[k][e]);
    m = wave_max(m);
    if (lane == 0) red_a[0][0][wave] = m;
    __syncthreads();
    m = red_a[0][0][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red_a[0][0][w]);
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < G4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) if (4 * (t + NT * k) + e <= S) sm += expf(alpha + vk[k][e] - m);
    sm = wave_sum(sm);
    if (lane == 0) red_b[0][0][wave] = sm;
    __syncthreads();
    if (t == 0) {
      float ssum;
      if (NW == 4) ssum = (red_b[0][0][0] + red_b[0][0][1]) + (red_b[0][0][2] + red_b[0][0][3]);
      else { ssum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 2) ssum += red_b[0][0][w] + red_b[0][0][w + 1]; }
      u[(long)n * (L + 1) + L] = logf((float)S) + norm - (m + logf(ssum));
    }
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

