// Linear layers of the LoFTR encoder on the matrix cores (split-fp16 GEMM core, gemm.h), with the
// surrounding elementwise work fused into the epilogue (feature map, attention normaliser, ReLU,
// bias, LayerNorm, residual, SP re-encoding for the next GEMM).
#include "linear.h"

using CfgGen = GemmCfg<128, 128, 2, 2>;     // generic tile: 4 waves, 64x64 per wave
#ifdef LOFTR_LN_DMA
using CfgLN256 = GemmCfg<128, 256, 2, 4, 3>;  // full 256-wide rows in one block (LayerNorm), 8 waves, DMA ring
#else
using CfgLN256 = GemmCfg<64, 256, 1, 4>;    // full 256-wide rows in one block (LayerNorm)
#endif
using CfgLN128 = GemmCfg<128, 128, 2, 2>;   // full 128-wide rows in one block

// ------------------------------------------------------------------------------------------
// Epilogues come in a FULL flavour (whole tile inside the matrix: no predicates, the common case)
// and a guarded one; 32-bit offsets from a block-uniform tile base pointer.
template <typename Cfg, bool FULL, bool OUT_F32, bool OUT_SP, int BIAS, bool RELU>
__device__ __forceinline__ void linear_epilogue(const LinearArgs& p, f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0) {
  const EpiLane<Cfg> e;
  float* of = OUT_F32 ? p.out_f32 + (long)m0 * p.ldo + n0 : nullptr;
  sp_t* os = OUT_SP ? p.out_sp + (long)m0 * p.ldo + n0 : nullptr;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + e.lcol + j * 32;
    const bool cok = FULL || col < p.N;
    const float bcol = (BIAS == 1 && cok) ? p.bias[col] : 0.f;
    const float wsc = (p.wscale_inv && cok) ? p.wscale_inv[col] : 1.f;      // undo the weight row's power-of-two scale
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      f32x16 v = acc[i][j] * wsc;
      if (p.ascale_inv) {                                                     // ... and the A row's (block-uniform branch)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= p.ascale_inv[FULL ? m0 + e.lrow + e.rr(i, r) : min(m0 + e.lrow + e.rr(i, r), p.M - 1)];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = e.lrow + e.rr(i, r);
        if (BIAS == 1) v[r] += bcol;
        if (BIAS == 2) { if (FULL || (m0 + trow < p.M && cok)) v[r] += p.bias[(unsigned)(((m0 + trow) / p.group) * p.N + col)]; }
        if (RELU) v[r] = fmaxf(v[r], 0.f);
      }
      const unsigned rowb = (unsigned)p.ldo * 4u;
      if (OUT_F32) {
        if (FULL) {                                  // 32-bit byte offsets from the block-uniform tile base (gemm.h: st_off)
          const unsigned off = (unsigned)(e.lrow * p.ldo + e.lcol + j * 32) * 4u;
#pragma unroll
          for (int r = 0; r < 16; ++r) st_off(of, off + (unsigned)e.rr(i, r) * rowb, v[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int trow = e.lrow + e.rr(i, r);
            if (m0 + trow < p.M && cok) of[(unsigned)(trow * p.ldo + e.lcol + j * 32)] = v[r];
          }
        }
      }
      if (OUT_SP) {
        uint32_t w[16];
        sp_words16(v, e.odd, w);
        if (FULL) {
          const unsigned off = (unsigned)(e.lrow * p.ldo + e.spcol + j * 32) * 4u;
#pragma unroll
          for (int r = 0; r < 16; ++r) st_off(os, off + (unsigned)e.rr(i, r) * rowb, w[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int trow = e.lrow + e.rr(i, r);
            if (m0 + trow < p.M && cok) os[(unsigned)(trow * p.ldo + e.spcol + j * 32)] = w[r];
          }
        }
      }
    }
  }
}

template <typename Cfg, bool OUT_F32, bool OUT_SP, int BIAS, bool RELU>
__global__ __launch_bounds__(Cfg::THREADS, 2) void linear_kernel(LinearArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  int tm, tn;
  if (!xcd_tile(ceil_div(p.M, Cfg::BM), ceil_div(p.N, Cfg::BN), tm, tn)) return;
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(p.a, p.w, p.ldw, p.M, p.N, p.K, m0, n0, lds, acc);
  if (m0 + Cfg::BM <= p.M && n0 + Cfg::BN <= p.N) linear_epilogue<Cfg, true, OUT_F32, OUT_SP, BIAS, RELU>(p, acc, m0, n0);
  else linear_epilogue<Cfg, false, OUT_F32, OUT_SP, BIAS, RELU>(p, acc, m0, n0);
}

int launch_linear(const LinearArgs& p, hipStream_t st) {
  if (p.M <= 0) return LOFTR_OK;
  if (p.K % 32 != 0 || p.N <= 0 || (p.out_sp && p.N % 32 != 0)) return LOFTR_ERR_UNSUPPORTED;
  dim3 grid(xcd_grid(ceil_div(p.M, CfgGen::BM), ceil_div(p.N, CfgGen::BN))), block(CfgGen::THREADS);
  TimedLaunch tl(LOFTR_T_LINEAR, st);
  const bool f = p.out_f32 != nullptr, s = p.out_sp != nullptr;
  // the combinations the matching path uses
  if (f && !s && p.bias_mode == 0 && !p.relu)
    hipLaunchKernelGGL((linear_kernel<CfgGen, true, false, 0, false>), grid, block, 0, st, p);       // loftr_linear_fwd
  else if (!f && s && p.bias_mode == 0 && p.relu) {
#ifdef LOFTR_LINEAR_DMA                              // A/B: 256 x 128 tiles on the LDS-DMA ring (as the convolutions) for the big mlp.0
#ifndef LOFTR_LINEAR_BIG
#define LOFTR_LINEAR_BIG 256, 128, 4, 2, 3
#endif
    using CfgBig = GemmCfg<LOFTR_LINEAR_BIG>;
    if (p.M >= 8192 && p.N % CfgBig::BN == 0)
      hipLaunchKernelGGL((linear_kernel<CfgBig, false, true, 0, true>), dim3(xcd_grid(ceil_div(p.M, CfgBig::BM), p.N / CfgBig::BN)),
                         dim3(CfgBig::THREADS), 0, st, p);
    else
#endif
    hipLaunchKernelGGL((linear_kernel<CfgGen, false, true, 0, true>), grid, block, 0, st, p);        // mlp.0 + ReLU
  }
  else if (!f && s && p.bias_mode == 1 && !p.relu)
    hipLaunchKernelGGL((linear_kernel<CfgGen, false, true, 1, false>), grid, block, 0, st, p);       // down_proj
  else if (f && !s && p.bias_mode == 1 && !p.relu)
    hipLaunchKernelGGL((linear_kernel<CfgGen, true, false, 1, false>), grid, block, 0, st, p);       // coarse context
  else if (f && !s && p.bias_mode == 2 && !p.relu)
    hipLaunchKernelGGL((linear_kernel<CfgGen, true, false, 2, false>), grid, block, 0, st, p);       // merge_feat windows
  else
    return LOFTR_ERR_UNSUPPORTED;
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ------------------------------------------------------------------------------------------
template <typename Cfg, bool FULL, bool ZSCALE>
__device__ __forceinline__ void proj_epilogue(const ProjArgs& p, f32x16 (&acc)[Cfg::TM][Cfg::TN], int seg, long n,
                                              int m0, int n0) {
  const int kind = p.kind[seg];
  const long row_base = n * p.M;
  const uint8_t* mask = p.mask ? p.mask + row_base + m0 : nullptr;
  const EpiLane<Cfg> e;
  float mk[Cfg::TM][16];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trow = e.lrow + e.rr(i, r);
      float m = 1.f;
      if (mask) m = mask[FULL ? trow : min(trow, p.M - 1 - m0)] ? 1.f : 0.f;
      mk[i][r] = kind == 2 ? m * p.inv_s : m;
    }
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + e.lcol + j * 32;                          // head = col / 32, d = col % 32 = lane & 31
    const float ksum = ZSCALE ? p.kv[(n * 8 + (col >> 5)) * (33 * 32) + 32 * 32 + (col & 31)] : 0.f;
    const float wsc = p.wsc[seg] ? p.wsc[seg][col] : 1.f;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      f32x16 v = acc[i][j] * wsc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = v[r];
        if (kind != 2) x = x > 0.f ? x + 1.f : __expf(x);          // elu(x)+1
        v[r] = x * mk[i][r];
      }
      if (ZSCALE) {
        // the 32 lanes of this half-wave hold the 32 channels of one head of each of these rows
        f32x16 den = v * ksum;
        half_sum16(den);
#pragma unroll
#ifdef LOFTR_EPI_OLD
        for (int r = 0; r < 16; ++r) v[r] *= p.v_length / (den[r] + p.eps);
#else
        for (int r = 0; r < 16; ++r) v[r] *= p.v_length * __builtin_amdgcn_rcpf(den[r] + p.eps);   // 1 / (Q . Ksum + eps), :44 -- v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE division
#endif
        uint32_t w[16];
        sp_words16(v, e.odd, w);
        sp_t* os = reinterpret_cast<sp_t*>(p.out[seg]) + (row_base + m0) * p.C + n0;
        if (FULL) {                                  // 32-bit byte offsets from the block-uniform tile base (gemm.h: st_off)
          const unsigned rowb = (unsigned)p.C * 4u, off = (unsigned)(e.lrow * p.C + e.spcol + j * 32) * 4u;
#pragma unroll
          for (int r = 0; r < 16; ++r) st_off(os, off + (unsigned)e.rr(i, r) * rowb, w[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int trow = e.lrow + e.rr(i, r);
            if (m0 + trow < p.M) os[(unsigned)(trow * p.C + e.spcol + j * 32)] = w[r];
          }
        }
      } else {
        float* of = reinterpret_cast<float*>(p.out[seg]) + (row_base + m0) * p.C + n0;
        if (FULL) {
          const unsigned rowb = (unsigned)p.C * 4u, off = (unsigned)(e.lrow * p.C + e.lcol + j * 32) * 4u;
#pragma unroll
          for (int r = 0; r < 16; ++r) st_off(of, off + (unsigned)e.rr(i, r) * rowb, v[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int trow = e.lrow + e.rr(i, r);
            if (m0 + trow < p.M) of[(unsigned)(trow * p.C + e.lcol + j * 32)] = v[r];
          }
        }
      }
    }
  }
}

template <typename Cfg, bool ZSCALE>
__global__ __launch_bounds__(Cfg::THREADS, 2) void proj_kernel(ProjArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  int tm, tn;
  if (!xcd_tile(ceil_div(p.M, Cfg::BM), p.nseg * p.C / Cfg::BN, tm, tn)) return;
  const int m0 = tm * Cfg::BM;
  const long n = blockIdx.y;                       // batch element
  const int nglob = tn * Cfg::BN;                  // column in the concatenated [nseg*C] output
  const int seg = nglob / p.C;                     // block-uniform (BN divides C)
  const int n0 = nglob - seg * p.C;
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(asrc_plain(p.a + n * p.M * p.C, p.C), p.w[seg], p.C, p.M, p.C, p.C, m0, n0, lds, acc);
  if (m0 + Cfg::BM <= p.M) proj_epilogue<Cfg, true, ZSCALE>(p, acc, seg, n, m0, n0);
  else proj_epilogue<Cfg, false, ZSCALE>(p, acc, seg, n, m0, n0);
}

int launch_proj(const ProjArgs& p, hipStream_t st) {
  if (p.M <= 0 || p.nbatch <= 0) return LOFTR_OK;
  if (p.C % CfgGen::BN != 0 || p.nseg < 1 || p.nseg > 3) return LOFTR_ERR_UNSUPPORTED;
  if (p.kv && (p.C != 256 || p.nseg != 1 || p.kind[0] != 0)) return LOFTR_ERR_UNSUPPORTED;
  dim3 grid(xcd_grid(ceil_div(p.M, CfgGen::BM), p.nseg * p.C / CfgGen::BN), p.nbatch);
  TimedLaunch tl(LOFTR_T_PROJ, st);
  if (p.kv) hipLaunchKernelGGL((proj_kernel<CfgGen, true>), grid, dim3(CfgGen::THREADS), 0, st, p);
  else hipLaunchKernelGGL((proj_kernel<CfgGen, false>), grid, dim3(CfgGen::THREADS), 0, st, p);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ------------------------------------------------------------------------------------------
// k / v projection + fused KV reduction (see linear.h: ProjKVArgs).   grid (xcd tiles, nbatch), 256 threads.
template <typename Cfg>
__global__ __launch_bounds__(Cfg::THREADS, 2) void proj_kv_kernel(ProjKVArgs p) {
  static_assert(Cfg::WM == 2 && Cfg::WN == 2 && Cfg::TM == 2 && Cfg::TN == 2, "one head ([K|V] = 64 columns) per wave");
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  int tm, tn;
  if (!xcd_tile(ceil_div(p.S, Cfg::BM), 2 * p.C / Cfg::BN, tm, tn)) return;
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
  const long n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % Cfg::WM, wn = wave / Cfg::WM;
  const uint8_t* mask = p.mask ? p.mask + n * p.S : nullptr;
  // Padding masks (MegaDepth batches, 840 x 840 padded from 840 x 560: a third of the row tiles): every row of a fully masked tile enters
  // K^T V and Ksum as K = +0 (elu + 1 > 0, times 0), V = +-0 -- its partial is +0 everywhere whatever the descriptors are.  Written as such,
  // without the projection (bit-identical).  Block-uniform.
  if (mask) {
    bool valid = false;
    for (int r = threadIdx.x; r < Cfg::BM; r += Cfg::THREADS) valid = valid || (m0 + r < p.S && mask[m0 + r] != 0);
    if (!__syncthreads_or(valid)) {
      if (wm == 0) {
        float* out = p.part + (((long)n * 8 + 2 * tn + wn) * p.splits + tm) * (33 * 32);
        for (int o = lane; o < 33 * 32; o += 64) out[o] = 0.f;
      }
      return;
    }
  }
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(asrc_plain(p.a + n * p.S * p.C, p.C), p.w_kv, p.C, p.S, 2 * p.C, p.C, m0, n0, lds, acc);
  const EpiLane<Cfg> e;
  // feature map / masks on the accumulators: tile j = 0 holds K (elu+1), j = 1 holds V (1/S); rows >= S contribute 0
  const float wk = p.wsc ? p.wsc[n0 + e.lcol] : 1.f, wv = p.wsc ? p.wsc[n0 + e.lcol + 32] : 1.f;
  float ksum = 0.f;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + e.lrow + e.rr(i, r);
      float mk = row < p.S ? 1.f : 0.f;
      if (mask) mk = (row < p.S && mask[row]) ? 1.f : 0.f;
      const float k = acc[i][0][r] * wk;
      acc[i][0][r] = (k > 0.f ? k + 1.f : __expf(k)) * mk;            // elu(k)+1, linear_attention.py:31-32,37-38
      acc[i][1][r] = acc[i][1][r] * (wv * (mk * p.inv_s));             // values * mask / v_length   :39-42
      ksum += acc[i][0][r];
    }
  // KV[d][v] += sum_rows K[row][d] V[row][v]: register r of the K tile IS the A operand of v_mfma_f32_32x32x2_f32
  // (A[i = d = lane&31][k = lane>>5] = K[row(r, lane>>5)][d]) and register r of the V tile the B operand.
  f32x16 kv;
#pragma unroll
  for (int r = 0; r < 16; ++r) kv[r] = 0.f;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) kv = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[i][0][r], acc[i][1][r], kv, 0, 0, 0);
  ksum += swap32(ksum);                                               // both halves hold rows of the same d
  // the two waves that share a head (wm = 0, 1) are summed through LDS in a fixed order
  float* red = lds + wn * (33 * 32);                                  // [wn][33][32]
  __syncthreads();                                                    // staging buffers no longer read
  if (wm == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = kv[r];
    if (lane < 32) red[32 * 32 + lane] = ksum;
  }
  __syncthreads();
  if (wm == 0) {
    const int head = 2 * tn + wn;
    float* out = p.part + (((long)n * 8 + head) * p.splits + tm) * (33 * 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31);
      out[o] = kv[r] + red[o];
    }
    if (lane < 32) out[32 * 32 + lane] = ksum + red[32 * 32 + lane];
  }
}

int launch_proj_kv(const ProjKVArgs& p, hipStream_t st) {
  if (p.S <= 0 || p.nbatch <= 0) return LOFTR_OK;
  if (p.C != 256 || p.splits != ceil_div(p.S, CfgGen::BM)) return LOFTR_ERR_UNSUPPORTED;
  dim3 grid(xcd_grid(ceil_div(p.S, CfgGen::BM), 2 * p.C / CfgGen::BN), p.nbatch);
  TimedLaunch tl(LOFTR_T_KV, st);
  hipLaunchKernelGGL((proj_kv_kernel<CfgGen>), grid, dim3(CfgGen::THREADS), 0, st, p);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ------------------------------------------------------------------------------------------
// GEMM + LayerNorm (+ residual).  One block spans the whole row (BN == C), so the row statistics
// are a reduction over the TN tiles of a lane, the 32 lanes of a half-wave and the WN waves.
template <typename Cfg, bool HAS_RES, bool OUT_F32, bool OUT_SP>
__global__ __launch_bounds__(Cfg::THREADS, 2) void linear_ln_kernel(LinearLNArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int m0 = blockIdx.x * Cfg::BM;
  const long n = blockIdx.y;
  const long row_base = n * p.M;
  f32x16 acc[Cfg::TM][Cfg::TN];
  {
    ASrc a = p.a;
    a.p0 += row_base * a.ld0;
    gemm_mainloop<Cfg>(a, p.w + n * p.w_batch_stride, p.ldw, p.M, p.C, p.K, m0, 0, lds, acc);
  }
  __syncthreads();                                  // all waves done with the staging buffers
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / Cfg::WM;
  const float inv_c = 1.f / (float)p.C;
  const EpiLane<Cfg> e;
  if (p.wscale_inv || p.out_scale != 0.f) {         // undo the power-of-two operand scales (exact) before the statistics
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const float sc = (p.wscale_inv ? p.wscale_inv[e.lcol + j * 32] : 1.f) * (p.out_scale != 0.f ? p.out_scale : 1.f);
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) acc[i][j] *= sc;
    }
  }
  const bool full = m0 + Cfg::BM <= p.M;            // block-uniform

  float* red = lds;                                 // [BM][WN] partial sums
  float* mean_s = lds + Cfg::BM * Cfg::WN;          // [BM]
  float* rstd_s = mean_s + Cfg::BM;                 // [BM]
  // pass 1: mean
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i) {
    f32x16 s = acc[i][0];
#pragma unroll
    for (int j = 1; j < Cfg::TN; ++j) s += acc[i][j];
    half_sum16(s);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((lane & 31) == 0) red[(e.lrow + e.rr(i, r)) * Cfg::WN + wn] = s[r];
  }
  __syncthreads();
  if (threadIdx.x < Cfg::BM) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < Cfg::WN; ++w) s += red[threadIdx.x * Cfg::WN + w];
    mean_s[threadIdx.x] = s * inv_c;
  }
  __syncthreads();
  // pass 2: variance of the deviations (two-pass, like a reference LayerNorm in fp32)
  float mu[Cfg::TM][16];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      mu[i][r] = mean_s[e.lrow + e.rr(i, r)];
      s[r] = 0.f;
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) { float d = acc[i][j][r] - mu[i][r]; s[r] += d * d; }
    }
    half_sum16(s);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((lane & 31) == 0) red[(e.lrow + e.rr(i, r)) * Cfg::WN + wn] = s[r];
  }
  __syncthreads();
  if (threadIdx.x < Cfg::BM) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < Cfg::WN; ++w) s += red[threadIdx.x * Cfg::WN + w];
    rstd_s[threadIdx.x] = 1.f / sqrtf(s * inv_c + p.eps);
  }
  __syncthreads();
  const long tile_base = (row_base + m0) * p.C;
  const float* res = HAS_RES ? p.residual + tile_base : nullptr;
  float* of = OUT_F32 ? p.out_f32 + tile_base : nullptr;
  sp_t* os = OUT_SP ? p.out_sp + tile_base : nullptr;
  float rs[Cfg::TM][16];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) rs[i][r] = rstd_s[e.lrow + e.rr(i, r)];
  constexpr unsigned CB = Cfg::BN * 4;              // bytes per row (the block spans the row: C == BN)
  const unsigned off_f = (unsigned)(e.lrow * Cfg::BN + e.lcol) * 4u, off_s = (unsigned)(e.lrow * Cfg::BN + e.spcol) * 4u;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const float gam = p.gamma[e.lcol + j * 32], bet = p.beta[e.lcol + j * 32];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      f32x16 v;
      if (full) {                                   // block-uniform: 32-bit byte offsets from the tile base (gemm.h: st_off)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = (acc[i][j][r] - mu[i][r]) * rs[i][r] * gam + bet;
          if (HAS_RES) v[r] += ld_off(res, off_f + (unsigned)(e.rr(i, r) * CB + j * 128));
        }
        if (OUT_F32) {
#pragma unroll
          for (int r = 0; r < 16; ++r) st_off(of, off_f + (unsigned)(e.rr(i, r) * CB + j * 128), v[r]);
        }
        if (OUT_SP) {
          uint32_t w[16];
          sp_words16(v, e.odd, w);
#pragma unroll
          for (int r = 0; r < 16; ++r) st_off(os, off_s + (unsigned)(e.rr(i, r) * CB + j * 128), w[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int trow = e.lrow + e.rr(i, r);
          const unsigned ro = (unsigned)(min(trow, p.M - 1 - m0) * p.C);
          v[r] = (acc[i][j][r] - mu[i][r]) * rs[i][r] * gam + bet;
          if (HAS_RES) v[r] += res[ro + e.lcol + j * 32];
        }
        if (OUT_F32) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int trow = e.lrow + e.rr(i, r);
            if (m0 + trow < p.M) of[(unsigned)(trow * p.C) + e.lcol + j * 32] = v[r];
          }
        }
        if (OUT_SP) {
          uint32_t w[16];
          sp_words16(v, e.odd, w);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int trow = e.lrow + e.rr(i, r);
            if (m0 + trow < p.M) os[(unsigned)(trow * p.C) + e.spcol + j * 32] = w[r];
          }
        }
      }
    }
  }
}

template <typename Cfg>
static void launch_ln_variant(const LinearLNArgs& p, hipStream_t st, int& rc) {
  const dim3 grid(ceil_div(p.M, Cfg::BM), p.nbatch), block(Cfg::THREADS);
  const bool r = p.residual != nullptr, f = p.out_f32 != nullptr, s = p.out_sp != nullptr;
  rc = LOFTR_OK;
  if (!r && !f && s) hipLaunchKernelGGL((linear_ln_kernel<Cfg, false, false, true>), grid, block, 0, st, p);      // merge + norm1
  else if (r && f && s) hipLaunchKernelGGL((linear_ln_kernel<Cfg, true, true, true>), grid, block, 0, st, p);     // mlp.2 + norm2 + x
  else if (r && f && !s) hipLaunchKernelGGL((linear_ln_kernel<Cfg, true, true, false>), grid, block, 0, st, p);   // same, single-layer API
  else rc = LOFTR_ERR_UNSUPPORTED;
}

int launch_linear_ln(const LinearLNArgs& p, hipStream_t st) {
  if (p.M <= 0 || p.nbatch <= 0) return LOFTR_OK;
  if (p.K % 32 != 0) return LOFTR_ERR_UNSUPPORTED;
  TimedLaunch tl(LOFTR_T_LINEAR_LN, st);
  int rc;
  if (p.C == 256) launch_ln_variant<CfgLN256>(p, st, rc);
  else if (p.C == 128) launch_ln_variant<CfgLN128>(p, st, rc);
  else return LOFTR_ERR_UNSUPPORTED;
  if (rc) return rc;
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
