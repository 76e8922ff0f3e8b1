// Linear layers of the LoFTR encoder on the matrix cores (fp16x3 split GEMM, gemm.h), with the surrounding elementwise
// work fused into the epilogue (feature map, ReLU, bias, LayerNorm, residual).
#include "linear.h"

using CfgGen = GemmCfg<128, 128, 2, 2>;     // generic tile: 4 waves, 64x64 per wave
using CfgLN256 = GemmCfg<64, 256, 1, 4>;    // full 256-wide rows in one block (LayerNorm)
using CfgLN128 = GemmCfg<128, 128, 2, 2>;   // full 128-wide rows in one block

// ------------------------------------------------------------------------------------------
template <typename Cfg, int EPI>
__global__ __launch_bounds__(Cfg::THREADS, 2) void linear_kernel(LinearArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int m0 = blockIdx.y * Cfg::BM, n0 = blockIdx.x * Cfg::BN;
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(p.a, p.w, p.ldw, p.M, p.N, p.K, m0, n0, lds, acc);
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = acc_col<Cfg>(n0, j);
      float bcol = 0.f;
      if (EPI == EPI_BIAS) bcol = col < p.N ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row<Cfg>(m0, i, r);
        if (row < p.M && col < p.N) {
          float v = acc[i][j][r];
          if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
          if (EPI == EPI_BIAS) v += bcol;
          if (EPI == EPI_GROUP_BIAS) v += p.bias[(long)(row / p.group) * p.N + col];
          p.out[(long)row * p.ldo + col] = v;
        }
      }
    }
}

int launch_linear(const LinearArgs& p, LinearEpi epi, hipStream_t st) {
  if (p.M <= 0) return LOFTR_OK;
  if (p.K % 4 != 0 || p.N <= 0) return LOFTR_ERR_UNSUPPORTED;
  dim3 grid(ceil_div(p.N, CfgGen::BN), ceil_div(p.M, CfgGen::BM));
  dim3 block(CfgGen::THREADS);
  TimedLaunch tl(LOFTR_T_LINEAR, st);
  switch (epi) {
    case EPI_STORE: hipLaunchKernelGGL((linear_kernel<CfgGen, EPI_STORE>), grid, block, 0, st, p); break;
    case EPI_RELU: hipLaunchKernelGGL((linear_kernel<CfgGen, EPI_RELU>), grid, block, 0, st, p); break;
    case EPI_BIAS: hipLaunchKernelGGL((linear_kernel<CfgGen, EPI_BIAS>), grid, block, 0, st, p); break;
    case EPI_GROUP_BIAS: hipLaunchKernelGGL((linear_kernel<CfgGen, EPI_GROUP_BIAS>), grid, block, 0, st, p); break;
  }
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ------------------------------------------------------------------------------------------
template <typename Cfg>
__global__ __launch_bounds__(Cfg::THREADS, 2) void proj_kernel(ProjArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int m0 = blockIdx.y * Cfg::BM;
  const int nglob = blockIdx.x * Cfg::BN;          // column in the concatenated [nseg*C] output
  const int seg = nglob / p.C;                     // block-uniform (BN divides C)
  const int n0 = nglob - seg * p.C;
  f32x16 acc[Cfg::TM][Cfg::TN];
  ASrc a = ASrc{p.a, p.C, nullptr, 0, 1 << 30, nullptr};
  gemm_mainloop<Cfg>(a, p.w[seg], p.C, p.M, p.C, p.C, m0, n0, lds, acc);
  const int kind = p.kind[seg];
  float* out = p.out[seg];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row<Cfg>(m0, i, r);
      if (row >= p.M) continue;
      float mk = p.mask ? (p.mask[row] ? 1.f : 0.f) : 1.f;
      if (kind == 2) mk *= p.inv_s;
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const int col = acc_col<Cfg>(n0, j);
        float v = acc[i][j][r];
        if (kind != 2) v = v > 0.f ? v + 1.f : expf(v);     // elu(v)+1
        out[(long)row * p.C + col] = v * mk;
      }
    }
}

int launch_proj(const ProjArgs& p, hipStream_t st) {
  if (p.M <= 0) return LOFTR_OK;
  if (p.C % CfgGen::BN != 0 || p.nseg < 1 || p.nseg > 3) return LOFTR_ERR_UNSUPPORTED;
  dim3 grid(p.nseg * p.C / CfgGen::BN, ceil_div(p.M, CfgGen::BM));
  TimedLaunch tl(LOFTR_T_PROJ, st);
  hipLaunchKernelGGL((proj_kernel<CfgGen>), grid, dim3(CfgGen::THREADS), 0, st, p);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ------------------------------------------------------------------------------------------
// GEMM + LayerNorm (+ residual).  One block spans the whole row (BN == C), so the row statistics
// are a reduction over the TN tiles of a lane, the 32 lanes of a half-wave and the WN waves.
template <typename Cfg, bool ATTN>
__global__ __launch_bounds__(Cfg::THREADS, 2) void linear_ln_kernel(LinearLNArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int m0 = blockIdx.x * Cfg::BM;
  f32x16 acc[Cfg::TM][Cfg::TN];
  if (ATTN) {
    const long n = blockIdx.y;
    p.a.p0 += n * p.M * (long)p.a.ld0;
    p.out += n * p.M * (long)p.C;
    AttnXform ax{p.attn_kv + n * (8 * 33 * 32), p.v_length, p.attn_eps, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    gemm_mainloop<Cfg, AttnXform>(p.a, p.w + n * (long)p.C * p.C, p.ldw, p.M, p.C, p.K, m0, 0, lds, acc, ax);
  } else {
    gemm_mainloop<Cfg>(p.a, p.w, p.ldw, p.M, p.C, p.K, m0, 0, lds, acc);
  }
  __syncthreads();                                  // all waves done with the staging buffers
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const float inv_c = 1.f / (float)p.C;

  float* red = lds;                                 // [BM][WN] partial sums
  float* mean_s = lds + Cfg::BM * Cfg::WN;          // [BM]
  float* rstd_s = mean_s + Cfg::BM;                 // [BM]
  auto lrow_of = [&](int i, int r) {
    return wm * Cfg::WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  };
  // pass 1: mean
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) s += acc[i][j][r];
      s = half_sum(s);
      if ((lane & 31) == 0) red[lrow_of(i, r) * Cfg::WN + wn] = s;
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
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float mu = mean_s[lrow_of(i, r)];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) { float d = acc[i][j][r] - mu; s += d * d; }
      s = half_sum(s);
      if ((lane & 31) == 0) red[lrow_of(i, r) * Cfg::WN + wn] = s;
    }
  __syncthreads();
  if (threadIdx.x < Cfg::BM) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < Cfg::WN; ++w) s += red[threadIdx.x * Cfg::WN + w];
    rstd_s[threadIdx.x] = 1.f / sqrtf(s * inv_c + p.eps);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = acc_col<Cfg>(0, j);
    const float g = p.gamma[col], b = p.beta[col];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row<Cfg>(m0, i, r);
        if (row < p.M) {
          const int lr = lrow_of(i, r);
          float v = (acc[i][j][r] - mean_s[lr]) * rstd_s[lr] * g + b;
          const long o = (long)row * p.C + col;
          if (p.residual) v += p.residual[o];
          p.out[o] = v;
        }
      }
  }
}

int launch_linear_ln(const LinearLNArgs& p, hipStream_t st) {
  if (p.M <= 0) return LOFTR_OK;
  if (p.K % 4 != 0) return LOFTR_ERR_UNSUPPORTED;
  TimedLaunch tl(LOFTR_T_LINEAR_LN, st);
  if (p.attn_kv) {
    if (p.C != 256 || p.K != 256 || p.residual || p.nb <= 0) return LOFTR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((linear_ln_kernel<CfgLN256, true>), dim3(ceil_div(p.M, CfgLN256::BM), p.nb),
                       dim3(CfgLN256::THREADS), 0, st, p);
  } else if (p.C == 256) {
    hipLaunchKernelGGL((linear_ln_kernel<CfgLN256, false>), dim3(ceil_div(p.M, CfgLN256::BM)),
                       dim3(CfgLN256::THREADS), 0, st, p);
  } else if (p.C == 128) {
    hipLaunchKernelGGL((linear_ln_kernel<CfgLN128, false>), dim3(ceil_div(p.M, CfgLN128::BM)),
                       dim3(CfgLN128::THREADS), 0, st, p);
  } else {
    return LOFTR_ERR_UNSUPPORTED;
  }
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int loftr_linear_fwd(const float* a, const float* w, float* out, int M, int N, int K,
                                void* stream) {
  LOFTR_CHECK_ARG(a && w && out && M >= 0 && N > 0 && K > 0);
  LinearArgs p{asrc_plain(a, K), w, K, out, N, M, N, K, nullptr, 1};
  return launch_linear(p, EPI_STORE, (hipStream_t)stream);
}
