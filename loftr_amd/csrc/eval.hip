// Evaluation caller of the matching path: what the reference's test_step computes from
// the matcher's output on the device, src/utils/metrics.py:31-68 (compute_symmetrical_epipolar_errors).
//   one thread per match: E = [t]_x R of the match's pair (9 FMAs, recomputed: cheaper than a second launch),
//   normalised homogeneous points, d = (p1' E p0)^2 * (1 / |(E p0)_xy|^2 + 1 / |(E' p1)_xy|^2), fp32 like the reference.
//   HBM: 2 x 8 B + 8 B in, 4 B out per match; the per-pair matrices (N x 34 floats) stay in L2.
#include "common.h"

namespace {

__global__ LOFTR_NO_PACKED_FP32 __launch_bounds__(256) void epipolar_errors_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                              const long* __restrict__ bids, const float* __restrict__ T,
                                                              const float* __restrict__ K0, const float* __restrict__ K1,
                                                              long M, int N, float* __restrict__ out) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const long b = bids[m];
  if (b < 0 || b >= N) { out[m] = __builtin_nanf(""); return; }      // not a pair of this batch (the reference would drop it)
  const float* t44 = T + b * 16;
  const float tx = t44[3], ty = t44[7], tz = t44[11];
  float E[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {                                       // E = [t]_x R   (metrics.py:55-56)
    const float r0 = t44[j], r1 = t44[4 + j], r2 = t44[8 + j];
    E[0][j] = -tz * r1 + ty * r2;
    E[1][j] = tz * r0 - tx * r2;
    E[2][j] = -ty * r0 + tx * r1;
  }
  const float* k0 = K0 + b * 9;
  const float* k1 = K1 + b * 9;
  const float x0 = (p0[2 * m] - k0[2]) / k0[0], y0 = (p0[2 * m + 1] - k0[5]) / k0[4];      // (:38-39)
  const float x1 = (p1[2 * m] - k1[2]) / k1[0], y1 = (p1[2 * m + 1] - k1[5]) / k1[4];
  const float a0 = x0 * E[0][0] + y0 * E[0][1] + E[0][2];             // Ep0 = p0 @ E.T
  const float a1 = x0 * E[1][0] + y0 * E[1][1] + E[1][2];
  const float a2 = x0 * E[2][0] + y0 * E[2][1] + E[2][2];
  const float s = x1 * a0 + y1 * a1 + a2;                             // p1 . Ep0
  const float c0 = x1 * E[0][0] + y1 * E[1][0] + E[2][0];             // Etp1 = p1 @ E
  const float c1 = x1 * E[0][1] + y1 * E[1][1] + E[2][1];
  out[m] = s * s * (1.0f / (a0 * a0 + a1 * a1) + 1.0f / (c0 * c0 + c1 * c1));             // (:46)
}

}  // namespace

extern "C" int loftr_epipolar_errors(const float* mkpts0_f, const float* mkpts1_f, const long* m_bids, const float* T_0to1,
                                     const float* K0, const float* K1, long M, int N, float* epi_errs, void* stream) {
  LOFTR_CHECK_ARG(M >= 0 && N >= 0);
  if (M == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(mkpts0_f && mkpts1_f && m_bids && T_0to1 && K0 && K1 && epi_errs && N > 0);
  if ((M + 255) / 256 >= (1L << 31)) return LOFTR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(epipolar_errors_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mkpts0_f,
                     mkpts1_f, m_bids, T_0to1, K0, K1, M, N, epi_errs);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
