// Bottleneck isolation for the split-fp16 GEMM main loop (csrc/gemm.h): the same kernel built with
//   -DVARIANT=0  full             -DVARIANT=1  no MFMA (loads + LDS traffic only)
//   -DVARIANT=2  loads hit ONE k-tile (all L2/LDS hits, no HBM streaming)      -DVARIANT=3 = 1 + 2
#ifndef VARIANT
#define VARIANT 0
#endif
#if VARIANT == 1 || VARIANT == 3
#define GEMM_PROBE_MFMA 0
#endif
#if VARIANT == 2 || VARIANT == 3
#define GEMM_PROBE_K(k) (0)
#endif
#include "../../loftr_amd/csrc/gemm.h"
#include <stdio.h>
#include <vector>
using Cfg = GemmCfg<128, 128, 2, 2>;
__global__ __launch_bounds__(Cfg::THREADS, 2) void probe(const sp_t* a, const sp_t* b, float* out, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  int tm, tn;
  if (!xcd_tile(ceil_div(M, Cfg::BM), ceil_div(N, Cfg::BN), tm, tn)) return;
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg>(asrc_plain(a, K), b, K, M, N, K, m0, n0, lds, acc);
  for (int i = 0; i < Cfg::TM; ++i) for (int j = 0; j < Cfg::TN; ++j) for (int r = 0; r < 16; ++r)
    out[(long)acc_row<Cfg>(m0, i, r) * N + acc_col<Cfg>(n0, j)] = acc[i][j][r];
}
int main() {
  const int M = 76800, N = 512, K = 512;
  sp_t *a, *b; float* o;
  hipMalloc(&a, (size_t)M * K * 4); hipMalloc(&b, (size_t)N * K * 4); hipMalloc(&o, (size_t)M * N * 4);
  hipMemset(a, 0, (size_t)M * K * 4); hipMemset(b, 0, (size_t)N * K * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(xcd_grid(ceil_div(M, Cfg::BM), ceil_div(N, Cfg::BN)));
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(probe, grid, dim3(Cfg::THREADS), 0, 0, a, b, o, M, N, K);
  hipEventRecord(e0);
  const int reps = 20;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(probe, grid, dim3(Cfg::THREADS), 0, 0, a, b, o, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("variant %d: %.1f us per launch, %.1f TFLOP/s fp32-equivalent\n", VARIANT, ms / reps * 1e3, 2.0 * M * N * K / (ms / reps * 1e-3) / 1e12);
  return 0;
}
