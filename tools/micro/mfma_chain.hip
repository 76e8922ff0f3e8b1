// Dependent-accumulator latency of v_mfma_f32_32x32x16_f16 with ONE wave per SIMD: NCH independent accumulator chains,
// round-robin.  Prints ns and shader cycles (s_memtime) per MFMA for NCH = 1..4.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NCH>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters, float seed) {
  h8 a[2], b[2];
  for (int i = 0; i < 8; ++i) { a[0][i] = (_Float16)(seed + threadIdx.x * 0.013f + i); a[1][i] = (_Float16)(seed - threadIdx.x * 0.007f + i);
                                b[0][i] = (_Float16)(0.5f * seed + threadIdx.x * 0.003f - i); b[1][i] = (_Float16)(seed * 0.25f + i * 0.11f); }
  f16v acc[NCH];
  for (int t = 0; t < NCH; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) acc[u % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 1], b[(u >> 1) & 1], acc[u % NCH], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < NCH; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 1234.5f) out[threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH> void run(float* d, long long* c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, blocks = 256;
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, d, c, 10, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, d, c, iters, 1.37f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 12;
  printf("chains %d: %.2f ns per MFMA per SIMD, %.1f memtime ticks (100 MHz -> x10 ns) per MFMA, %.0f TFLOP/s chip\n", NCH, ms * 1e6 / n,
         (double)h / n, (double)blocks * 4 * n * 2.0 * 32 * 32 * 16 / ms / 1e9);
}
int main() {
  float* d; long long* c; hipMalloc(&d, 4096); hipMalloc(&c, 64);
  run<1>(d, c); run<2>(d, c); run<3>(d, c); run<4>(d, c);
  return 0;
}
