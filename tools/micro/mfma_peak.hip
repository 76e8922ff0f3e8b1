// Practical ceiling of v_mfma_f32_32x32x16_f16 on this box: NW waves per SIMD, 4 independent accumulators,
// non-trivial operand data (power!), no memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  h8 a[2], b[2];
  for (int i = 0; i < 8; ++i) { a[0][i] = (_Float16)(seed + threadIdx.x * 0.013f + i); a[1][i] = (_Float16)(seed - threadIdx.x * 0.007f + i);
                                b[0][i] = (_Float16)(0.5f * seed + threadIdx.x * 0.003f - i); b[1][i] = (_Float16)(seed * 0.25f + i * 0.11f); }
  f16v acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 1234.5f) out[threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512}) {
    const int iters = 4000, blocks = 256 * (threads == 256 ? 2 : 1) * 4;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.37f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * (threads / 64) * iters * 24 * 2.0 * 32 * 32 * 16;
    printf("%d threads/block x %d blocks: %.2f ms, %.0f TFLOP/s fp16 MFMA (random-ish data)\n", threads, blocks, ms, flops / ms / 1e9);
  }
  return 0;
}
