// Effective shader clock under load: s_memtime (shader cycles) against s_memrealtime (100 MHz) around
//   (a) a pure MFMA loop, (b) MFMA + ds_read_b128 traffic, (c) ds_read only, (d) idle-ish scalar loop.
// Answers "what is 100 % matrix-pipe utilisation in FLOP/s on this box while the kernel is running?"
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* clk, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = seed * i;
  __syncthreads();
  h8 a[2], b[2];
  for (int i = 0; i < 8; ++i) { a[0][i] = (_Float16)(seed + threadIdx.x * 0.013f + i); a[1][i] = (_Float16)(seed - threadIdx.x * 0.007f + i);
                                b[0][i] = (_Float16)(0.5f * seed + threadIdx.x * 0.003f - i); b[1][i] = (_Float16)(seed * 0.25f + i * 0.11f); }
  f16v acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const h8* lp = reinterpret_cast<const h8*>(lds) + (threadIdx.x & 63) + (threadIdx.x >> 6) * 64;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        h8 v = lp[u * 512];
        asm volatile("" : "+v"(v));
        if (MODE == 2) { a[u & 1] = v; }
        else b[u & 1] = v;
      }
    }
    if (MODE == 0 || MODE == 1) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], acc[3], 0, 0, 0);
      }
    }
    if (MODE == 3) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  s += (float)a[0][0] + (float)b[0][0];
  if (s == 1234.5f) out[threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 17) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
template <int MODE>
void run(const char* name, float* d, unsigned long long* c, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4, threads = 512;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, c, 10, 1.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, c, iters, 1.37f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
  const double mhz = (double)h[0] / ((double)h[1] / 100.0);           // cycles per microsecond
  const double flops = (MODE <= 1) ? (double)blocks * (threads / 64) * iters * 24 * 2.0 * 32 * 32 * 16 : 0.0;
  printf("%-28s %.2f ms  shader clock %.0f MHz  (memtime %llu, realtime %llu)  %.0f TFLOP/s\n", name, ms, mhz, h[0], h[1], flops / ms / 1e9);
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  unsigned long long* c; (void)hipMalloc(&c, 64);
  run<3>("scalar nops", d, c, 20000);
  run<0>("pure MFMA", d, c, 4000);
  run<1>("MFMA + ds_read_b128", d, c, 4000);
  run<2>("ds_read_b128 only", d, c, 20000);
  run<0>("pure MFMA (again)", d, c, 8000);
  return 0;
}
