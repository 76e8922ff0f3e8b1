// Layout check of v_mfma_f32_4x4x4_16B_f16 (16 independent 4x4x4 blocks per wave) on gfx950, and its issue rate.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/mfma4_probe tools/micro/mfma4_probe.hip && tools/micro/mfma4_probe
// Assumed: lane l = (block b = l / 4, index q = l % 4).  A: lane holds A_b[i = q][k = 0..3];  B: lane holds B_b[k = 0..3][j = q];
// D: lane holds D_b[i = 0..3][j = q] in its four registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {      // A [16][4][4] (b, i, k), B [16][4][4] (b, k, j), D [16][4][4] (b, i, j)
  const int l = threadIdx.x, b = l / 4, q = l % 4;
  h4 a, bb;
  for (int kk = 0; kk < 4; ++kk) { a[kk] = (_Float16)A[(b * 4 + q) * 4 + kk]; bb[kk] = (_Float16)B[(b * 4 + kk) * 4 + q]; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, bb, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(b * 4 + i) * 4 + q] = c[i];
}
__global__ void rate(float* out, int iters) {
  h4 a = {(_Float16)1.f, (_Float16)2.f, (_Float16)3.f, (_Float16)(float)threadIdx.x}, b = a;
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c3, 0, 0, 0);
  }
  out[threadIdx.x + blockIdx.x * 64] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  float hA[256], hB[256], hD[256], ref[256];
  for (int i = 0; i < 256; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
    float s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[(b * 4 + i) * 4 + kk] * hB[(b * 4 + kk) * 4 + j];
    ref[(b * 4 + i) * 4 + j] = s;
  }
  float *dA, *dB, *dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1 << 20);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(hD[i] - ref[i]));
  printf("layout check: max |D - ref| = %g (%s)\n", e, e == 0 ? "assumed layout is right" : "WRONG");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 100000;
  rate<<<1024, 64>>>(dD, 100); hipEventRecord(e0); rate<<<1024, 64>>>(dD, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("4x4x4_16B_f16: %.2f ns per instruction and wave (one wave per SIMD; 4 independent chains)\n", ms * 1e6 / (iters * 4.0));
  return 0;
}
