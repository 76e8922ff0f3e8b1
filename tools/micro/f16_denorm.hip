// Micro-probe: does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (needed by the 2-term fp16
// split of fp32 operands)?  Also checks v_cvt_pk rounding behaviour on overflow.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(float a_val, float b_val, float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  // A[i][k]: lane (i = lane&31, g = lane>>5) holds k = g*8..g*8+7
  if ((threadIdx.x >> 5) == 0) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
  f16v acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void cvt(float x, float y, float* out) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  f2 v = {x, y};
  h2 h = __builtin_convertvector(v, h2);
  out[0] = (float)h[0]; out[1] = (float)h[1];
  auto r = __builtin_amdgcn_cvt_pkrtz(x, y);
  out[2] = (float)r[0]; out[3] = (float)r[1];
}
int main() {
  float* d; hipMalloc(&d, 64);
  float h[4];
  float tests[][2] = {{1.f, 1.f}, {3.0e-6f, 1.f}, {3.0e-6f, 1024.f}, {6.0e-8f, 1.f}, {5.0e-5f, 5.0e-5f}, {1.0e-3f, 3e-5f}};
  for (auto& t : tests) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, t[0], t[1], d);
    hipMemcpy(h, d, 4, hipMemcpyDeviceToHost);
    printf("mfma f16: a=%g b=%g -> %.9g (expect %.9g)\n", t[0], t[1], h[0], (float)(_Float16)t[0] * (float)(_Float16)t[1]);
  }
  float c[][2] = {{1.0009765f, 70000.f}, {-1e6f, 6.1e-5f}, {1.00048828125f, 1.00146484375f}};
  for (auto& t : c) {
    hipLaunchKernelGGL(cvt, dim3(1), dim3(1), 0, 0, t[0], t[1], d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("cvt: x=%.10g y=%.10g  rn=(%.10g, %.10g) rtz=(%.10g, %.10g)\n", t[0], t[1], h[0], h[1], h[2], h[3]);
  }
  return 0;
}
