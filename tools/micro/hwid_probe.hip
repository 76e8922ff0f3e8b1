// Which SIMD does wave w of a 512-thread workgroup run on?  (The sweep kernel's early / late wave split assumes that waves
// w and w + 4 share a SIMD.)   hipcc --offload-arch=gfx950 -O2 -o tools/micro/hwid_probe tools/micro/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void probe(unsigned* out) {
  __shared__ char big[140 * 1024];                    // one workgroup per CU, as the sweep
  big[threadIdx.x] = 0;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; unsigned h[64];
  hipMalloc(&d, sizeof(h));
  probe<<<8, 512>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 8; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) printf("  w%d simd=%u wave=%u cu=%u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
    printf("\n");
  }
  return 0;
}
