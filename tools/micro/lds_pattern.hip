// ds_read_b128 throughput per CU for candidate fragment-read address patterns (which layouts are bank-conflict
// free for the 64 lanes of one MFMA operand read?).  Lane (tx = lane & 31, g = lane >> 5) reads 16 B.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_pattern.hip -o tools/micro/lds_pattern && tools/micro/lds_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ int pattern_addr(int pat, int lane, int wave) {
  const int tx = lane & 31, g = lane >> 5;
  const int p = tx + wave * 34;                  // pixel / row index (waves read different rows)
  switch (pat) {
    case 0: return p * 128 + ((g ^ ((p >> 1) & 7)) << 4);            // current: 128-B rows, slot c ^ (r>>1)&7
    case 1: return p * 128 + (g << 4);                               // 128-B rows, no swizzle
    case 2: return p * 64 + ((g ^ ((p >> 1) & 3)) << 4);             // 64-B rows, slot c ^ (p>>1)&3
    case 3: return p * 64 + ((g ^ (p & 3)) << 4);                    // 64-B rows, slot c ^ p&3
    case 4: return p * 64 + (g << 4);                                // 64-B rows, no swizzle
    case 5: return p * 64 + ((g ^ ((p >> 2) & 3)) << 4);             // 64-B rows, slot c ^ (p>>2)&3
    case 6: return p * 128 + ((g ^ (p & 7)) << 4);                   // 128-B rows, slot c ^ r&7
    case 7: return p * 32 + (g << 4);                                // 32-B rows (two chunks), linear
    case 8: return (p * 64 + ((g ^ ((p >> 1) & 3)) << 4)) ^ 32;      // pattern 2, other chunk pair
    case 9: return p * 80 + (g << 4);                                // 64-B rows padded to 80 B
    default: return lane * 16;                                       // fully linear
  }
}
__global__ __launch_bounds__(512) void k(unsigned* out, int iters, int pat) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const int addr = pattern_addr(pat, threadIdx.x & 63, threadIdx.x >> 6) & (64 * 1024 - 16);
  typedef __attribute__((address_space(3))) char* lp;
  const unsigned a = (unsigned)(size_t)(lp)(lds) + addr;
  u4 s = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u4 v0, v1, v2, v3;
    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4352\n ds_read_b128 %2, %4 offset:8704\n ds_read_b128 %3, %4 offset:13056\n s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a));
    s ^= v0 ^ v1 ^ v2 ^ v3;
  }
  if ((s[0] ^ s[1] ^ s[2] ^ s[3]) == 0x12345u) out[threadIdx.x] = s[0];
}
int main() {
  unsigned* d; hipMalloc(&d, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"128B rows, c^(r>>1)&7 (current)", "128B rows, linear", "64B rows, c^(p>>1)&3", "64B rows, c^p&3", "64B rows, linear",
                         "64B rows, c^(p>>2)&3", "128B rows, c^r&7", "32B rows linear", "64B rows c^(p>>1)&3 ^32", "80B padded rows", "lane*16 linear"};
  for (int pat = 0; pat <= 10; ++pat) {
    const int iters = 20000, blocks = 256, threads = 512;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 10, pat);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, pat);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)(threads / 64) * iters * 4 * 1024;
    printf("pat %2d %-34s %.3f ms  %.1f GB/s per CU  (%.1f B/clk @2.4GHz)\n", pat, names[pat], ms, bytes_per_cu / ms / 1e6, bytes_per_cu / ms / 1e6 / 2.4);
  }
  return 0;
}
