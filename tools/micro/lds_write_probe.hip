// Do register -> LDS stores work for the SECOND workgroup of a CU, whose LDS window lies above / across 64 KB?  (head_grad_kernel's
// co-residency fault: the workgroup with the higher LDS base computes with partly missing tile data.)  Every workgroup fills its 48 KB
// with a per-thread pattern through ds_write_b64 / ds_write_b128 / ds_write2_b32, barrier, reads everything back with ds_read_b128
// and plain reads, counts mismatches; repeated with the roles of the threads rotated.  2 x 48 KB -> two workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_write_probe.hip -o tools/micro/lds_write_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr int BYTES = 48 * 1024;
template <int MODE, int PAD_KB>
__global__ __launch_bounds__(256, PAD_KB ? 1 : 2) void probe(int rounds, unsigned* bad) {
  __shared__ __attribute__((aligned(16))) char lds[BYTES + PAD_KB * 1024];
  const int t = threadIdx.x;
  unsigned errs = 0;
  for (int r = 0; r < rounds; ++r) {
    const uint32_t tag = (uint32_t)(r * 2654435761u) ^ (blockIdx.x * 97u);
    // 48 KB = 256 threads x 192 B: thread t owns bytes [192 p, 192 p + 192) with p = (t + r) % 256
    const int p = (t + r) & 255;
    char* base = lds + p * 192;
    if (MODE == 0) {
      for (int i = 0; i < 12; ++i) *reinterpret_cast<u4*>(base + 16 * i) = u4{tag + i, tag ^ i, tag + p, (uint32_t)p};
    } else if (MODE == 1) {
      for (int i = 0; i < 24; ++i) *reinterpret_cast<uint2*>(base + 8 * i) = uint2{(i & 1) ? tag + p : tag + i / 2, (i & 1) ? (uint32_t)p : (tag ^ (i / 2))};
    } else {
      for (int i = 0; i < 48; ++i) *reinterpret_cast<volatile uint32_t*>(base + 4 * i) = (i & 3) == 0 ? tag + i / 4 : (i & 3) == 1 ? (tag ^ (i / 4)) : (i & 3) == 2 ? tag + p : (uint32_t)p;
    }
    __syncthreads();
    const int q = (t * 7 + r) & 255;                       // read somebody else's block
    const char* rb = lds + q * 192;
    for (int i = 0; i < 12; ++i) {
      const u4 v = *reinterpret_cast<const u4*>(rb + 16 * i);
      errs += (v[0] != tag + i) + (v[1] != (tag ^ i)) + (v[2] != tag + q) + (v[3] != (uint32_t)q);
    }
    __syncthreads();
  }
  if (errs) atomicAdd(bad, errs);
}
template <int MODE, int PAD_KB>
void run(const char* name, int wgs) {
  unsigned* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL((probe<MODE, PAD_KB>), dim3(wgs), dim3(256), 0, 0, 500, bad);
  unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("%-58s %5d workgroups: %u mismatching dwords\n", name, wgs, h);
}
int main() {
  for (int wgs : {256, 512, 1024}) {
    run<0, 0>("ds_write_b128, two workgroups per CU", wgs);
    run<1, 0>("ds_write_b64,  two workgroups per CU", wgs);
    run<2, 0>("ds_write_b32,  two workgroups per CU", wgs);
    run<1, 40>("ds_write_b64,  one workgroup per CU (88 KB)", wgs);
  }
  return 0;
}
