// Does s_barrier hold when two 4-wave workgroups share a CU?  (head_grad_kernel's co-residency fault looks like waves passing a barrier
// early: stale per-wave maxima, partly stale LDS tiles.)  Each round: wave w writes (round, w) to its LDS slot, barrier, every lane
// reads all four slots and counts the ones that do not carry this round, barrier.  Some waves are slowed by a data-dependent spin so
// that a barrier that releases early is seen.  LDS footprint 48 KB -> two or three workgroups per CU; 90 KB -> one.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/barrier_probe.hip -o tools/micro/barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS_KB, int MINWG>
__global__ __launch_bounds__(256, MINWG) void probe(int rounds, unsigned* bad, float* sink) {
  __shared__ int slot[4];
  __shared__ char pad[LDS_KB * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned errs = 0;
  float x = (float)threadIdx.x;
  if (threadIdx.x == 0) pad[blockIdx.x % (LDS_KB * 1024)] = 1;
  for (int r = 1; r <= rounds; ++r) {
    const int spin = ((r * 7 + wave * 13 + blockIdx.x) & 3) == 0 ? 400 : 10;      // one wave in four is late, a different one every round
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    if (lane == 0) slot[wave] = r;
    __syncthreads();
    for (int w = 0; w < 4; ++w) errs += slot[w] != r;
    __syncthreads();
  }
  if (errs) atomicAdd(bad, errs);
  if (x == 123.456f) *sink = x + pad[7];
}
template <int LDS_KB, int MINWG>
void run(const char* name, int wgs) {
  unsigned* bad; float* sink;
  hipMalloc(&bad, 4); hipMalloc(&sink, 4); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL((probe<LDS_KB, MINWG>), dim3(wgs), dim3(256), 0, 0, 2000, bad, sink);
  unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("%-46s %5d workgroups: %u stale reads\n", name, wgs, h);
}
int main() {
  for (int wgs : {256, 304, 512, 1024}) {
    run<48, 2>("48 KB LDS, launch_bounds(256, 2)", wgs);
    run<90, 1>("90 KB LDS, launch_bounds(256, 1)", wgs);
  }
  return 0;
}
