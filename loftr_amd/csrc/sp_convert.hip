// fp32 -> SP ("split pair" fp16 hi/lo, gemm.h) conversion of whole row-major tensors.
#include "gemm.h"

namespace {
// one wave handles two 32-column groups of one row per iteration: lane l -> column 2-group layout
//   thread t of a 256-thread block: group-in-block = t / 32, column c = t % 32
// Each half-wave converts one 128-B group: reads 32 floats (128 B, coalesced), writes 32 dwords.
__global__ __launch_bounds__(256) void sp_convert_kernel(SpJobs jobs) {
  const int job = blockIdx.y;
  const float* src = jobs.src[job];
  sp_t* dst = jobs.dst[job];
  const int K = jobs.K[job], ld = jobs.ld[job];
  const int Kp = (K + 31) / 32 * 32;
  const int gpr = Kp / 32;                                  // groups per row
  const long ngroups = (long)jobs.rows[job] * gpr;
  const int c = threadIdx.x & 31;
  for (long gidx = (long)blockIdx.x * 8 + (threadIdx.x >> 5); gidx < ngroups; gidx += (long)gridDim.x * 8) {
    const long row = gidx / gpr;
    const int grp = (int)(gidx - row * gpr);
    const int col = grp * 32 + c;
    const float v = col < K ? src[row * ld + col] : 0.f;
    sp_store(dst + row * Kp, col, v, true);
  }
}
}  // namespace

int launch_sp_convert(const SpJobs& jobs, hipStream_t st) {
  if (jobs.n <= 0) return LOFTR_OK;
  long maxg = 0;
  for (int i = 0; i < jobs.n; ++i) {
    const long g = (long)jobs.rows[i] * ((jobs.K[i] + 31) / 32);
    if (g > maxg) maxg = g;
  }
  if (maxg == 0) return LOFTR_OK;
  long bx = (maxg + 7) / 8;
  if (bx > 16384) bx = 16384;
  hipLaunchKernelGGL(sp_convert_kernel, dim3((unsigned)bx, jobs.n), dim3(256), 0, st, jobs);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

int launch_sp_convert1(const float* src, int ld, sp_t* dst, long rows, int K, hipStream_t st) {
  if (rows <= 0) return LOFTR_OK;
  if (rows > 0x7fffffffL) return LOFTR_ERR_UNSUPPORTED;
  SpJobs j;
  j.n = 1; j.src[0] = src; j.dst[0] = dst; j.rows[0] = (int)rows; j.K[0] = K; j.ld[0] = ld;
  return launch_sp_convert(j, st);
}
