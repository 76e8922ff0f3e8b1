// fp32 -> SP ("split pair" fp16 hi/lo, gemm.h) conversion of whole row-major tensors.
#include "gemm.h"

namespace {
// Per-row scale of the jobs that ask for one: one wave per row, max |x| -> inv_scale[row] (gemm.h: sp_row_scale).
__global__ __launch_bounds__(256) void sp_rowscale_kernel(SpJobs jobs) {
  const int job = blockIdx.y;
  float* inv = jobs.inv_scale[job];
  if (!inv) return;
  const float* src = jobs.src[job];
  const int K = jobs.K[job], ld = jobs.ld[job], rows = jobs.rows[job];
  const int lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    float m = 0.f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(src[row * ld + k]));
    m = wave_max(m);
    float iv;
    (void)sp_row_scale(m, &iv);
    if (lane == 0) inv[row] = iv;
  }
}

// Per-tensor scale (loftr_sp_from_f32_scaled): max |x| of the whole tensor by integer atomicMax on the bit patterns of
// the absolute values (monotonic for non-negative floats), then one thread turns it into the inverse power-of-two scale.
__global__ __launch_bounds__(256) void tensor_absmax_kernel(const float* __restrict__ src, long n, int* __restrict__ bits) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(src[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(bits, __float_as_int(m));
}
__global__ void tensor_scale_finish_kernel(float* __restrict__ slot) {
  float inv;
  (void)sp_row_scale(__int_as_float(*reinterpret_cast<int*>(slot)), &inv);
  *slot = inv;
}

// Range guard (debug aid, loftr_hip_range_check_enable): flags a value stored UNSCALED whose magnitude is not below the fp16
// maximum (or is not finite) -- its hi half would be inf and every product it enters NaN, where the fp32 reference is fine.
__global__ __launch_bounds__(256) void sp_range_kernel(SpJobs jobs, int* __restrict__ flag) {
  const int job = blockIdx.y;
  if (jobs.inv_scale[job] || jobs.tensor_inv[job]) return;          // scaled operands cannot overflow
  const float* src = jobs.src[job];
  const int K = jobs.K[job], ld = jobs.ld[job];
  const long n = (long)jobs.rows[job] * K;
  bool bad = false;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long row = i / K;
    bad |= !(fabsf(src[row * ld + (i - row * K)]) < 65504.f);
  }
  if (bad) atomicOr(flag, 1);
}

// One thread per (row, channel octet): two 16-B loads of fp32, one 16-B store each of the hi and the lo chunk.
// Sources whose row pitch is not a multiple of 4 floats (or K not a multiple of 8) take the scalar tail path.
__global__ __launch_bounds__(256) void sp_convert_kernel(SpJobs jobs) {
  const int job = blockIdx.y;
  const float* src = jobs.src[job];
  sp_t* dst = jobs.dst[job];
  const int K = jobs.K[job], ld = jobs.ld[job];
  const int Kp = (K + 31) / 32 * 32;
  const int octs = Kp >> 3;
  const long nitems = (long)jobs.rows[job] * octs;
  const bool vec = (ld & 3) == 0 && (((size_t)src) & 15) == 0;
  const float* inv = jobs.inv_scale[job];
  const float tsc = jobs.tensor_inv[job] ? 1.f / *jobs.tensor_inv[job] : 1.f;      // exact power of two
  for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < nitems; item += (long)gridDim.x * 256) {
    const long row = item / octs;
    const int oct = (int)(item - row * octs);
    const int c0 = oct * 8;
    float v[8];
    const float* p = src + row * ld + c0;
    if (vec && c0 + 8 <= K) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p), b2 = *reinterpret_cast<const f32x4*>(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b2.x; v[5] = b2.y; v[6] = b2.z; v[7] = b2.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = c0 + e < K ? p[e] : 0.f;
    }
    if (inv) {                                    // exact: a power of two (sp_rowscale_kernel ran before)
      const float sc = 1.f / inv[row];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= sc;
    }
    if (jobs.tensor_inv[job]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= tsc;
    }
    u32x4 hi, lo;
    sp_pack8(v, hi, lo);
    sp_t* o = dst + row * Kp + sp_octet_off(oct);
    *reinterpret_cast<u32x4*>(o) = hi;
    *reinterpret_cast<u32x4*>(o + 16) = lo;
  }
}
}  // namespace

int launch_sp_convert(const SpJobs& jobs, hipStream_t st) {
  if (jobs.n <= 0) return LOFTR_OK;
  long maxg = 0;
  for (int i = 0; i < jobs.n; ++i) {
    const long g = (long)jobs.rows[i] * ((jobs.K[i] + 31) / 32) * 4;      // octets
    if (g > maxg) maxg = g;
  }
  if (maxg == 0) return LOFTR_OK;
  long bx = (maxg + 255) / 256;
  if (bx > 32768) bx = 32768;
  bool any_scale = false;
  int maxrows = 0;
  for (int i = 0; i < jobs.n; ++i)
    if (jobs.inv_scale[i]) { any_scale = true; if (jobs.rows[i] > maxrows) maxrows = jobs.rows[i]; }
  if (any_scale) {
    int gx = (maxrows + 3) / 4;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(sp_rowscale_kernel, dim3((unsigned)gx, jobs.n), dim3(256), 0, st, jobs);
  }
  hipLaunchKernelGGL(sp_convert_kernel, dim3((unsigned)bx, jobs.n), dim3(256), 0, st, jobs);
  LOFTR_CHECK_LAUNCH();
  if (g_loftr_range_check) {                 // debug: synchronous (one host-mapped flag word, allocated on first use)
    static int* flag = nullptr;
    if (!flag && hipHostMalloc(reinterpret_cast<void**>(&flag), sizeof(int), hipHostMallocMapped) != hipSuccess) return LOFTR_ERR_LAUNCH;
    *flag = 0;
    long cx = bx > 1024 ? 1024 : bx;
    hipLaunchKernelGGL(sp_range_kernel, dim3((unsigned)cx, jobs.n), dim3(256), 0, st, jobs, flag);
    if (hipStreamSynchronize(st) != hipSuccess) return LOFTR_ERR_LAUNCH;
    if (*flag) return LOFTR_ERR_RANGE;
  }
  return LOFTR_OK;
}

int launch_sp_convert1(const float* src, int ld, sp_t* dst, long rows, int K, hipStream_t st, float* tensor_inv_out) {
  if (rows <= 0) return LOFTR_OK;
  if (rows > 0x7fffffffL) return LOFTR_ERR_UNSUPPORTED;
  SpJobs j;
  if (tensor_inv_out) {                      // contiguous rows only (ld == K)
    if (ld != K) return LOFTR_ERR_UNSUPPORTED;
    (void)hipMemsetAsync(tensor_inv_out, 0, sizeof(float), st);
    long nb = (rows * K + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(tensor_absmax_kernel, dim3((unsigned)nb), dim3(256), 0, st, src, rows * (long)K, reinterpret_cast<int*>(tensor_inv_out));
    hipLaunchKernelGGL(tensor_scale_finish_kernel, dim3(1), dim3(1), 0, st, tensor_inv_out);
    j.tensor_inv[0] = tensor_inv_out;
  }
  j.n = 1; j.src[0] = src; j.dst[0] = dst; j.rows[0] = (int)rows; j.K[0] = K; j.ld[0] = ld;
  return launch_sp_convert(j, st);
}
