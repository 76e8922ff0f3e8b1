// Training-mode glue of the ResNet-FPN backbone on the device (round 5): what sits between the convolutions of a
// training step and was PyTorch autograd until round 4.
//   reference: src/loftr/backbone/resnet_fpn.py:22-40 (BasicBlock: conv - bn - relu - conv - bn - (+ x) - relu), :66-77 (_fuse_head: LeakyReLU),
//              :110-116 (F.interpolate(scale_factor=2., mode='bilinear', align_corners=True) + add); nn.BatchNorm2d in .train() mode:
//              batch statistics, biased variance for the normalisation, unbiased for the running estimate (train.py:108: the reference
//              trains with SyncBatchNorm, which is the same arithmetic over the ranks' union -- one process here).
// Tensors are NCHW fp32, as the training path's autograd graph carries them.  None of this is on the inference path; the kernels are
// plain HBM-bound vector code: every reduction is a two-stage sum with float64 partials merged in a fixed order (deterministic).
#include "common.h"

namespace {
namespace tg {
constexpr int RED_THREADS = 256;
constexpr int CHUNK = 8192;                 // elements of one (n, c) plane range per partial

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < RED_THREADS / 64; ++w) s += sh[w];      // fixed order
  return s;
}

// partial[c][p][0..1] = (sum x, sum x^2) over plane chunk p of channel c (p = n * chunks_per_plane + k)
__global__ __launch_bounds__(RED_THREADS) void bn_stats_kernel(const float* __restrict__ x, int C, long HW, int cpp, double* __restrict__ part) {
  __shared__ double sh[RED_THREADS / 64];
  const int c = blockIdx.x, p = blockIdx.y, n = p / cpp, k = p - n * cpp;
  const float* src = x + ((long)n * C + c) * HW;
  const long i0 = (long)k * CHUNK, i1 = min(HW, i0 + CHUNK);
  double s = 0.0, q = 0.0;
  for (long i = i0 + threadIdx.x; i < i1; i += RED_THREADS) { const double v = src[i]; s += v; q += v * v; }
  s = block_sum(s, sh);
  q = block_sum(q, sh);
  if (threadIdx.x == 0) { part[((long)c * gridDim.y + p) * 2] = s; part[((long)c * gridDim.y + p) * 2 + 1] = q; }
}
// mean, invstd (biased variance: what normalises), unbiased variance (what the running estimate takes); one thread per channel
__global__ void bn_finalize_kernel(const double* __restrict__ part, int C, int P, double count, float eps, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ var_unbiased) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int p = 0; p < P; ++p) { s += part[((long)c * P + p) * 2]; q += part[((long)c * P + p) * 2 + 1]; }
  const double m = s / count;
  double var = q / count - m * m;
  var = var > 0.0 ? var : 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (var_unbiased) var_unbiased[c] = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
}
// y = (x - mean) * invstd * gamma + beta
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int C, long HW, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y) {
  const long plane = blockIdx.y;                         // n * C + c
  const int c = (int)(plane % C);
  // (x - mean) FIRST: exact for values near the mean.  The folded form x * a + (beta - mean * a) loses |mean| / std digits to cancellation --
  // 1e-5 relative on channels whose mean is ~100 standard deviations, which the ill-conditioned early-layer gradients of a training step
  // amplify to 1e-2 (found with the hipglue variant of test_training_step_full_backward_against_reference: layer1.1.bn1.bias 48 x the
  // reference's own float32 noise with the folded form)
  const float a = invstd[c] * (gamma ? gamma[c] : 1.f), b = beta ? beta[c] : 0.f, m = mean[c];
  const float* src = x + plane * HW;
  float* dst = y + plane * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) dst[i] = fmaf(src[i] - m, a, b);
}
// backward, stage 1: partial[c][p] = (sum dy, sum dy * xhat)
__global__ __launch_bounds__(RED_THREADS) void bn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x, int C, long HW,
                                                                  int cpp, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  double* __restrict__ part) {
  __shared__ double sh[RED_THREADS / 64];
  const int c = blockIdx.x, p = blockIdx.y, n = p / cpp, k = p - n * cpp;
  const long base = ((long)n * C + c) * HW;
  const long i0 = (long)k * CHUNK, i1 = min(HW, i0 + CHUNK);
  const float m = mean[c], is = invstd[c];
  double s = 0.0, q = 0.0;
  for (long i = i0 + threadIdx.x; i < i1; i += RED_THREADS) {
    const double g = dy[base + i];
    s += g; q += g * (double)((x[base + i] - m) * is);
  }
  s = block_sum(s, sh);
  q = block_sum(q, sh);
  if (threadIdx.x == 0) { part[((long)c * gridDim.y + p) * 2] = s; part[((long)c * gridDim.y + p) * 2 + 1] = q; }
}
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, int C, int P, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int p = 0; p < P; ++p) { s += part[((long)c * P + p) * 2]; q += part[((long)c * P + p) * 2 + 1]; }
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
}
// dx = gamma * invstd * (dy - dbeta / m - xhat * dgamma / m)        (batch statistics: the mean and the variance depend on x)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, int C, long HW,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, float inv_count, float* __restrict__ dx) {
  const long plane = blockIdx.y;
  const int c = (int)(plane % C);
  const float m = mean[c], is = invstd[c], g = (gamma ? gamma[c] : 1.f) * is;
  const float kb = dbeta[c] * inv_count, kg = dgamma[c] * inv_count;
  const long base = plane * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
    const float xh = (x[base + i] - m) * is;
    dx[base + i] = g * (dy[base + i] - kb - xh * kg);
  }
}

// y = act(a [+ b]): act 1 = ReLU, 2 = LeakyReLU(slope); 0 = none.  backward: dx = dy * (y > 0 ? 1 : slope') -- for ReLU and LeakyReLU with a
// positive slope the sign of the output is the sign of the input, so the (in-place) forward result is all the backward needs.
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, int act, float slope,
                                                      float* __restrict__ y) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a + i);
      if (b) v += *reinterpret_cast<const f32x4*>(b + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act == 1 ? fmaxf(v[e], 0.f) : (act == 2 ? (v[e] > 0.f ? v[e] : slope * v[e]) : v[e]);
      *reinterpret_cast<f32x4*>(y + i) = v;
    } else {
      for (long j = i; j < n; ++j) {
        float v = a[j] + (b ? b[j] : 0.f);
        y[j] = act == 1 ? fmaxf(v, 0.f) : (act == 2 ? (v > 0.f ? v : slope * v) : v);
      }
    }
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long n, int act, float slope,
                                                      float* __restrict__ dx) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = dy[i];
    dx[i] = act == 0 ? g : (y[i] > 0.f ? g : (act == 1 ? 0.f : slope * g));
  }
}

// F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True): out [.., 2H, 2W]; src = dst * (in - 1) / (out - 1)   (torch's
// area_pixel_compute_scale / upsample_bilinear2d: the scale is (in - 1) / (out - 1) as a float, the index its truncation)
__global__ __launch_bounds__(256) void up2_fwd_kernel(const float* __restrict__ x, long planes, int H, int W, float sy, float sx,
                                                      float* __restrict__ y) {
#pragma clang fp contract(off)      // (HIP's __fmul_rn is an ordinary multiply: only the pragma keeps it out of an fma)
  const int Ho = 2 * H, Wo = 2 * W;
  const long total = planes * Ho * Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    const long t = i / Wo;
    const int yo = (int)(t % Ho);
    const long pl = t / Ho;
    // every product and sum rounded on its own (no fma contraction): torch's kernels -- CPU and GPU agree bit for bit -- evaluate
    // src = scale * dst, lambda = src - floor(src) and the interpolation in separately rounded float operations
    const float fy = __fmul_rn(sy, (float)yo), fx = __fmul_rn(sx, (float)xo);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* p = x + pl * H * W;
    const float top = __fadd_rn(__fmul_rn(hx, p[y0 * W + x0]), __fmul_rn(lx, p[y0 * W + x1]));
    const float bot = __fadd_rn(__fmul_rn(hx, p[y1 * W + x0]), __fmul_rn(lx, p[y1 * W + x1]));
    y[i] = __fadd_rn(__fmul_rn(hy, top), __fmul_rn(ly, bot));
  }
}
// The adjoint as a GATHER (no atomics, fixed order): input pixel (yi, xi) collects dy of every output pixel whose (y0, y1) x (x0, x1)
// footprint contains it, with the weight the forward used.  Output rows whose source row y0 equals yi or yi - 1 lie in a window of at most
// five rows around yi / sy; each candidate is tested with the forward's own index arithmetic.
__global__ __launch_bounds__(256) void up2_bwd_kernel(const float* __restrict__ dy, long planes, int H, int W, float sy, float sx,
                                                      float* __restrict__ dx) {
#pragma clang fp contract(off)
  const int Ho = 2 * H, Wo = 2 * W;
  const long total = planes * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xi = (int)(i % W);
    const long t = i / W;
    const int yi = (int)(t % H);
    const long pl = t / H;
    const float* g = dy + pl * (long)Ho * Wo;
    const int yc = sy > 0.f ? (int)((float)yi / sy) : 0, xc = sx > 0.f ? (int)((float)xi / sx) : 0;
    float acc = 0.f;
    for (int yo = max(0, yc - 3); yo <= min(Ho - 1, yc + 3); ++yo) {
      const float fy = __fmul_rn(sy, (float)yo);
      const int y0 = (int)fy, y1 = y0 + (y0 < H - 1);
      const float ly = fy - (float)y0, hy = 1.f - ly;
      float wy = 0.f;
      if (y0 == yi) wy += hy;
      if (y1 == yi) wy += ly;
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int xo = max(0, xc - 3); xo <= min(Wo - 1, xc + 3); ++xo) {
        const float fx = __fmul_rn(sx, (float)xo);
        const int x0 = (int)fx, x1 = x0 + (x0 < W - 1);
        const float lx = fx - (float)x0, hx = 1.f - lx;
        float wx = 0.f;
        if (x0 == xi) wx += hx;
        if (x1 == xi) wx += lx;
        if (wx != 0.f) row = fmaf(wx, g[(long)yo * Wo + xo], row);
      }
      acc = fmaf(wy, row, acc);
    }
    dx[i] = acc;
  }
}

// ---- channels-last (NHWC memory) variants: what the convolution nodes of the training path produce and consume (autograd.conv2d works on
// NHWC tensors), so no layout copy sits between a convolution and its BatchNorm / activation.  A block owns PIX_CHUNK pixels x all channels;
// a thread owns one group of four channels (16-B accesses, consecutive threads = consecutive channels: coalesced) and every R-th pixel.
constexpr int PIX_CHUNK = 512;
// partial[chunk][c][0..1]
template <bool BWD>
__global__ __launch_bounds__(RED_THREADS) void bn_stats_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ dy, long P, int C,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   double* __restrict__ part) {
  __shared__ double sh[RED_THREADS][8];
  const int G = C >> 2, R = RED_THREADS / G;                 // channel groups, pixel rows per sweep
  const int grp = threadIdx.x % G, r = threadIdx.x / G;
  const long p0 = (long)blockIdx.x * PIX_CHUNK, p1 = min(P, p0 + PIX_CHUNK);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (r < R) {
    f32x4 m4 = {0, 0, 0, 0}, is4 = {0, 0, 0, 0};
    if (BWD) { m4 = *reinterpret_cast<const f32x4*>(mean + 4 * grp); is4 = *reinterpret_cast<const f32x4*>(invstd + 4 * grp); }
    for (long p = p0 + r; p < p1; p += R) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * C + 4 * grp);
      if (!BWD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const double d = v[e]; s[e] += d; q[e] += d * d; }
      } else {
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + p * C + 4 * grp);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const double d = g[e]; s[e] += d; q[e] += d * (double)((v[e] - m4[e]) * is4[e]); }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { sh[threadIdx.x][e] = s[e]; sh[threadIdx.x][4 + e] = q[e]; }
  __syncthreads();
  if (threadIdx.x < G) {                                     // fixed order over the rows
    double ts[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
    for (int rr = 0; rr < R; ++rr)
#pragma unroll
      for (int e = 0; e < 4; ++e) { ts[e] += sh[rr * G + grp][e]; tq[e] += sh[rr * G + grp][4 + e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      part[((long)blockIdx.x * C + 4 * grp + e) * 2] = ts[e];
      part[((long)blockIdx.x * C + 4 * grp + e) * 2 + 1] = tq[e];
    }
  }
}
// the partials of the NHWC kernels are [chunk][c]: finalize with a channel stride of 1 and a partial stride of C
__global__ void bn_finalize_nhwc_kernel(const double* __restrict__ part, int C, int P, double count, float eps, float* __restrict__ mean,
                                        float* __restrict__ invstd, float* __restrict__ var_unbiased) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int p = 0; p < P; ++p) { s += part[((long)p * C + c) * 2]; q += part[((long)p * C + c) * 2 + 1]; }
  const double m = s / count;
  double var = q / count - m * m;
  var = var > 0.0 ? var : 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (var_unbiased) var_unbiased[c] = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
}
__global__ void bn_bwd_finalize_nhwc_kernel(const double* __restrict__ part, int C, int P, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int p = 0; p < P; ++p) { s += part[((long)p * C + c) * 2]; q += part[((long)p * C + c) * 2 + 1]; }
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
}
__global__ __launch_bounds__(256) void bn_apply_nhwc_kernel(const float* __restrict__ x, long P, int C, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y) {
  const int G = C >> 2;
  const long n4 = P * G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % G) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaf(v[e] - mean[c + e], invstd[c + e] * (gamma ? gamma[c + e] : 1.f), beta ? beta[c + e] : 0.f);
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_nhwc_kernel(const float* __restrict__ dy, const float* __restrict__ x, long P, int C,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                                const float* __restrict__ dbeta, float inv_count, float* __restrict__ dx) {
  const int G = C >> 2;
  const long n4 = P * G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % G) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4), g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float is = invstd[c + e], xh = (v[e] - mean[c + e]) * is;
      o[e] = (gamma ? gamma[c + e] : 1.f) * is * (g[e] - dbeta[c + e] * inv_count - xh * (dgamma[c + e] * inv_count));
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
  }
}
// bilinear x2, NHWC: one thread per (output pixel, four channels)
__global__ __launch_bounds__(256) void up2_fwd_nhwc_kernel(const float* __restrict__ x, long N, int H, int W, int C, float sy, float sx,
                                                           float* __restrict__ y) {
#pragma clang fp contract(off)
  const int Ho = 2 * H, Wo = 2 * W, G = C >> 2;
  const long total = N * Ho * Wo * G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % G) * 4;
    long t = i / G;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const long n = t / Ho;
    const float fy = __fmul_rn(sy, (float)yo), fx = __fmul_rn(sx, (float)xo);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* p = x + n * (long)H * W * C + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + ((long)y0 * W + x0) * C), b = *reinterpret_cast<const f32x4*>(p + ((long)y0 * W + x1) * C);
    const f32x4 cc = *reinterpret_cast<const f32x4*>(p + ((long)y1 * W + x0) * C), d = *reinterpret_cast<const f32x4*>(p + ((long)y1 * W + x1) * C);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float top = __fadd_rn(__fmul_rn(hx, a[e]), __fmul_rn(lx, b[e])), bot = __fadd_rn(__fmul_rn(hx, cc[e]), __fmul_rn(lx, d[e]));
      o[e] = __fadd_rn(__fmul_rn(hy, top), __fmul_rn(ly, bot));
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}
__global__ __launch_bounds__(256) void up2_bwd_nhwc_kernel(const float* __restrict__ dy, long N, int H, int W, int C, float sy, float sx,
                                                           float* __restrict__ dx) {
#pragma clang fp contract(off)
  const int Ho = 2 * H, Wo = 2 * W, G = C >> 2;
  const long total = N * H * W * G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % G) * 4;
    long t = i / G;
    const int xi = (int)(t % W); t /= W;
    const int yi = (int)(t % H);
    const long n = t / H;
    const float* g = dy + n * (long)Ho * Wo * C + c;
    const int yc = sy > 0.f ? (int)((float)yi / sy) : 0, xc = sx > 0.f ? (int)((float)xi / sx) : 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int yo = max(0, yc - 3); yo <= min(Ho - 1, yc + 3); ++yo) {
      const float fy = __fmul_rn(sy, (float)yo);
      const int y0 = (int)fy, y1 = y0 + (y0 < H - 1);
      const float ly = fy - (float)y0, hy = 1.f - ly;
      float wy = 0.f;
      if (y0 == yi) wy += hy;
      if (y1 == yi) wy += ly;
      if (wy == 0.f) continue;
      f32x4 row = {0.f, 0.f, 0.f, 0.f};
      for (int xo = max(0, xc - 3); xo <= min(Wo - 1, xc + 3); ++xo) {
        const float fx = __fmul_rn(sx, (float)xo);
        const int x0 = (int)fx, x1 = x0 + (x0 < W - 1);
        const float lx = fx - (float)x0, hx = 1.f - lx;
        float wx = 0.f;
        if (x0 == xi) wx += hx;
        if (x1 == xi) wx += lx;
        if (wx != 0.f) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(g + ((long)yo * Wo + xo) * C);
#pragma unroll
          for (int e = 0; e < 4; ++e) row[e] = fmaf(wx, v[e], row[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(wy, row[e], acc[e]);
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
  }
}
inline dim3 grid1d(long n, int per_thread = 1) {
  long b = (n + 256L * per_thread - 1) / (256L * per_thread);
  return dim3((unsigned)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b)));
}
}  // namespace tg
}  // namespace

// ---- C ABI (include/loftr_hip.h) ---------------------------------------------------------------------------------------------------------
extern "C" size_t loftr_bn_train_workspace_bytes(int N, int C, long HW) {           // covers both layouts
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  const long cpp = (HW + tg::CHUNK - 1) / tg::CHUNK;
  const long chunks = ((long)N * HW + tg::PIX_CHUNK - 1) / tg::PIX_CHUNK;
  const size_t a = (size_t)C * N * cpp * 2 * sizeof(double), b = (size_t)chunks * C * 2 * sizeof(double);
  return (a > b ? a : b) + 256;
}
static inline bool nhwc_ok(int C) { return C % 4 == 0 && C / 4 <= tg::RED_THREADS; }
extern "C" int loftr_bn_train_fwd(const float* x, int N, int C, long HW, int channels_last, const float* gamma, const float* beta, float eps, float* y,
                                  float* mean, float* invstd, float* var_unbiased, void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(x && y && mean && invstd && N >= 0 && C > 0 && HW > 0);
  if (N == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(ws != nullptr);
  if (ws_bytes < loftr_bn_train_workspace_bytes(N, C, HW)) return LOFTR_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (channels_last) {
    if (!nhwc_ok(C)) return LOFTR_ERR_UNSUPPORTED;
    const long P = (long)N * HW;
    const int chunks = (int)((P + tg::PIX_CHUNK - 1) / tg::PIX_CHUNK);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL((tg::bn_stats_nhwc_kernel<false>), dim3(chunks), dim3(tg::RED_THREADS), 0, st, x, nullptr, P, C, nullptr, nullptr, part);
    hipLaunchKernelGGL(tg::bn_finalize_nhwc_kernel, dim3((C + 63) / 64), dim3(64), 0, st, part, C, chunks, (double)P, eps, mean, invstd, var_unbiased);
    hipLaunchKernelGGL(tg::bn_apply_nhwc_kernel, tg::grid1d(P * (C / 4)), dim3(256), 0, st, x, P, C, mean, invstd, gamma, beta, y);
    LOFTR_CHECK_LAUNCH();
    return LOFTR_OK;
  }
  const int cpp = (int)((HW + tg::CHUNK - 1) / tg::CHUNK), P = N * cpp;
  if (P > 65535) return LOFTR_ERR_UNSUPPORTED;
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(tg::bn_stats_kernel, dim3(C, P), dim3(tg::RED_THREADS), 0, st, x, C, HW, cpp, part);
  hipLaunchKernelGGL(tg::bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, part, C, P, (double)N * (double)HW, eps, mean, invstd, var_unbiased);
  const long planes = (long)N * C;
  if (planes > 65535) return LOFTR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(tg::bn_apply_kernel, dim3((unsigned)min(64L, (HW + 255) / 256), (unsigned)planes), dim3(256), 0, st, x, C, HW, mean, invstd, gamma, beta, y);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
extern "C" int loftr_bn_train_bwd(const float* dy, const float* x, int N, int C, long HW, int channels_last, const float* mean, const float* invstd,
                                  const float* gamma, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(dy && x && mean && invstd && dx && dgamma && dbeta && N >= 0 && C > 0 && HW > 0);
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) { (void)hipMemsetAsync(dgamma, 0, sizeof(float) * C, st); (void)hipMemsetAsync(dbeta, 0, sizeof(float) * C, st); return LOFTR_OK; }
  LOFTR_CHECK_ARG(ws != nullptr);
  if (ws_bytes < loftr_bn_train_workspace_bytes(N, C, HW)) return LOFTR_ERR_WORKSPACE;
  if (channels_last) {
    if (!nhwc_ok(C)) return LOFTR_ERR_UNSUPPORTED;
    const long Pn = (long)N * HW;
    const int chunks = (int)((Pn + tg::PIX_CHUNK - 1) / tg::PIX_CHUNK);
    double* partn = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL((tg::bn_stats_nhwc_kernel<true>), dim3(chunks), dim3(tg::RED_THREADS), 0, st, x, dy, Pn, C, mean, invstd, partn);
    hipLaunchKernelGGL(tg::bn_bwd_finalize_nhwc_kernel, dim3((C + 63) / 64), dim3(64), 0, st, partn, C, chunks, dgamma, dbeta);
    hipLaunchKernelGGL(tg::bn_bwd_apply_nhwc_kernel, tg::grid1d(Pn * (C / 4)), dim3(256), 0, st, dy, x, Pn, C, mean, invstd, gamma, dgamma, dbeta,
                       (float)(1.0 / (double)Pn), dx);
    LOFTR_CHECK_LAUNCH();
    return LOFTR_OK;
  }
  const int cpp = (int)((HW + tg::CHUNK - 1) / tg::CHUNK), P = N * cpp;
  const long planes = (long)N * C;
  if (P > 65535 || planes > 65535) return LOFTR_ERR_UNSUPPORTED;
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(tg::bn_bwd_stats_kernel, dim3(C, P), dim3(tg::RED_THREADS), 0, st, dy, x, C, HW, cpp, mean, invstd, part);
  hipLaunchKernelGGL(tg::bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, part, C, P, dgamma, dbeta);
  hipLaunchKernelGGL(tg::bn_bwd_apply_kernel, dim3((unsigned)min(64L, (HW + 255) / 256), (unsigned)planes), dim3(256), 0, st, dy, x, C, HW, mean, invstd,
                     gamma, dgamma, dbeta, (float)(1.0 / ((double)N * (double)HW)), dx);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
extern "C" int loftr_act_fwd(const float* a, const float* b, long n, int act, float slope, float* y, void* stream) {
  LOFTR_CHECK_ARG(a && y && n >= 0 && act >= 0 && act <= 2);
  if (act == 2 && !(slope >= 0.f)) return LOFTR_ERR_UNSUPPORTED;      // the backward reads the derivative off the sign of the OUTPUT: valid for slope >= 0 only
  if (n == 0) return LOFTR_OK;
  hipLaunchKernelGGL(tg::act_fwd_kernel, tg::grid1d(n, 4), dim3(256), 0, (hipStream_t)stream, a, b, n, act, slope, y);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
extern "C" int loftr_act_bwd(const float* dy, const float* y, long n, int act, float slope, float* dx, void* stream) {
  LOFTR_CHECK_ARG(dy && dx && n >= 0 && act >= 0 && act <= 2 && (act == 0 || y));
  if (act == 2 && !(slope >= 0.f)) return LOFTR_ERR_UNSUPPORTED;
  if (n == 0) return LOFTR_OK;
  hipLaunchKernelGGL(tg::act_bwd_kernel, tg::grid1d(n), dim3(256), 0, (hipStream_t)stream, dy, y, n, act, slope, dx);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
static inline float up2_scale(int in) { return in > 1 ? (float)(in - 1) / (float)(2 * in - 1) : 0.f; }
extern "C" int loftr_upsample2x_bilinear_fwd(const float* x, int N, int C, int H, int W, int channels_last, float* y, void* stream) {
  LOFTR_CHECK_ARG(x && y && N >= 0 && C > 0 && H > 0 && W > 0);
  const long planes = (long)N * C;
  if (planes == 0) return LOFTR_OK;
  if (channels_last) {
    if (C % 4) return LOFTR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(tg::up2_fwd_nhwc_kernel, tg::grid1d((long)N * 4L * H * W * (C / 4)), dim3(256), 0, (hipStream_t)stream, x, (long)N, H, W, C, up2_scale(H), up2_scale(W), y);
    LOFTR_CHECK_LAUNCH();
    return LOFTR_OK;
  }
  hipLaunchKernelGGL(tg::up2_fwd_kernel, tg::grid1d(planes * 4L * H * W), dim3(256), 0, (hipStream_t)stream, x, planes, H, W, up2_scale(H), up2_scale(W), y);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
extern "C" int loftr_upsample2x_bilinear_bwd(const float* dy, int N, int C, int H, int W, int channels_last, float* dx, void* stream) {
  LOFTR_CHECK_ARG(dy && dx && N >= 0 && C > 0 && H > 0 && W > 0);
  const long planes = (long)N * C;
  if (planes == 0) return LOFTR_OK;
  if (channels_last) {
    if (C % 4) return LOFTR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(tg::up2_bwd_nhwc_kernel, tg::grid1d((long)N * H * W * (C / 4)), dim3(256), 0, (hipStream_t)stream, dy, (long)N, H, W, C, up2_scale(H), up2_scale(W), dx);
    LOFTR_CHECK_LAUNCH();
    return LOFTR_OK;
  }
  hipLaunchKernelGGL(tg::up2_bwd_kernel, tg::grid1d(planes * (long)H * W), dim3(256), 0, (hipStream_t)stream, dy, planes, H, W, up2_scale(H), up2_scale(W), dx);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
