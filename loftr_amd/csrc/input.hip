// Input wire format of the matching path: what the reference's data loaders do on the host
// after decoding / resizing a grayscale image (src/utils/dataset.py:78-89,111-118,149; src/datasets/megadepth.py:116-121),
// done on the device from the uint8 pixels -- the host uploads 1 B per pixel instead of 4 and no longer pads,
// converts and builds masks per image.
//   src   [N, SH, SW] uint8 (row pitch / image pitch in bytes), hw [N,2] int32 valid (h, w) of each image
//   image [N, 1, PH, PW] float32 = src / 255 inside the valid rectangle, 0 outside   (pad_bottom_right + `/ 255`)
//   mask  [N, PH, PW]    uint8 1 inside the valid rectangle                           (optional)
//   maskc [N, PH/d, PW/d] uint8 = mask[d*y, d*x]   (F.interpolate nearest, scale 1/d; optional)
// One thread per 4 consecutive output pixels: a 4-B source read, one 16-B image store, one 4-B mask store.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void pack_gray_kernel(const uint8_t* __restrict__ src, long img_pitch, long row_pitch,
                                                        const int* __restrict__ hw, int PH, int PW,
                                                        float* __restrict__ image, uint8_t* __restrict__ mask) {
  const int n = blockIdx.z, y = blockIdx.y;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x0 >= PW) return;
  const int h = hw[2 * n], w = hw[2 * n + 1];
  const uint8_t* row = src + n * img_pitch + y * row_pitch;
  float v[4]; uint8_t m[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool in = y < h && x0 + e < w;
    v[e] = in ? (float)row[x0 + e] / 255.0f : 0.f;           // IEEE division: bit-identical to torch's `.float() / 255`
    m[e] = in ? 1 : 0;
  }
  const long o = ((long)n * PH + y) * PW + x0;
  if (x0 + 3 < PW && (o & 3) == 0) {
    *reinterpret_cast<f32x4*>(image + o) = f32x4{v[0], v[1], v[2], v[3]};
    if (mask) *reinterpret_cast<uint32_t*>(mask + o) = m[0] | (m[1] << 8) | (m[2] << 16) | ((uint32_t)m[3] << 24);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (x0 + e < PW) { image[o + e] = v[e]; if (mask) mask[o + e] = m[e]; }
  }
}

__global__ __launch_bounds__(256) void coarse_mask_kernel(const int* __restrict__ hw, int CH, int CW, int d,
                                                          uint8_t* __restrict__ maskc) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= CH * CW) return;
  const int y = i / CW, x = i - y * CW;
  maskc[(long)n * CH * CW + i] = (y * d < hw[2 * n] && x * d < hw[2 * n + 1]) ? 1 : 0;
}

// cv2.resize(uint8 gray, INTER_LINEAR) restated (OpenCV 4.x resize.cpp, 8UC1, fixed point with 11-bit coefficients):
// one thread per output pixel, both passes fused.  PARITY UNPINNED (OpenCV absent here): pinned to the numpy restatement
// oracle/input_oracle.py:resize_linear_u8 only.
__device__ __forceinline__ void resize_axis(int d, int n_dst, int n_src, bool zero_at_border, int& s, int& c0, int& c1) {
  const double scale = 1.0 / ((double)n_dst / (double)n_src);
  float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);      // separately rounded like the host code (no fma)
  s = (int)floorf(f);
  f -= (float)s;
  if (zero_at_border) {                       // columns: the coefficient is forced to 0 outside, rows only clamp the index
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
  }
  c0 = __float2int_rn((1.0f - f) * 2048.0f);  // saturate_cast<short>(cvRound(.)): |value| <= 2048, never saturates
  c1 = __float2int_rn(f * 2048.0f);
}

__global__ __launch_bounds__(256) void resize_linear_kernel(const uint8_t* __restrict__ src, int sh, int sw, long src_pitch,
                                                            uint8_t* __restrict__ dst, int dh, int dw, long dst_pitch) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= dw) return;
  int sx, a0, a1, sy, b0, b1;
  resize_axis(x, dw, sw, true, sx, a0, a1);
  resize_axis(y, dh, sh, false, sy, b0, b1);
  const int x1 = min(sx + 1, sw - 1);
  const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
  const uint8_t* r0 = src + y0 * src_pitch;
  const uint8_t* r1 = src + y1 * src_pitch;
  const int h0 = (int)r0[sx] * a0 + (int)r0[x1] * a1;            // horizontal pass, scale 2^11
  const int h1 = (int)r1[sx] * a0 + (int)r1[x1] * a1;
  const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
  dst[y * dst_pitch + x] = (uint8_t)min(max(v, 0), 255);
}

}  // namespace

extern "C" int loftr_resize_linear_u8(const uint8_t* src, int sh, int sw, long src_pitch, uint8_t* dst, int dh, int dw,
                                      long dst_pitch, void* stream) {
  LOFTR_CHECK_ARG(src && dst && sh > 0 && sw > 0 && dh > 0 && dw > 0 && src_pitch >= sw && dst_pitch >= dw);
  if (dh > 65535) return LOFTR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(resize_linear_kernel, dim3(ceil_div(dw, 256), dh), dim3(256), 0, (hipStream_t)stream, src, sh, sw, src_pitch,
                     dst, dh, dw, dst_pitch);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

extern "C" int loftr_pack_gray_u8(const uint8_t* src, long src_image_pitch, long src_row_pitch, const int* hw, int N, int PH,
                                  int PW, float* image, uint8_t* mask, uint8_t* mask_c, int coarse_div, void* stream) {
  LOFTR_CHECK_ARG(N >= 0 && PH > 0 && PW > 0 && coarse_div >= 0);
  if (N == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(src && hw && image && src_row_pitch > 0 && src_image_pitch > 0 && (!mask_c || coarse_div > 0));
  if (N > 65535 || PH > 65535) return LOFTR_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pack_gray_kernel, dim3(ceil_div(ceil_div(PW, 4), 256), PH, N), dim3(256), 0, st, src, src_image_pitch,
                     src_row_pitch, hw, PH, PW, image, mask);
  LOFTR_CHECK_LAUNCH();
  if (mask_c) {
    const int CH = PH / coarse_div, CW = PW / coarse_div;
    if (CH > 0 && CW > 0) {
      hipLaunchKernelGGL(coarse_mask_kernel, dim3(ceil_div(CH * CW, 256), N), dim3(256), 0, st, hw, CH, CW, coarse_div, mask_c);
      LOFTR_CHECK_LAUNCH();
    }
  }
  return LOFTR_OK;
}
