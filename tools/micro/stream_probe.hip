// Which inner loop feeds the matrix cores better from an LDS weight stream?  (design probe for encoder_fused.hip)
//   A: 4 waves / workgroup (one per SIMD), a wave holds 32 tokens as B fragments, v_mfma_f32_32x32x16_f16, 48 MFMAs per 32 KB panel
//      (what encoder_x_kernel does);
//   B: 8 waves / workgroup (two per SIMD), a wave holds 16 tokens, v_mfma_f32_16x16x32_f16, 48 MFMAs per panel of half the duration:
//      finer work units and a partner wave to hide latencies, but every weight fragment read feeds half the flops (2x LDS traffic).
// Both: panels of 32 KB streamed by global_load_lds through a 4-stage ring from a 2 MB L2-resident buffer, one barrier per panel,
// 3 MFMAs per (hi, lo) fragment pair, ~40 VALU of "epilogue" per panel.  Prints time per panel and the executed fp16 MFMA rate.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_probe.hip -o tools/micro/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
#define WAITCNT_VM(n_) __builtin_amdgcn_s_waitcnt((((n_) & 0xF) | (((n_) >> 4) << 14) | (0x7 << 4) | (0xF << 8)))
constexpr int STAGE = 32 * 1024, NST = 4, NPANEL = 64;

template <int WAVES>
__device__ __forceinline__ void issue(const uint32_t* w, char* lds, int p, int wave, int lane) {
  // panel p: 32 KB contiguous; WAVES waves x (32 / WAVES) DMA instructions of 1 KB
  const uint32_t* src = w + (size_t)(p % 64) * (STAGE / 4);
  char* st = lds + (p % NST) * STAGE;
#pragma unroll
  for (int q = 0; q < 32 / WAVES; ++q) {
    const int blk = q * WAVES + wave;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + blk * 256 + lane * 4), (lds_ptr_t)(st + blk * 1024), 16, 0, 0);
  }
}

// A: 32x32x16, one wave per SIMD
__global__ __launch_bounds__(256, 1) void probe_a(const uint32_t* w, float* out, int rounds) {
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h8 xh[16], xl[16];
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 8; ++e) { xh[i][e] = (_Float16)(0.01f * (lane + i + e)); xl[i][e] = (_Float16)(0.001f * (lane - i + e)); }
  f16v big[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) big[j][r] = 0.f;
  float sink = 0.f;
  for (int rd = 0; rd < rounds; ++rd) {
    issue<4>(w, lds, 0, wave, lane); issue<4>(w, lds, 1, wave, lane); issue<4>(w, lds, 2, wave, lane);
#pragma unroll 1
    for (int p = 0; p < NPANEL; ++p) {
      if (p + 2 < NPANEL) WAITCNT_VM(16); else if (p + 1 < NPANEL) WAITCNT_VM(8); else WAITCNT_VM(0);
      __builtin_amdgcn_s_barrier();
      if (p + 3 < NPANEL) issue<4>(w, lds, p + 3, wave, lane);
      const char* st = lds + (p % NST) * STAGE + lane * 16;
      f16v acc;
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const h8 wh = *reinterpret_cast<const h8*>(st + ks * 2048), wl = *reinterpret_cast<const h8*>(st + ks * 2048 + 1024);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[ks], acc, 0, 0, 0);
      }
      // "epilogue": ~3 VALU per value
      for (int r = 0; r < 16; ++r) { float v = acc[r] * 1.0001f; v = v > 0.f ? v + 1.f : v * 0.5f; big[p & 7][r] += v; }
    }
  }
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) sink += big[j][r];
  if (sink == 1234.5f) out[threadIdx.x] = sink;
}

// B: 16x16x32, two waves per SIMD
__global__ __launch_bounds__(512, 2) void probe_b(const uint32_t* w, float* out, int rounds) {
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h8 xh[8], xl[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) { xh[i][e] = (_Float16)(0.01f * (lane + i + e)); xl[i][e] = (_Float16)(0.001f * (lane - i + e)); }
  f4v big[16];
  for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) big[j][r] = 0.f;
  float sink = 0.f;
  for (int rd = 0; rd < rounds; ++rd) {
    issue<8>(w, lds, 0, wave, lane); issue<8>(w, lds, 1, wave, lane); issue<8>(w, lds, 2, wave, lane);
#pragma unroll 1
    for (int p = 0; p < NPANEL; ++p) {
      if (p + 2 < NPANEL) WAITCNT_VM(8); else if (p + 1 < NPANEL) WAITCNT_VM(4); else WAITCNT_VM(0);
      __builtin_amdgcn_s_barrier();
      if (p + 3 < NPANEL) issue<8>(w, lds, p + 3, wave, lane);
      const char* st = lds + (p % NST) * STAGE + lane * 16;
      f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {                       // k-step of 32: (row half 0, row half 1) x (hi, lo)
        const h8 w0h = *reinterpret_cast<const h8*>(st + ks * 4096), w0l = *reinterpret_cast<const h8*>(st + ks * 4096 + 1024);
        const h8 w1h = *reinterpret_cast<const h8*>(st + ks * 4096 + 2048), w1l = *reinterpret_cast<const h8*>(st + ks * 4096 + 3072);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h, xl[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, xl[ks], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0l, xh[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l, xh[ks], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h, xh[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, xh[ks], acc1, 0, 0, 0);
      }
      for (int r = 0; r < 4; ++r) { float v = acc0[r] * 1.0001f; v = v > 0.f ? v + 1.f : v * 0.5f; big[(2 * p) & 15][r] += v;
                                    float u = acc1[r] * 1.0001f; u = u > 0.f ? u + 1.f : u * 0.5f; big[(2 * p + 1) & 15][r] += u; }
    }
  }
  for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) sink += big[j][r];
  if (sink == 1234.5f) out[threadIdx.x] = sink;
}

int main() {
  uint32_t* w; float* o;
  hipMalloc(&w, 64 * STAGE); hipMalloc(&o, 4096);
  hipMemset(w, 0x3c, 64 * STAGE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int rounds = 20;
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(probe_a, dim3(256), dim3(256), 0, 0, w, o, rounds);
      else hipLaunchKernelGGL(probe_b, dim3(256), dim3(512), 0, 0, w, o, rounds);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 0) continue;
      const double panels = (double)rounds * NPANEL;
      // flops per panel and CU: A: 4 waves x 48 MFMA x 32768; B: 8 waves x 48 x 16384 -- the same
      const double fl = 256.0 * panels * 4 * 48 * 32768.0;
      printf("%s: %.2f us per panel, %.0f TFLOP/s executed fp16 MFMA, tokens per CU in flight %d\n", which == 0 ? "A 32x32x16, 1 wave/SIMD" : "B 16x16x32, 2 waves/SIMD",
             ms * 1e3 / panels, fl / ms / 1e9, which == 0 ? 128 : 128);
    }
  }
  return 0;
}
