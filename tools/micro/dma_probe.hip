// How fast can a CU stream an L2-resident weight set into LDS?  (the bound of encoder_x_kernel's 2 MB-per-128-tokens weight stream)
// 256 workgroups x 4 waves, 32 KB panels through a 4-stage ring from a 2 MB buffer every workgroup reads in full, as the kernel does.
//   mode 0: global_load_lds only (waitcnt + barrier per panel)      mode 1: + the 32 ds_read_b128 fragment reads per wave and panel
//   mode 2: no DMA, 48 MFMAs per panel on stale LDS                   mode 3: mode 0 with the panel order rotated per workgroup
//   mode 4: mode 0 without the barrier                                mode 5: global_load_dwordx4 into registers (no LDS)
//   mode 6: mode 0 with 4-byte DMA (global_load_lds_dword)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_probe.hip -o tools/micro/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
#define WAITCNT_VM(n_) __builtin_amdgcn_s_waitcnt((((n_) & 0xF) | (((n_) >> 4) << 14) | (0x7 << 4) | (0xF << 8)))
constexpr int STAGE = 32 * 1024, NST = 4, NPANEL = 64;

template <int BYTES>
__device__ __forceinline__ void issue(const uint32_t* w, char* lds, int p, int rot, int wave, int lane) {
  const uint32_t* src = w + (size_t)((p + rot) % 64) * (STAGE / 4);
  char* st = lds + (p % NST) * STAGE;
  if (BYTES == 16) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int blk = q * 4 + wave;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + blk * 256 + lane * 4), (lds_ptr_t)(st + blk * 1024), 16, 0, 0);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int blk = q * 4 + wave;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + blk * 64 + lane), (lds_ptr_t)(st + blk * 256), 4, 0, 0);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const uint32_t* w, float* out, int rounds) {
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rot = MODE == 3 ? (int)(blockIdx.x * 5 % 64) : 0;
  h8 xh, xl;
  for (int e = 0; e < 8; ++e) { xh[e] = (_Float16)(0.01f * (lane + e)); xl[e] = (_Float16)(0.001f * (lane - e)); }
  f16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  u4 sum = {0, 0, 0, 0};
  constexpr int B = MODE == 6 ? 4 : 16, PER = MODE == 6 ? 32 : 8;
  for (int rd = 0; rd < rounds; ++rd) {
    if (MODE == 5) {
#pragma unroll 1
      for (int p = 0; p < NPANEL; ++p) {
        const uint32_t* src = w + (size_t)p * (STAGE / 4);
        u4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const u4*>(src + (q * 4 + wave) * 256 + lane * 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += v[q];
      }
      continue;
    }
    if (MODE != 2) { issue<B>(w, lds, 0, rot, wave, lane); issue<B>(w, lds, 1, rot, wave, lane); issue<B>(w, lds, 2, rot, wave, lane); }
#pragma unroll 1
    for (int p = 0; p < NPANEL; ++p) {
      if (MODE != 2) { if (p + 2 < NPANEL) WAITCNT_VM(2 * PER); else if (p + 1 < NPANEL) WAITCNT_VM(PER); else WAITCNT_VM(0); }
      if (MODE != 4) __builtin_amdgcn_s_barrier();
      if (MODE != 2 && p + 3 < NPANEL) issue<B>(w, lds, p + 3, rot, wave, lane);
      const char* st = lds + (p % NST) * STAGE + lane * 16;
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const h8 wh = *reinterpret_cast<const h8*>(st + ks * 2048), wl = *reinterpret_cast<const h8*>(st + ks * 2048 + 1024);
          if (MODE == 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
          } else {
            const u4 a = __builtin_bit_cast(u4, wh), b = __builtin_bit_cast(u4, wl);
            sum += a ^ b;
          }
        }
      }
    }
  }
  float sink = 0.f;
  for (int r = 0; r < 16; ++r) sink += acc[r];
  if (sink == 1234.5f || sum[0] + sum[1] + sum[2] + sum[3] == 0x12345u) out[threadIdx.x] = sink;
}

template <int MODE>
void run(const char* name, const uint32_t* w, float* o, int wgs) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int rounds = 20;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), 0, 0, w, o, rounds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double panels = (double)rounds * NPANEL;
  const double us = ms * 1e3 / panels;
  printf("%-44s %3d workgroups: %.2f us per 32 KB panel = %5.1f GB/s per CU, %5.2f TB/s chip\n", name, wgs, us, STAGE / us * 1e-3,
         STAGE / us * 1e-6 * (wgs < 256 ? wgs : 256));
}

int main() {
  uint32_t* w; float* o;
  hipMalloc(&w, 64 * STAGE); hipMalloc(&o, 4096);
  hipMemset(w, 0x3c, 64 * STAGE);
  for (int wgs : {256, 32, 8}) {
    run<0>("0 DMA b128 + barrier", w, o, wgs);
    run<6>("6 DMA b32 + barrier", w, o, wgs);
    run<4>("4 DMA b128, no barrier", w, o, wgs);
    run<3>("3 DMA b128, rotated panel order", w, o, wgs);
    run<1>("1 DMA b128 + fragment reads", w, o, wgs);
    run<5>("5 global_load_dwordx4 to registers", w, o, wgs);
    run<2>("2 no DMA: 48 MFMA per panel", w, o, wgs);
  }
  return 0;
}
