// Linear attention core (src/loftr/loftr_module/linear_attention.py:20-47) on pre-mapped inputs:
// the projection epilogue (linear.hip: proj_kernel) has already applied elu()+1 and the padding
// masks to Q and K and the mask and 1/S scaling to V, so what is left is
//     KV[n,h,d,v] = sum_s K[n,s,h,d] V[n,s,h,v]          (:43)
//     Ksum[n,h,d] = sum_s K[n,s,h,d]                     (:44, K.sum(dim=1))
//     out[n,l,h,v] = (sum_d Q[n,l,h,d] KV[n,h,d,v]) * S / (sum_d Q[n,l,h,d] Ksum[n,h,d] + eps)   (:44-45)
// At the coarse level the last line never runs as such: with z[l,h] = S / (Q[l,h,:].Ksum[h,:] + eps)
//     merge(out)[l,:] = sum_h z[l,h] (Q[l,h,:] @ KV_h) @ Wm[:, h-block]^T = (z (.) Q)[l,:] @ P,
//     P[(h,d), j] = sum_v KV[h,d,v] Wm[j, h*32+v]
// so the Q projection's epilogue applies z (the KV reduction runs before it) and the merge GEMM
// takes the per-pair P as its B operand: no message tensor, no separate apply kernel.
#include "attention.h"

// ------------------------------------------------------------------------------------------
// Coarse level (D = 32): the K^T V / Ksum reduction itself runs in the epilogue of the k/v projection
// (linear.hip: proj_kv_kernel, one partial per 128-row tile); what is left here is the deterministic
// fixed-order sum of those partials and the fold of KV into the merge projection.
// kv [nb,H,33,32]: rows 0..31 = KV[d][v], row 32 = Ksum[d].   grid (H, nb, 4), 256 threads.
// Also folds KV into the merge projection: pm[n][j][h*32+d] = sum_v KV[n,h,d,v] * Wm[j][h*32+v], written as SP.
// Block z handles output rows j = 64z .. 64z+63; thread (j, q) computes the channel octet d = 8q .. 8q+7 of
// row j (256 FMAs) and stores it as one hi and one lo 16-B chunk.  Every block re-sums the split partials
// (cheap) so the launch stays a single wave of work.
__global__ __launch_bounds__(256) void kv_finalize_kernel(const float* __restrict__ part,
                                                          float* __restrict__ kv, int splits,
                                                          const float* __restrict__ wm,
                                                          sp_t* __restrict__ pm) {
  __shared__ __attribute__((aligned(16))) float skv[33 * 32];
  __shared__ __attribute__((aligned(16))) float sgrp[4][33 * 32];
  const int h = blockIdx.x, n = blockIdx.y, H = gridDim.x;
  const long base = (long)n * H + h;
  const float* p = part + base * splits * (33 * 32);
  float* o = kv + base * (33 * 32);
  // wave w sums the partials k = w, w+4, ... as 16-B vectors (264 per partial, <= 5 per lane), then the four group
  // sums are added in a fixed order: ~4x fewer and 4x wider dependent loads than one thread per element
  {
    constexpr int NV = 33 * 32 / 4;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 a[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4                     // the loads of four partials in flight (same summation order): the loop is a DRAM-latency chain otherwise
    for (int k = w; k < splits; k += 4) {
      const f32x4* pk = reinterpret_cast<const f32x4*>(p + (long)k * (33 * 32));
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int v = lane + u * 64;
        if (v < NV) a[u] += pk[v];
      }
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int v = lane + u * 64;
      if (v < NV) reinterpret_cast<f32x4*>(sgrp[w])[v] = a[u];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 33 * 32; e += 256) {
    const float s = ((sgrp[0][e] + sgrp[1][e]) + sgrp[2][e]) + sgrp[3][e];
    if (blockIdx.z == 0) o[e] = s;
    skv[e] = s;
  }
  __syncthreads();
  const int C = H * 32;
  const int j = blockIdx.z * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
  f32x4 w4[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w4[i] = reinterpret_cast<const f32x4*>(wm + (long)j * C + h * 32)[i];
  float r[8];
#pragma unroll
  for (int dd = 0; dd < 8; ++dd) {
    const float* kvrow = &skv[(q * 8 + dd) * 32];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 k4 = *reinterpret_cast<const f32x4*>(kvrow + i * 4);
      s += k4.x * w4[i].x + k4.y * w4[i].y + k4.z * w4[i].z + k4.w * w4[i].w;
    }
    r[dd] = s;
  }
  // P feeds the merge GEMM as its B operand.  Its entries (KV / S times a merge weight) are O(0.01-0.1): stored times
  // the fixed power of two ATTN_P_SCALE so that the fp16 (hi, lo) pair keeps its 22 bits (gemm.h); the merge GEMM's
  // epilogue multiplies by 1 / ATTN_P_SCALE before the LayerNorm (linear.h: LinearLNArgs::out_scale).
#pragma unroll
  for (int dd = 0; dd < 8; ++dd) r[dd] *= ATTN_P_SCALE;
  u32x4 hi, lo;
  sp_pack8(r, hi, lo);
  sp_t* dst = pm + ((long)n * C + j) * C + h * 32 + q * 4;
  *reinterpret_cast<u32x4*>(dst) = hi;
  *reinterpret_cast<u32x4*>(dst + 16) = lo;
}

// ------------------------------------------------------------------------------------------
// Fine level (per-match windows: L = S = 25, C = 128, 8 heads of 16): the whole attention of one
// window in one block.  Thread c <-> channel c = (head h, column v).   grid (nb), C threads.
//   dynamic LDS: max(L, S) x C floats (K, then Q).
template <int D>
__global__ __launch_bounds__(128) void attn_small_kernel(const float* __restrict__ Qf,
                                                         const float* __restrict__ Kf,
                                                         const float* __restrict__ Vf,
                                                         sp_t* __restrict__ msg, int L, int S,
                                                         int C, float v_length, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sk = sm;               // [S][C] during the KV phase, then [L][C] = Q: one buffer, twice the resident windows
  float* sq = sm;
  const long n = blockIdx.x;
  const int c = threadIdx.x;
  const float* qn = Qf + n * L * C;
  const float* kn = Kf + n * S * C;
  const float* vn = Vf + n * S * C;
  for (int e = c; e < S * C / 4; e += blockDim.x)
    reinterpret_cast<f32x4*>(sk)[e] = reinterpret_cast<const f32x4*>(kn)[e];
  // Q is fetched now (registers) and parked in the same LDS buffer once the KV phase is done with K
  constexpr int QV = 8;                                       // f32x4 per thread: covers L * C <= 4096 floats
  f32x4 qreg[QV];
#pragma unroll
  for (int u = 0; u < QV; ++u) {
    const int e = c + u * 128;
    qreg[u] = e < L * C / 4 ? reinterpret_cast<const f32x4*>(qn)[e] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const int hb = (c / D) * D;   // first channel of this thread's head
  float kvr[D], ks[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { kvr[d] = 0.f; ks[d] = 0.f; }
  for (int s = 0; s < S; ++s) {
    const float v = vn[s * C + c];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float k = sk[s * C + hb + d];
      kvr[d] += k * v;
      ks[d] += k;
    }
  }
  __syncthreads();                                            // every thread is done with K
#pragma unroll
  for (int u = 0; u < QV; ++u) {
    const int e = c + u * 128;
    if (e < L * C / 4) reinterpret_cast<f32x4*>(sq)[e] = qreg[u];
  }
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float q = sq[l * C + hb + d];
      num += q * kvr[d];
      den += q * ks[d];
    }
    sp_store(msg + (n * L + l) * C, c, num * (v_length / (den + eps)), true);
  }
}

// ------------------------------------------------------------------------------------------
size_t attention_workspace_bytes(int nb, int S, int C) {
  if (C != 256) return 0;
  const int splits = ceil_div(S, 128);       // one partial per 128-row tile of the fused k/v projection
  return align_up((size_t)nb * 8 * splits * 33 * 32 * sizeof(float), 256) +
         align_up((size_t)nb * 8 * 33 * 32 * sizeof(float), 256) +
         align_up((size_t)nb * C * C * sizeof(sp_t), 256) + 1024;
}

float* attention_part_buffer(void* ws, size_t ws_bytes, int nb, int S) {
  WsAlloc wa(ws, ws_bytes);
  float* part = wa.take<float>((size_t)nb * 8 * ceil_div(S, 128) * 33 * 32);
  return wa.ok() ? part : nullptr;
}

int launch_attention_finalize(const float* merge_w, int nb, int S, int C, int H, void* ws, size_t ws_bytes,
                              const float** kv_out, const sp_t** pm_out, hipStream_t st) {
  if (!(C == 256 && H == 8)) return LOFTR_ERR_UNSUPPORTED;
  const int splits = ceil_div(S, 128);
  WsAlloc wa(ws, ws_bytes);
  float* part = wa.take<float>((size_t)nb * 8 * splits * 33 * 32);
  float* kv = wa.take<float>((size_t)nb * 8 * 33 * 32);
  sp_t* pm = wa.take<sp_t>((size_t)nb * C * C);
  if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
  hipLaunchKernelGGL(kv_finalize_kernel, dim3(8, nb, 4), dim3(256), 0, st, part, kv, splits, merge_w, pm);
  LOFTR_CHECK_LAUNCH();
  *kv_out = kv;
  *pm_out = pm;
  return LOFTR_OK;
}

int launch_attention_small(const float* Qf, const float* Kf, const float* Vf, sp_t* msg, int nb, int L, int S,
                           int C, int H, hipStream_t st) {
  if (nb <= 0) return LOFTR_OK;
  const float eps = 1e-6f;                         // LinearAttention(eps=1e-6), linear_attention.py:15
  if (C == 128 && H == 8 && L * C <= 4096 && S * C <= 16384) {
    const size_t lds = (size_t)(L > S ? L : S) * C * sizeof(float);
    TimedLaunch tl(LOFTR_T_ATTN_SMALL, st);
    hipLaunchKernelGGL((attn_small_kernel<16>), dim3(nb), dim3(128), lds, st, Qf, Kf, Vf, msg, L, S, C,
                       (float)S, eps);
    LOFTR_CHECK_LAUNCH();
    return LOFTR_OK;
  }
  return LOFTR_ERR_UNSUPPORTED;
}
