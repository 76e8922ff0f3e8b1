// Linear attention core (src/loftr/loftr_module/linear_attention.py:20-47) on pre-mapped inputs:
// the projection epilogue (linear.hip: proj_kernel) has already applied elu()+1 and the padding
// masks to Q and K and the mask and 1/S scaling to V, so what is left is
//     KV[n,h,d,v] = sum_s K[n,s,h,d] V[n,s,h,v]          (:43)
//     Ksum[n,h,d] = sum_s K[n,s,h,d]                     (:44, K.sum(dim=1))
//     out[n,l,h,v] = (sum_d Q[n,l,h,d] KV[n,h,d,v]) * S / (sum_d Q[n,l,h,d] Ksum[n,h,d] + eps)   (:44-45)
#include "attention.h"

// ------------------------------------------------------------------------------------------
// Coarse level (D = 32): K^T V is a [32 x S] x [S x 32] product per (n, head) -> fp32 MFMA fed
// straight from global memory (lane (d, half) reads K[s0 + half][d]: 128-B coalesced rows), the
// S axis split over blocks and waves; partials are summed in a fixed order (deterministic).
//   grid (splits, H, nb), 256 threads; each wave contracts KV_CHUNK/4 consecutive s.
constexpr int KV_CHUNK = 256;     // s-values per block

__global__ __launch_bounds__(256) void kv_partial_kernel(const float* __restrict__ Kf,
                                                         const float* __restrict__ Vf,
                                                         float* __restrict__ part,  // [nb,H,splits,33,32]
                                                         int S, int C, int splits) {
  __shared__ float red[4][33][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int split = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int H = gridDim.y;
  const int d = lane & 31, half = lane >> 5;
  const int s_begin = split * KV_CHUNK + wave * (KV_CHUNK / 4);
  const float* kp = Kf + ((long)n * S) * C + h * 32 + d;
  const float* vp = Vf + ((long)n * S) * C + h * 32 + d;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ksum = 0.f;
#pragma unroll 8
  for (int t = 0; t < KV_CHUNK / 4 / 2; ++t) {
    const int s = s_begin + 2 * t + half;
    float a = 0.f, b = 0.f;
    if (s < S) { a = kp[(long)s * C]; b = vp[(long)s * C]; }
    ksum += a;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);   // A[i=d][k=s], B[k=s][j=v]
  }
  ksum += __shfl_xor(ksum, 32, 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][d] = acc[r];   // [d][v]
  if (half == 0) red[wave][32][d] = ksum;
  __syncthreads();
  float* out = part + (((long)n * H + h) * splits + split) * (33 * 32);
  for (int e = threadIdx.x; e < 33 * 32; e += 256) {
    const float* r0 = &red[0][0][0];
    out[e] = (r0[e] + r0[33 * 32 + e]) + (r0[2 * 33 * 32 + e] + r0[3 * 33 * 32 + e]);
  }
}

// kv [nb,H,33,32]: rows 0..31 = KV[d][v], row 32 = Ksum[d].   grid (H, nb), 256 threads.
__global__ __launch_bounds__(256) void kv_finalize_kernel(const float* __restrict__ part,
                                                          float* __restrict__ kv, int splits) {
  const long base = ((long)blockIdx.y * gridDim.x + blockIdx.x);
  const float* p = part + base * splits * (33 * 32);
  float* o = kv + base * (33 * 32);
  for (int e = threadIdx.x; e < 33 * 32; e += 256) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += p[(long)k * (33 * 32) + e];
    o[e] = s;
  }
}

// out[n,l,h,:] for 64 tokens x 8 heads per block: one thread per (token, head), the 32x32 KV of
// its head broadcast from LDS.   grid (ceil(L/64), nb), 256 threads; wave w handles heads w, w+4.
__global__ __launch_bounds__(256) void attn_apply_kernel(const float* __restrict__ Qf,
                                                         const float* __restrict__ kv,
                                                         float* __restrict__ msg, int L, int C,
                                                         float v_length, float eps) {
  __shared__ __attribute__((aligned(16))) float skv[8][33][32];
  const int n = blockIdx.y;
  const float* kvn = kv + (long)n * 8 * 33 * 32;
  for (int e = threadIdx.x; e < 8 * 33 * 32 / 4; e += 256)
    reinterpret_cast<f32x4*>(&skv[0][0][0])[e] = reinterpret_cast<const f32x4*>(kvn)[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = blockIdx.x * 64 + lane;
  if (l >= L) return;
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = wave + 4 * hh;
    const float* qp = Qf + ((long)n * L + l) * C + h * 32;
    f32x4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = reinterpret_cast<const f32x4*>(qp)[i];
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 ks = *reinterpret_cast<const f32x4*>(&skv[h][32][i * 4]);
      z += q[i].x * ks.x + q[i].y * ks.y + q[i].z * ks.z + q[i].w * ks.w;
    }
    z = v_length / (z + eps);
    f32x4 o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dq = 0; dq < 8; ++dq) {
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        const float qv = q[dq][dd];
        const int dI = dq * 4 + dd;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 kvv = *reinterpret_cast<const f32x4*>(&skv[h][dI][i * 4]);
          o[i] += qv * kvv;
        }
      }
    }
    float* op = msg + ((long)n * L + l) * C + h * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<f32x4*>(op)[i] = o[i] * z;
  }
}

// ------------------------------------------------------------------------------------------
// Fine level (per-match windows: L = S = 25, C = 128, 8 heads of 16): the whole attention of one
// window in one block.  Thread c <-> channel c = (head h, column v).   grid (nb), C threads.
//   dynamic LDS: Q [L][C] + K [S][C].
template <int D>
__global__ __launch_bounds__(128) void attn_small_kernel(const float* __restrict__ Qf,
                                                         const float* __restrict__ Kf,
                                                         const float* __restrict__ Vf,
                                                         float* __restrict__ msg, int L, int S,
                                                         int C, float v_length, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sq = sm;               // [L][C]
  float* sk = sm + L * C;       // [S][C]
  const long n = blockIdx.x;
  const int c = threadIdx.x;
  const float* qn = Qf + n * L * C;
  const float* kn = Kf + n * S * C;
  const float* vn = Vf + n * S * C;
  for (int e = c; e < L * C / 4; e += blockDim.x)
    reinterpret_cast<f32x4*>(sq)[e] = reinterpret_cast<const f32x4*>(qn)[e];
  for (int e = c; e < S * C / 4; e += blockDim.x)
    reinterpret_cast<f32x4*>(sk)[e] = reinterpret_cast<const f32x4*>(kn)[e];
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
  for (int l = 0; l < L; ++l) {
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float q = sq[l * C + hb + d];
      num += q * kvr[d];
      den += q * ks[d];
    }
    msg[(n * L + l) * C + c] = num * (v_length / (den + eps));
  }
}

// ------------------------------------------------------------------------------------------
size_t attention_workspace_bytes(int nb, int S, int C) {
  if (C != 256) return 0;
  const int splits = ceil_div(S, KV_CHUNK);
  return align_up((size_t)nb * 8 * splits * 33 * 32 * sizeof(float), 256) +
         align_up((size_t)nb * 8 * 33 * 32 * sizeof(float), 256) + 512;
}

int launch_linear_attention(const float* Qf, const float* Kf, const float* Vf, float* msg, int nb,
                            int L, int S, int C, int H, void* ws, size_t ws_bytes, hipStream_t st) {
  if (nb <= 0) return LOFTR_OK;
  const float eps = 1e-6f;                         // LinearAttention(eps=1e-6), linear_attention.py:15
  if (C == 256 && H == 8) {
    const int splits = ceil_div(S, KV_CHUNK);
    WsAlloc wa(ws, ws_bytes);
    float* part = wa.take<float>((size_t)nb * 8 * splits * 33 * 32);
    float* kv = wa.take<float>((size_t)nb * 8 * 33 * 32);
    if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
    { TimedLaunch tl(LOFTR_T_KV, st);
      hipLaunchKernelGGL(kv_partial_kernel, dim3(splits, 8, nb), dim3(256), 0, st, Kf, Vf, part, S, C, splits); }
    hipLaunchKernelGGL(kv_finalize_kernel, dim3(8, nb), dim3(256), 0, st, part, kv, splits);
    { TimedLaunch tl(LOFTR_T_ATTN_APPLY, st);
      hipLaunchKernelGGL(attn_apply_kernel, dim3(ceil_div(L, 64), nb), dim3(256), 0, st, Qf, kv, msg, L, C,
                         (float)S, eps); }
    LOFTR_CHECK_LAUNCH();
    return LOFTR_OK;
  }
  if (C == 128 && H == 8 && (size_t)(L + S) * C * sizeof(float) <= 64 * 1024) {
    const size_t lds = (size_t)(L + S) * C * sizeof(float);
    { TimedLaunch tl(LOFTR_T_ATTN_SMALL, st);
      hipLaunchKernelGGL((attn_small_kernel<16>), dim3(nb), dim3(128), lds, st, Qf, Kf, Vf, msg, L, S, C,
                         (float)S, eps); }
    LOFTR_CHECK_LAUNCH();
    return LOFTR_OK;
  }
  return LOFTR_ERR_UNSUPPORTED;
}
