// Backward of one LoFTREncoderLayer (the node between the matching heads' backward and the backbone).
//   reference: src/loftr/loftr_module/transformer.py:35-58 (q/k/v projections, merge, norm1, mlp on cat[x, message], norm2, residual)
//              src/loftr/loftr_module/linear_attention.py:20-47 (elu + 1 feature map, masks, V / S, KV, Ksum, Z, message * S)
// What torch.autograd derives from that forward, restated as kernels.  Nothing is saved by the (fused, inference-shaped) forward:
// the layer is RECOMPUTED here from (x, source) in an unfused form that keeps every intermediate in the workspace, then
// differentiated back to front:
//     q = x Wq^T, k = s Wk^T, v = s Wv^T;  Q = (elu(q)+1) mq, K = (elu(k)+1) ms, V' = v ms / S
//     KV_h = K_h^T V'_h, Ksum_h = sum_s K_h;  A = Q_h KV_h, Z = 1 / (Q_h . Ksum_h + eps), msg0 = A Z S
//     m1 = msg0 Wm^T, m2 = LN1(m1), hcat = [x, m2], h1 = relu(hcat W0^T), m3 = h1 W2^T, out = x + LN2(m3)
// Matrix products: the fp32-accurate split-fp16 MFMA GEMMs of this library -- loftr_linear_fwd (A W^T; the data gradients use a
// transposed copy of the weight) and launch_head_grad (A^T B with a long reduction: the weight gradients, as a split-K batch whose
// partial sums are added in a fixed order).  Everything else (feature map, the 32 x 32 per-head contractions of linear attention,
// LayerNorm, ReLU) is plain fp32 vector code: ~1 GFLOP per call against ~25 GFLOP x 3 of GEMMs.
// Training-path code: correct and deterministic first, not tuned.
#include "linear.h"
#include "head_grads.h"

extern "C" size_t loftr_linear_workspace_bytes(int M, int N, int K);
extern "C" int loftr_linear_fwd(const float* a, const float* w, float* out, int M, int N, int K, void* ws, size_t ws_bytes, void* stream);

namespace {
namespace eb {
constexpr int MAXD = 32;            // head dimension (coarse 32, fine 16)

__device__ __forceinline__ float elu1d(float x) { return x > 0.f ? 1.f : __expf(x); }      // d/dx (elu(x) + 1)

// out[c][r] = w[r][c]
__global__ void transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int R, int Cc) {
  __shared__ float t[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    t[i][threadIdx.x] = (r < R && c < Cc) ? w[(long)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < Cc) wt[(long)c * R + r] = t[threadIdx.x][i];
  }
}

// Per (sequence n, head h): mat[d][e] = sum_t fa(a[t][d]) * b[t][e] * bscale,  vec[d] = sum_t fa(a[t][d]) * w[t]
//   fa(x) = (elu(x) + 1) * mask[t]   (the feature map of linear_attention.py:31-39);  b is multiplied by mask[t] too when mask_b.
//   KV / Ksum:   a = k, b = v, bscale = 1 / S, mask_b, w = null (1)            (linear_attention.py:41-43)
//   dKV / dKsum: a = q, b = dA, bscale = 1, w = dDen                            (their gradients)
// grid (nb * H), 256 threads; thread i owns entries i, i + 256, .. of the D x D matrix.  Fixed summation order.
// gridDim.y > 1: block (., y) sums tokens [y chunk, (y + 1) chunk) into partial y (mat / vec then point at the partial buffers,
// [gridDim.y][nb * H][D * D] and [gridDim.y][nb * H][D]; outer_reduce_kernel adds them in order): nb * H blocks alone are 16 .. 32
// workgroups on 256 CUs (408 us per call at 4 x 4800 tokens, 30 % of the matcher's backward).
__global__ __launch_bounds__(256) void outer_accum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const uint8_t* __restrict__ mask, int mask_b, const float* __restrict__ w,
                                                          float bscale, int Tn, int C, int H, int D, int chunk, float* __restrict__ mat,
                                                          float* __restrict__ vec) {
  __shared__ float As[64][MAXD + 1], Bs[64][MAXD + 1], Ws[64];
  const int n = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, vacc = 0.f;
  const int DD = D * D;
  const int tb = blockIdx.y * chunk, te = min(Tn, tb + chunk);
  mat += (long)blockIdx.y * gridDim.x * DD;
  vec += (long)blockIdx.y * gridDim.x * D;
  for (int t0 = tb; t0 < te; t0 += 64) {
    for (int i = tid; i < 64 * D; i += 256) {
      const int tt = i / D, d = i - tt * D, t = t0 + tt;
      float av = 0.f, bv = 0.f;
      if (t < te) {
        const long off = ((long)n * Tn + t) * C + h * D + d;
        const float mk = mask ? (mask[(long)n * Tn + t] ? 1.f : 0.f) : 1.f;
        av = elu1(a[off]) * mk;
        bv = b[off] * bscale * (mask_b ? mk : 1.f);
      }
      As[tt][d] = av; Bs[tt][d] = bv;
    }
    if (tid < 64) { const int t = t0 + tid; Ws[tid] = t < te ? (w ? w[((long)n * Tn + t) * H + h] : 1.f) : 0.f; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + u * 256;
      if (idx < DD) {
        const int d = idx / D, e = idx - d * D;
        float s = acc[u];
        for (int tt = 0; tt < 64; ++tt) s = fmaf(As[tt][d], Bs[tt][e], s);
        acc[u] = s;
      }
    }
    if (tid < D) { float s = vacc; for (int tt = 0; tt < 64; ++tt) s = fmaf(As[tt][tid], Ws[tt], s); vacc = s; }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int idx = tid + u * 256; if (idx < DD) mat[(long)blockIdx.x * DD + idx] = acc[u]; }
  if (tid < D) vec[(long)blockIdx.x * D + tid] = vacc;
}

// out[i] = part[0][i] + part[1][i] + ...  (fixed order)
__global__ void outer_reduce_kernel(const float* __restrict__ part, int np, long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = part[i];
  for (int p = 1; p < np; ++p) s += part[(long)p * n + i];
  out[i] = s;
}
// tokens split over up to 16 chunks of >= 512 when the scratch holds the partials (scratch_floats), else one block per (pair, head)
void launch_outer_accum(const float* a, const float* b, const uint8_t* mask, int mask_b, const float* w, float bscale, int Tn, int C, int H, int D,
                        int nb, float* mat, float* vec, float* scratch, size_t scratch_floats, hipStream_t st) {
  int splits = Tn > 1024 ? ceil_div(Tn, 512) : 1;
  if (splits > 16) splits = 16;
  const long nm = (long)nb * H * D * D, nv = (long)nb * H * D;
  if ((size_t)splits * (size_t)(nm + nv) > scratch_floats) splits = 1;
  if (splits == 1) {
    hipLaunchKernelGGL(outer_accum_kernel, dim3(nb * H), dim3(256), 0, st, a, b, mask, mask_b, w, bscale, Tn, C, H, D, Tn, mat, vec);
    return;
  }
  const int chunk = ceil_div(ceil_div(Tn, splits), 64) * 64;
  splits = ceil_div(Tn, chunk);
  float* pm = scratch; float* pv = scratch + (size_t)splits * nm;
  hipLaunchKernelGGL(outer_accum_kernel, dim3(nb * H, splits), dim3(256), 0, st, a, b, mask, mask_b, w, bscale, Tn, C, H, D, chunk, pm, pv);
  hipLaunchKernelGGL(outer_reduce_kernel, dim3((unsigned)ceil_div((int)nm, 256)), dim3(256), 0, st, pm, splits, nm, mat);
  hipLaunchKernelGGL(outer_reduce_kernel, dim3((unsigned)ceil_div((int)nv, 256)), dim3(256), 0, st, pv, splits, nv, vec);
}

// Per (token l, head h):  Q = (elu(q)+1) mq;  A = Q KV;  Z = 1 / (Q . Ksum + eps);  msg0 = A Z S            (linear_attention.py:44-45)
// BWD: additionally, with G = dmsg0 * S:  dA = G Z,  dDen = -(sum_e G A) Z^2,  dq = (dA KV^T + dDen Ksum) * mq * elu'(q)
// grid (ceil(L / 256), nb * H), 256 threads.
template <bool BWD, int D>
__global__ __launch_bounds__(256) void attn_q_kernel(const float* __restrict__ q, const uint8_t* __restrict__ mask,
                                                     const float* __restrict__ KV, const float* __restrict__ Ksum, float vlen, float eps,
                                                     int L, int C, int H, float* __restrict__ msg0,
                                                     const float* __restrict__ dmsg0, float* __restrict__ dA_out,
                                                     float* __restrict__ dDen_out, float* __restrict__ dq) {
  __shared__ float kv[MAXD][MAXD + 1], ks[MAXD];
  const int nh = blockIdx.y, n = nh / H, h = nh % H;
  for (int i = threadIdx.x; i < D * D; i += 256) kv[i / D][i % D] = KV[(long)nh * D * D + i];
  if (threadIdx.x < D) ks[threadIdx.x] = Ksum[(long)nh * D + threadIdx.x];
  __syncthreads();
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const long row = (long)n * L + l, off = row * C + h * D;
  const float mk = mask ? (mask[row] ? 1.f : 0.f) : 1.f;
  float qv[D], Q[D], A[D];
  float den = eps;
#pragma unroll
  for (int d = 0; d < D; ++d) { qv[d] = q[off + d]; Q[d] = elu1(qv[d]) * mk; den = fmaf(Q[d], ks[d], den); }
  const float Z = 1.f / den;
#pragma unroll
  for (int e = 0; e < D; ++e) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) s = fmaf(Q[d], kv[d][e], s);
    A[e] = s;
  }
  if (!BWD) {
#pragma unroll
    for (int e = 0; e < D; ++e) msg0[off + e] = A[e] * Z * vlen;
    return;
  }
  float dAe[D], dZ = 0.f;
#pragma unroll
  for (int e = 0; e < D; ++e) { const float G = dmsg0[off + e] * vlen; dAe[e] = G * Z; dZ = fmaf(G, A[e], dZ); dA_out[off + e] = dAe[e]; }
  const float dDen = -dZ * Z * Z;
  dDen_out[row * H + h] = dDen;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float s = dDen * ks[d];
#pragma unroll
    for (int e = 0; e < D; ++e) s = fmaf(dAe[e], kv[d][e], s);
    dq[off + d] = s * mk * elu1d(qv[d]);
  }
}

// Per (source token s, head h):  dK = dKV V' + dKsum,  dV' = dKV^T K;  dk = dK ms elu'(k),  dv = dV' ms / S
template <int D>
__global__ __launch_bounds__(256) void attn_kv_bwd_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                          const uint8_t* __restrict__ mask, const float* __restrict__ dKV,
                                                          const float* __restrict__ dKsum, float inv_s, int S, int C, int H,
                                                          float* __restrict__ dk, float* __restrict__ dv) {
  __shared__ float g[MAXD][MAXD + 1], gs[MAXD];
  const int nh = blockIdx.y, n = nh / H, h = nh % H;
  for (int i = threadIdx.x; i < D * D; i += 256) g[i / D][i % D] = dKV[(long)nh * D * D + i];
  if (threadIdx.x < D) gs[threadIdx.x] = dKsum[(long)nh * D + threadIdx.x];
  __syncthreads();
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const long row = (long)n * S + s, off = row * C + h * D;
  const float mk = mask ? (mask[row] ? 1.f : 0.f) : 1.f;
  float kvv[D], K[D], V[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { kvv[d] = k[off + d]; K[d] = elu1(kvv[d]) * mk; V[d] = v[off + d] * mk * inv_s; }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float t = gs[d];
#pragma unroll
    for (int e = 0; e < D; ++e) t = fmaf(g[d][e], V[e], t);
    dk[off + d] = t * mk * elu1d(kvv[d]);
  }
#pragma unroll
  for (int e = 0; e < D; ++e) {
    float t = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) t = fmaf(K[d], g[d][e], t);
    dv[off + e] = t * mk * inv_s;
  }
}

// LayerNorm over the last dimension (C <= 512), one wave per row.  y (optional) = xhat * gamma + beta; stats[row] = (mean, rstd).
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, long rows, int C, float* __restrict__ y,
                                                     float2* __restrict__ stats) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[8], s = 0.f;
  int n = 0;
  for (int c = lane; c < C; c += 64) { v[n] = x[row * C + c]; s += v[n]; ++n; }
  const float mean = wave_sum(s) / (float)C;
  float m2 = 0.f;
  for (int i = 0; i < n; ++i) { const float d = v[i] - mean; m2 = fmaf(d, d, m2); }
  const float rstd = rsqrtf(wave_sum(m2) / (float)C + eps);
  if (lane == 0) stats[row] = make_float2(mean, rstd);
  if (y) { n = 0; for (int c = lane; c < C; c += 64) { y[row * C + c] = (v[n] - mean) * rstd * gamma[c] + beta[c]; ++n; } }
}
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; per block partial sums of dgamma = dy * xhat, dbeta = dy over
// its ROWS_PB rows (fixed order), part[block][2][C].
constexpr int LN_ROWS_PB = 64;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float2* __restrict__ stats, const float* __restrict__ gamma, long rows, int C,
                                                     float* __restrict__ dx, float* __restrict__ part) {
  __shared__ float red[4][2][512];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float dg[8], db[8];
  for (int i = 0; i < 8; ++i) { dg[i] = 0.f; db[i] = 0.f; }
  for (int rr = wave; rr < LN_ROWS_PB; rr += 4) {
    const long row = (long)blockIdx.x * LN_ROWS_PB + rr;
    if (row >= rows) break;
    const float2 st = stats[row];
    float g[8], xh[8], sg = 0.f, sgx = 0.f;
    int n = 0;
    for (int c = lane; c < C; c += 64) {
      const float d = dy[row * C + c];
      xh[n] = (x[row * C + c] - st.x) * st.y;
      g[n] = d * gamma[c];
      sg += g[n]; sgx = fmaf(g[n], xh[n], sgx);
      dg[n] = fmaf(d, xh[n], dg[n]); db[n] += d;
      ++n;
    }
    const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
    n = 0;
    for (int c = lane; c < C; c += 64) { dx[row * C + c] = st.y * (g[n] - mg - xh[n] * mgx); ++n; }
  }
  int n = 0;
  for (int c = lane; c < C; c += 64) { red[wave][0][c] = dg[n]; red[wave][1][c] = db[n]; ++n; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    part[((long)blockIdx.x * 2 + 0) * C + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
    part[((long)blockIdx.x * 2 + 1) * C + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
  }
}
// hcat[t] = [x[t], m[t]]
__global__ void concat_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ out, long rows, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * 2 * C) return;
  const long r = i / (2 * C); const int c = (int)(i - r * 2 * C);
  out[i] = c < C ? x[r * C + c] : m[r * C + c - C];
}
__global__ void relu_kernel(float* __restrict__ h, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) h[i] = fmaxf(h[i], 0.f);
}
__global__ void relu_bwd_kernel(float* __restrict__ dh, const float* __restrict__ h, long n) {     // dh *= (h > 0)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dh[i] = h[i] > 0.f ? dh[i] : 0.f;
}
// out[r][c] = a[r][c] + b[r * ldb + c] (+ c3[r][c])
__global__ void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, long ldb, const float* __restrict__ c3,
                                float* __restrict__ out, long rows, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const long r = i / C; const int c = (int)(i - r * C);
  float s = a[i] + b[r * ldb + c];
  if (c3) s += c3[i];
  out[i] = s;
}
// out[r][c] = src[r * lds + c]   (column slice -> contiguous)
__global__ void slice_cols_kernel(const float* __restrict__ src, long lds_, float* __restrict__ out, long rows, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const long r = i / C; const int c = (int)(i - r * C);
  out[i] = src[r * lds_ + c];
}

struct Ws {
  float *q, *k, *v, *KV, *Ksum, *msg0, *m1, *m2, *hcat, *h1, *m3, *dm3, *dh1, *dhcat, *dm2, *dm1, *dmsg0, *dA, *dDen, *dKV, *dKsum,
        *dq, *dk, *dv, *t0, *t1, *wqT, *wkT, *wvT, *wmT, *w0T, *w2T, *wpart, *lnpart;
  float2 *st1, *st2;
  void* lin; size_t lin_bytes, wpart_floats;
  bool ok;
};
// split-K scratch for every weight gradient of the layer: the partial count is not monotone in the token count (wgrad_chunk rounds and
// clamps), so the x side (T tokens) and the source side (Ts tokens) are sized separately
static size_t wpart_need(long T, long Ts, int C) {
  const size_t a = wgrad_part_floats(T, 2 * C, 2 * C), b = wgrad_part_floats(Ts, 2 * C, 2 * C);
  return a > b ? a : b;
}
Ws carve(WsAlloc& wa, int nb, int L, int S, int C, int H) {
  Ws w{};
  const size_t T = (size_t)nb * L, Ts = (size_t)nb * S, D = C / H, Tm = T > Ts ? T : Ts;
  w.q = wa.take<float>(T * C); w.k = wa.take<float>(Ts * C); w.v = wa.take<float>(Ts * C);
  w.KV = wa.take<float>((size_t)nb * H * D * D); w.Ksum = wa.take<float>((size_t)nb * H * D);
  w.msg0 = wa.take<float>(T * C); w.m1 = wa.take<float>(T * C); w.m2 = wa.take<float>(T * C);
  w.hcat = wa.take<float>(T * 2 * C); w.h1 = wa.take<float>(T * 2 * C); w.m3 = wa.take<float>(T * C);
  w.dm3 = wa.take<float>(T * C); w.dh1 = wa.take<float>(T * 2 * C); w.dhcat = wa.take<float>(T * 2 * C);
  w.dm2 = wa.take<float>(T * C); w.dm1 = wa.take<float>(T * C); w.dmsg0 = wa.take<float>(T * C);
  w.dA = wa.take<float>(T * C); w.dDen = wa.take<float>(T * H);
  w.dKV = wa.take<float>((size_t)nb * H * D * D); w.dKsum = wa.take<float>((size_t)nb * H * D);
  w.dq = wa.take<float>(T * C); w.dk = wa.take<float>(Ts * C); w.dv = wa.take<float>(Ts * C);
  w.t0 = wa.take<float>(Tm * C); w.t1 = wa.take<float>(Tm * C);
  w.wqT = wa.take<float>((size_t)C * C); w.wkT = wa.take<float>((size_t)C * C); w.wvT = wa.take<float>((size_t)C * C);
  w.wmT = wa.take<float>((size_t)C * C); w.w0T = wa.take<float>((size_t)4 * C * C); w.w2T = wa.take<float>((size_t)2 * C * C);
  w.wpart_floats = wpart_need((long)T, (long)Ts, C);
  w.wpart = wa.take<float>(w.wpart_floats);
  w.lnpart = wa.take<float>((size_t)ceil_div((int)T, LN_ROWS_PB) * 2 * C);
  w.st1 = wa.take<float2>(T); w.st2 = wa.take<float2>(T);
  w.lin_bytes = loftr_linear_workspace_bytes((int)Tm, 2 * C, 2 * C);
  w.lin = wa.take<char>(w.lin_bytes);
  w.ok = wa.ok();
  return w;
}
size_t ws_bytes_needed(int nb, int L, int S, int C, int H) {
  WsAlloc wa(nullptr, ~(size_t)0);                    // the same carve on a null base: pointer arithmetic only
  (void)carve(wa, nb, L, S, C, H);
  return wa.off + 256;
}
inline dim3 g1d(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace eb
}  // namespace

extern "C" size_t loftr_encoder_layer_bwd_workspace_bytes(int nb, int L, int S, int C, int H) {
  if (nb <= 0 || L <= 0 || S <= 0 || C <= 0 || H <= 0 || C % H) return 0;
  return eb::ws_bytes_needed(nb, L, S, C, H);
}

extern "C" int loftr_encoder_layer_bwd(const float* x, const float* source, const uint8_t* x_mask, const uint8_t* source_mask,
                                       const loftr_layer_weights* w, const float* grad_out, float* grad_x, float* grad_source,
                                       const loftr_layer_grads* gw, int nb, int L, int S, int C, int H, void* ws, size_t ws_bytes,
                                       void* stream) {
  using namespace eb;
  LOFTR_CHECK_ARG(x && source && w && grad_out && grad_x && grad_source && gw && nb >= 0 && L > 0 && S > 0 && C > 0 && H > 0);
  if (C % H != 0 || (C / H != 32 && C / H != 16) || C % 32 != 0 || C > 256) return LOFTR_ERR_UNSUPPORTED;
  if (nb == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(ws != nullptr);
  hipStream_t st = (hipStream_t)stream;
  WsAlloc wsa(ws, ws_bytes);
  Ws a = carve(wsa, nb, L, S, C, H);
  if (!a.ok) return LOFTR_ERR_WORKSPACE;
  const int D = C / H, C2 = 2 * C;
  const long T = (long)nb * L, Ts = (long)nb * S;
  const size_t wpart_floats = a.wpart_floats;      // (a.wpart also holds the K V partials: not alive together)
  const float vlen = (float)S, inv_s = 1.f / (float)S, attn_eps = 1e-6f, ln_eps = 1e-5f;      // linear_attention.py:26,41; nn.LayerNorm default
  int rc;
#define LIN(A_, W_, OUT_, M_, N_, K_) if ((rc = loftr_linear_fwd(A_, W_, OUT_, (int)(M_), N_, K_, a.lin, a.lin_bytes, stream))) return rc
  const dim3 tb(32, 8);
  auto transpose = [&](const float* src, float* dst, int R, int Cc) {
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(Cc, 32), ceil_div(R, 32)), tb, 0, st, src, dst, R, Cc);
  };
  // ---- forward, unfused, every intermediate kept
  LIN(x, w->q_proj, a.q, T, C, C);
  LIN(source, w->k_proj, a.k, Ts, C, C);
  LIN(source, w->v_proj, a.v, Ts, C, C);
  launch_outer_accum(a.k, a.v, source_mask, 1, nullptr, inv_s, S, C, H, D, nb, a.KV, a.Ksum, a.wpart, wpart_floats, st);
  if (D == 32)
    hipLaunchKernelGGL((attn_q_kernel<false, 32>), dim3(ceil_div(L, 256), nb * H), dim3(256), 0, st, a.q, x_mask, a.KV, a.Ksum, vlen, attn_eps,
                       L, C, H, a.msg0, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  else
    hipLaunchKernelGGL((attn_q_kernel<false, 16>), dim3(ceil_div(L, 256), nb * H), dim3(256), 0, st, a.q, x_mask, a.KV, a.Ksum, vlen, attn_eps,
                       L, C, H, a.msg0, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  LIN(a.msg0, w->merge, a.m1, T, C, C);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)ceil_div((int)T, 4)), dim3(256), 0, st, a.m1, w->norm1_w, w->norm1_b, ln_eps, T, C, a.m2, a.st1);
  hipLaunchKernelGGL(concat_kernel, g1d(T * C2), dim3(256), 0, st, x, a.m2, a.hcat, T, C);
  LIN(a.hcat, w->mlp0, a.h1, T, C2, C2);
  hipLaunchKernelGGL(relu_kernel, g1d(T * C2), dim3(256), 0, st, a.h1, T * C2);
  LIN(a.h1, w->mlp2, a.m3, T, C, C2);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)ceil_div((int)T, 4)), dim3(256), 0, st, a.m3, w->norm2_w, w->norm2_b, ln_eps, T, C,
                     (float*)nullptr, a.st2);
  // ---- backward: out = x + LN2(m3)
  const int nlb = ceil_div((int)T, LN_ROWS_PB);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nlb), dim3(256), 0, st, grad_out, a.m3, a.st2, w->norm2_w, T, C, a.dm3, a.lnpart);
  launch_reduce_partials(a.lnpart, gw->norm2_w, nlb, (long)2 * C, (long)C, st);
  launch_reduce_partials(a.lnpart + C, gw->norm2_b, nlb, (long)2 * C, (long)C, st);
  // m3 = h1 W2^T
  transpose(w->mlp2, a.w2T, C, C2);                                    // [C, 2C] -> [2C, C]
  LIN(a.dm3, a.w2T, a.dh1, T, C2, C);                                  // dh1 = dm3 W2
  if ((rc = launch_wgrad(a.dm3, C, a.h1, C2, T, gw->mlp2, a.wpart, a.wpart_floats, st))) return rc;
  hipLaunchKernelGGL(relu_bwd_kernel, g1d(T * C2), dim3(256), 0, st, a.dh1, a.h1, T * C2);
  // h1 = relu(hcat W0^T)
  transpose(w->mlp0, a.w0T, C2, C2);
  LIN(a.dh1, a.w0T, a.dhcat, T, C2, C2);
  if ((rc = launch_wgrad(a.dh1, C2, a.hcat, C2, T, gw->mlp0, a.wpart, a.wpart_floats, st))) return rc;
  hipLaunchKernelGGL(slice_cols_kernel, g1d(T * C), dim3(256), 0, st, a.dhcat + C, (long)C2, a.dm2, T, C);
  // m2 = LN1(m1)
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nlb), dim3(256), 0, st, a.dm2, a.m1, a.st1, w->norm1_w, T, C, a.dm1, a.lnpart);
  launch_reduce_partials(a.lnpart, gw->norm1_w, nlb, (long)2 * C, (long)C, st);
  launch_reduce_partials(a.lnpart + C, gw->norm1_b, nlb, (long)2 * C, (long)C, st);
  // m1 = msg0 Wm^T
  transpose(w->merge, a.wmT, C, C);
  LIN(a.dm1, a.wmT, a.dmsg0, T, C, C);
  if ((rc = launch_wgrad(a.dm1, C, a.msg0, C, T, gw->merge, a.wpart, a.wpart_floats, st))) return rc;
  // linear attention
  if (D == 32)
    hipLaunchKernelGGL((attn_q_kernel<true, 32>), dim3(ceil_div(L, 256), nb * H), dim3(256), 0, st, a.q, x_mask, a.KV, a.Ksum, vlen, attn_eps,
                       L, C, H, (float*)nullptr, (const float*)a.dmsg0, a.dA, a.dDen, a.dq);
  else
    hipLaunchKernelGGL((attn_q_kernel<true, 16>), dim3(ceil_div(L, 256), nb * H), dim3(256), 0, st, a.q, x_mask, a.KV, a.Ksum, vlen, attn_eps,
                       L, C, H, (float*)nullptr, (const float*)a.dmsg0, a.dA, a.dDen, a.dq);
  launch_outer_accum(a.q, a.dA, x_mask, 0, a.dDen, 1.f, L, C, H, D, nb, a.dKV, a.dKsum, a.wpart, wpart_floats, st);
  if (D == 32)
    hipLaunchKernelGGL((attn_kv_bwd_kernel<32>), dim3(ceil_div(S, 256), nb * H), dim3(256), 0, st, a.k, a.v, source_mask, a.dKV, a.dKsum, inv_s,
                       S, C, H, a.dk, a.dv);
  else
    hipLaunchKernelGGL((attn_kv_bwd_kernel<16>), dim3(ceil_div(S, 256), nb * H), dim3(256), 0, st, a.k, a.v, source_mask, a.dKV, a.dKsum, inv_s,
                       S, C, H, a.dk, a.dv);
  // projections
  transpose(w->q_proj, a.wqT, C, C); transpose(w->k_proj, a.wkT, C, C); transpose(w->v_proj, a.wvT, C, C);
  LIN(a.dq, a.wqT, a.t0, T, C, C);                                     // dx (q path)
  hipLaunchKernelGGL(add_rows_kernel, g1d(T * C), dim3(256), 0, st, grad_out, a.dhcat, (long)C2, a.t0, grad_x, T, C);   // + residual + mlp x half
  LIN(a.dk, a.wkT, a.t0, Ts, C, C);
  LIN(a.dv, a.wvT, a.t1, Ts, C, C);
  hipLaunchKernelGGL(add_rows_kernel, g1d(Ts * C), dim3(256), 0, st, a.t0, a.t1, (long)C, (const float*)nullptr, grad_source, Ts, C);
  if ((rc = launch_wgrad(a.dq, C, x, C, T, gw->q_proj, a.wpart, a.wpart_floats, st))) return rc;
  if ((rc = launch_wgrad(a.dk, C, source, C, Ts, gw->k_proj, a.wpart, a.wpart_floats, st))) return rc;
  if ((rc = launch_wgrad(a.dv, C, source, C, Ts, gw->v_proj, a.wpart, a.wpart_floats, st))) return rc;
#undef LIN
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
