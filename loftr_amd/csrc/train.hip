// Training-side consumers of the matching path, forward values (their gradients: train_bwd.hip):
//   * coarse supervision      spvs_coarse, src/loftr/utils/supervision.py:22-109 (+ warp_kpts, src/loftr/utils/geometry.py:5-54)
//   * fine supervision        spvs_fine,   src/loftr/utils/supervision.py:124-142
//   * loss values             LoFTRLoss,   src/losses/loftr_loss.py:22-192 (focal / cross-entropy, sparse / dense; l2 / l2_with_std)
// The reference runs these as dozens of elementwise / indexing ATen ops over [N, L, S] volumes (conf_matrix_gt alone is
// another 92 MB per pair).  Here the supervision is two small kernels over the N (L + S) grid cells (the ground-truth
// matrix is only materialised on request) and every loss is ONE pass: the sparse losses gather the supervised entries,
// the dense ones stream conf_matrix once and correct for the positives (sum over negatives = sum over all - sum over
// positives), with fp64 block partials reduced in a fixed order (deterministic).
#include "common.h"

namespace {

struct Mat3 { float m[9]; };
// inverse of an upper-triangular-or-general 3x3 by cofactors (fp32 like the reference's K0.inverse(), geometry.py:31;
// the evaluation ORDER differs from LAPACK's LU, which is why a warped coordinate may differ in its last bits)
__device__ __forceinline__ Mat3 inv3(const float* k) {
  const float a = k[0], b = k[1], c = k[2], d = k[3], e = k[4], f = k[5], g = k[6], h = k[7], i = k[8];
  const float A = e * i - f * h, B = -(d * i - f * g), Cc = d * h - e * g;
  const float det = a * A + b * B + c * Cc;
  const float r = 1.f / det;
  Mat3 o;
  o.m[0] = A * r; o.m[1] = -(b * i - c * h) * r; o.m[2] = (b * f - c * e) * r;
  o.m[3] = B * r; o.m[4] = (a * i - c * g) * r;  o.m[5] = -(a * f - c * d) * r;
  o.m[6] = Cc * r; o.m[7] = -(a * h - b * g) * r; o.m[8] = (a * e - b * d) * r;
  return o;
}

struct SpvsGeom {
  int N, h0, w0, h1, w1;            // coarse grids
  int dh0, dw0, dh1, dw1;           // depth map sizes
  float scale;
};

// One thread per grid cell of either image: warp it into the other image (geometry.py:21-39) and record the nearest
// coarse cell there (supervision.py:66-78).  dir 0: cells of image 0 (writes w_pt0_i, nearest1), dir 1: image 1
// (writes pt1_i, nearest0).
__global__ LOFTR_NO_PACKED_FP32 void spvs_warp_kernel(SpvsGeom g, const float* __restrict__ depth0, const float* __restrict__ depth1,
                                 const float* __restrict__ T01, const float* __restrict__ T10, const float* __restrict__ K0,
                                 const float* __restrict__ K1, const float* __restrict__ scale0, const float* __restrict__ scale1,
                                 const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                 float* __restrict__ w_pt0_i, float* __restrict__ pt1_i, int* __restrict__ nearest1,
                                 int* __restrict__ nearest0) {
  const int L = g.h0 * g.w0, S = g.h1 * g.w1;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)g.N * (L + S)) return;
  const int n = (int)(idx / (L + S)), r = (int)(idx - (long)n * (L + S));
  const int dir = r >= L, c = dir ? r - L : r;
  const int ws_ = dir ? g.w1 : g.w0;
  // source / destination quantities
  const float* dep = dir ? depth1 + (long)n * g.dh1 * g.dw1 : depth0 + (long)n * g.dh0 * g.dw0;
  const int dw = dir ? g.dw1 : g.dw0, dh = dir ? g.dh1 : g.dh0;
  const float* Ks = (dir ? K1 : K0) + n * 9;
  const float* Kd = (dir ? K0 : K1) + n * 9;
  const float* T = (dir ? T10 : T01) + n * 16;
  const float* ssrc = dir ? scale1 : scale0;
  const float* sdst = dir ? scale0 : scale1;
  const uint8_t* msrc = dir ? mask1 : mask0;
  const float sx = ssrc ? g.scale * ssrc[n * 2] : g.scale, sy = ssrc ? g.scale * ssrc[n * 2 + 1] : g.scale;     // :45-46
  float x = sx * (float)(c % ws_), y = sy * (float)(c / ws_);                                                   // :51-54
  if (msrc && !msrc[(long)n * (dir ? S : L) + c]) { x = 0.f; y = 0.f; }                                          // :57-59
  if (dir) { pt1_i[((long)n * S + c) * 2] = x; pt1_i[((long)n * S + c) * 2 + 1] = y; }
  // warp_kpts
  int xi = (int)rintf(x), yi = (int)rintf(y);                             // torch.round: half to even
  xi = min(max(xi, 0), dw - 1); yi = min(max(yi, 0), dh - 1);             // (the reference would raise on an out-of-map index)
  const float d = dep[(long)yi * dw + xi];                                // :24-26
  const float hx = x * d, hy = y * d, hz = d;                             // :30
  const Mat3 Ki = inv3(Ks);
  const float cx = Ki.m[0] * hx + Ki.m[1] * hy + Ki.m[2] * hz, cy = Ki.m[3] * hx + Ki.m[4] * hy + Ki.m[5] * hz,
              cz = Ki.m[6] * hx + Ki.m[7] * hy + Ki.m[8] * hz;            // :31
  const float wx = T[0] * cx + T[1] * cy + T[2] * cz + T[3], wy = T[4] * cx + T[5] * cy + T[6] * cz + T[7],
              wz = T[8] * cx + T[9] * cy + T[10] * cz + T[11];            // :34
  const float px = Kd[0] * wx + Kd[1] * wy + Kd[2] * wz, py = Kd[3] * wx + Kd[4] * wy + Kd[5] * wz,
              pz = Kd[6] * wx + Kd[7] * wy + Kd[8] * wz;                  // :38
  const float ux = px / (pz + 1e-4f), uy = py / (pz + 1e-4f);             // :39
  if (!dir) { w_pt0_i[((long)n * L + c) * 2] = ux; w_pt0_i[((long)n * L + c) * 2 + 1] = uy; }
  // to the other image's coarse grid, rounded                              supervision.py:66-78
  const float dsx = sdst ? g.scale * sdst[n * 2] : g.scale, dsy = sdst ? g.scale * sdst[n * 2 + 1] : g.scale;
  const float qx = rintf(ux / dsx), qy = rintf(uy / dsy);
  const int wd = dir ? g.w0 : g.w1, hd = dir ? g.h0 : g.h1;
  int near = 0;
  if (qx >= 0.f && qx < (float)wd && qy >= 0.f && qy < (float)hd) near = (int)qx + (int)qy * wd;                  // else 0: :76-78
  if (dir) nearest0[(long)n * S + c] = near; else nearest1[(long)n * L + c] = near;
}

// mutual-nearest check (supervision.py:81-83) + block-local ranks for the ordered compaction
__global__ __launch_bounds__(256) void spvs_flag_kernel(SpvsGeom g, const int* __restrict__ nearest1, const int* __restrict__ nearest0,
                                                        int* __restrict__ rank, int* __restrict__ block_count) {
  const int L = g.h0 * g.w0, S = g.h1 * g.w1;
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  bool flag = false;
  if (row < (long)g.N * L) {
    const int n = (int)(row / L), i = (int)(row - (long)n * L);
    const int j = nearest1[row];
    flag = i != 0 && nearest0[(long)n * S + j] == i;
  }
  __shared__ int wave_tot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(flag);
  if (lane == 0) wave_tot[wave] = __popcll(bal);
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += wave_tot[w];
  if (row < (long)g.N * L) rank[row] = flag ? off + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
  if (threadIdx.x == 0) block_count[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

__global__ __launch_bounds__(1024) void spvs_scan_kernel(const int* __restrict__ block_count, int* __restrict__ block_off, int nblk,
                                                         int* __restrict__ total) {
  __shared__ int buf[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblk ? block_count[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblk) block_off[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void spvs_scatter_kernel(SpvsGeom g, const int* __restrict__ nearest1, const int* __restrict__ rank,
                                                           const int* __restrict__ block_off, int64_t* __restrict__ b_ids,
                                                           int64_t* __restrict__ i_ids, int64_t* __restrict__ j_ids,
                                                           float* __restrict__ conf_gt) {
  const int L = g.h0 * g.w0, S = g.h1 * g.w1;
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= (long)g.N * L) return;
  const int rk = rank[row];
  if (rk < 0) return;
  const long dst = block_off[blockIdx.x] + rk;
  const int n = (int)(row / L), i = (int)(row - (long)n * L), j = nearest1[row];
  b_ids[dst] = n; i_ids[dst] = i; j_ids[dst] = j;
  if (conf_gt) conf_gt[((long)n * L + i) * S + j] = 1.f;                  // :85-89 (the caller zeroed the volume)
}

__global__ void spvs_fine_kernel(const float* __restrict__ w_pt0_i, const float* __restrict__ pt1_i, int L, int S,
                                 const int64_t* __restrict__ b, const int64_t* __restrict__ i, const int64_t* __restrict__ j,
                                 long M, float scale, float radius, const float* __restrict__ scale1, float* __restrict__ out) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const long bb = b[m];
  const float sx = scale1 ? scale * scale1[bb * 2] : scale, sy = scale1 ? scale * scale1[bb * 2 + 1] : scale;   // :137
  const float* p = w_pt0_i + (bb * L + i[m]) * 2;
  const float* q = pt1_i + (bb * S + j[m]) * 2;
  out[m * 2] = (p[0] - q[0]) / sx / radius;                               // :139
  out[m * 2 + 1] = (p[1] - q[1]) / sy / radius;
}

// ---- losses: fp64 block partials, reduced in a fixed order ------------------------------------------------------
__device__ __forceinline__ void block_partial(double v, double* __restrict__ part) {
  __shared__ double sm[256];
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double sm[256];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

__device__ __forceinline__ double clampc(float c) { return (double)fminf(fmaxf(c, 1e-6f), 1.f - 1e-6f); }      // torch.clamp(conf, 1e-6, 1-1e-6)
// mode 0: -alpha (1-p)^gamma log p (focal, supervised entry)   1: -alpha p^gamma log(1-p) (focal, dense negative)
// mode 2: -log p (CE positive)                                  3: -log(1-p) (CE negative)
__device__ __forceinline__ double loss_term(double p, int mode, double alpha, double gamma) {
  switch (mode) {
    case 0: return -alpha * pow(1.0 - p, gamma) * log(p);
    case 1: return -alpha * pow(p, gamma) * log(1.0 - p);
    case 2: return -log(p);
    default: return -log(1.0 - p);
  }
}

// sum over the listed entries (b, i, j) of term(conf[b, i, j]) * weight;  conf has row pitch ldS and pair pitch ldL * ldS
__global__ __launch_bounds__(256) void loss_gather_kernel(const float* __restrict__ conf, long ldL, long ldS, const int64_t* __restrict__ b,
                                                          const int64_t* __restrict__ i, const int64_t* __restrict__ j, long M, int L, int S,
                                                          const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1, int mode,
                                                          double alpha, double gamma, double* __restrict__ part) {
  double s = 0;
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const long bb = b[m], ii = i[m], jj = j[m];
    const double w = mask0 ? (double)((mask0[bb * L + ii] != 0) && (mask1[bb * S + jj] != 0)) : 1.0;
    s += loss_term(clampc(conf[(bb * ldL + ii) * ldS + jj]), mode, alpha, gamma) * w;
  }
  block_partial(s, part);
}

// sum over ALL entries of the [N, L, S] volume of term(conf) * (mask0 x mask1)
__global__ __launch_bounds__(256) void loss_dense_kernel(const float* __restrict__ conf, int N, int L, int S,
                                                         const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1, int mode,
                                                         double alpha, double gamma, double* __restrict__ part) {
  double s = 0;
  const long rows = (long)N * L;
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = (int)(row / L);
    if (mask0 && !mask0[row]) continue;
    const float* cr = conf + row * S;
    for (int jj = threadIdx.x; jj < S; jj += 256) {
      if (mask1 && !mask1[(long)n * S + jj]) continue;
      s += loss_term(clampc(cr[jj]), mode, alpha, gamma);
    }
  }
  block_partial(s, part);
}

// sparse Sinkhorn negatives (loftr_loss.py:63-79): dustbin entries of the rows / columns without a ground-truth match.
//   has0 [N, L], has1 [N, S]: 1 where a GT match exists;  any0 [N], any1 [N]: some cell of the pair is unmasked.
//   part[2 b] = sum of terms, part[2 b + 1] = number of terms
__global__ __launch_bounds__(256) void loss_bins_kernel(const float* __restrict__ conf_bin, int N, int L, int S,
                                                        const uint8_t* __restrict__ has0, const uint8_t* __restrict__ has1,
                                                        const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                        const uint8_t* __restrict__ any0, const uint8_t* __restrict__ any1,
                                                        double alpha, double gamma, double* __restrict__ part, double* __restrict__ cnt) {
  double s = 0, c = 0;
  const long tot = (long)N * (L + S);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (long)gridDim.x * 256) {
    const int n = (int)(idx / (L + S)), r = (int)(idx - (long)n * (L + S));
    const float* cb = conf_bin + (long)n * (L + 1) * (S + 1);
    if (r < L) {                                   // conf[:, :-1, -1][neg0], kept iff weight.sum(-1) != 0
      if (has0[(long)n * L + r]) continue;
      if (mask0 && !(mask0[(long)n * L + r] && any1[n])) continue;
      s += loss_term(clampc(cb[(long)r * (S + 1) + S]), 0, alpha, gamma); c += 1;
    } else {                                       // conf[:, -1, :-1][neg1], kept iff weight.sum(1) != 0
      const int jj = r - L;
      if (has1[(long)n * S + jj]) continue;
      if (mask1 && !(mask1[(long)n * S + jj] && any0[n])) continue;
      s += loss_term(clampc(cb[(long)L * (S + 1) + jj]), 0, alpha, gamma); c += 1;
    }
  }
  block_partial(s, part);
  __syncthreads();
  block_partial(c, cnt);
}

__global__ void mark_gt_kernel(const int64_t* __restrict__ b, const int64_t* __restrict__ i, const int64_t* __restrict__ j, long M, int L,
                               int S, uint8_t* __restrict__ has0, uint8_t* __restrict__ has1) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  has0[b[m] * L + i[m]] = 1;
  has1[b[m] * S + j[m]] = 1;
}
__global__ void any_mask_kernel(const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1, int L, int S,
                                uint8_t* __restrict__ any0, uint8_t* __restrict__ any1) {
  const int n = blockIdx.x;
  __shared__ int a0, a1;
  if (threadIdx.x == 0) { a0 = 0; a1 = 0; }
  __syncthreads();
  for (int k = threadIdx.x; k < L; k += blockDim.x) if (mask0[(long)n * L + k]) a0 = 1;
  for (int k = threadIdx.x; k < S; k += blockDim.x) if (mask1[(long)n * S + k]) a1 = 1;
  __syncthreads();
  if (threadIdx.x == 0) { any0[n] = (uint8_t)a0; any1[n] = (uint8_t)a1; }
}

// fine loss terms (loftr_loss.py:108-157).  part0: sum of offset_l2 [* 1/std] over the correct entries, part1: their
// number, part2: sum of 1/clamp(std) over ALL entries (the weight normaliser)
__global__ __launch_bounds__(256) void fine_loss_kernel(const float* __restrict__ expec_f, int ld, const float* __restrict__ gt, long M,
                                                        int with_std, float thr, double* __restrict__ p0, double* __restrict__ p1,
                                                        double* __restrict__ p2) {
  double s = 0, c = 0, w = 0;
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const float gx = gt[m * 2], gy = gt[m * 2 + 1];
    const double inv = with_std ? 1.0 / (double)fmaxf(expec_f[m * ld + 2], 1e-10f) : 1.0;
    w += inv;
    if (fmaxf(fabsf(gx), fabsf(gy)) < thr) {
      const double dx = (double)gx - (double)expec_f[m * ld], dy = (double)gy - (double)expec_f[m * ld + 1];
      s += (dx * dx + dy * dy) * inv; c += 1;
    }
  }
  block_partial(s, p0);
  __syncthreads();
  block_partial(c, p1);
  __syncthreads();
  block_partial(w, p2);
}

constexpr int LOSS_BLOCKS = 1024;

}  // namespace

extern "C" size_t loftr_spvs_coarse_workspace_bytes(int N, int L, int S) {
  if (N <= 0 || L <= 0 || S <= 0) return 0;
  const size_t NL = (size_t)N * L, NS = (size_t)N * S, nblk = (NL + 255) / 256;
  return align_up(NL * 4, 256) * 2 + align_up(NS * 4, 256) + align_up(nblk * 4, 256) * 2 + 1024;
}

extern "C" int loftr_spvs_coarse(const loftr_spvs_params* p, float* w_pt0_i, float* pt1_i, int64_t* spv_b, int64_t* spv_i,
                                 int64_t* spv_j, int32_t* count, float* conf_gt, void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(p && w_pt0_i && pt1_i && spv_b && spv_i && spv_j && count && ws);
  LOFTR_CHECK_ARG(p->N > 0 && p->scale > 0 && p->H0 > 0 && p->W0 > 0 && p->H1 > 0 && p->W1 > 0 && p->depth0 && p->depth1 &&
                  p->T_0to1 && p->T_1to0 && p->K0 && p->K1 && (p->mask0 == nullptr) == (p->mask1 == nullptr) &&
                  (p->scale0 == nullptr) == (p->scale1 == nullptr));
  hipStream_t st = (hipStream_t)stream;
  SpvsGeom g{p->N, p->H0 / p->scale, p->W0 / p->scale, p->H1 / p->scale, p->W1 / p->scale, p->dh0, p->dw0, p->dh1, p->dw1, (float)p->scale};
  const int L = g.h0 * g.w0, S = g.h1 * g.w1;
  const long NL = (long)g.N * L;
  const int nblk = (int)((NL + 255) / 256);
  WsAlloc wa(ws, ws_bytes);
  int* nearest1 = wa.take<int>(NL);
  int* rank = wa.take<int>(NL);
  int* nearest0 = wa.take<int>((size_t)g.N * S);
  int* bcount = wa.take<int>(nblk);
  int* boff = wa.take<int>(nblk);
  if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
  const long cells = (long)g.N * (L + S);
  hipLaunchKernelGGL(spvs_warp_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, g, p->depth0, p->depth1, p->T_0to1,
                     p->T_1to0, p->K0, p->K1, p->scale0, p->scale1, p->mask0, p->mask1, w_pt0_i, pt1_i, nearest1, nearest0);
  hipLaunchKernelGGL(spvs_flag_kernel, dim3(nblk), dim3(256), 0, st, g, nearest1, nearest0, rank, bcount);
  hipLaunchKernelGGL(spvs_scan_kernel, dim3(1), dim3(1024), 0, st, bcount, boff, nblk, count);
  if (conf_gt) (void)hipMemsetAsync(conf_gt, 0, sizeof(float) * (size_t)g.N * L * S, st);
  hipLaunchKernelGGL(spvs_scatter_kernel, dim3(nblk), dim3(256), 0, st, g, nearest1, rank, boff, spv_b, spv_i, spv_j, conf_gt);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

extern "C" int loftr_spvs_fine(const float* w_pt0_i, const float* pt1_i, int L, int S, const int64_t* b_ids, const int64_t* i_ids,
                               const int64_t* j_ids, long M, float scale, float radius, const float* scale1, float* expec_f_gt,
                               void* stream) {
  LOFTR_CHECK_ARG(w_pt0_i && pt1_i && L > 0 && S > 0 && M >= 0 && scale > 0.f && radius > 0.f);
  if (M == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(b_ids && i_ids && j_ids && expec_f_gt);
  hipLaunchKernelGGL(spvs_fine_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_pt0_i, pt1_i, L, S, b_ids,
                     i_ids, j_ids, M, scale, radius, scale1, expec_f_gt);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

extern "C" size_t loftr_loss_workspace_bytes(int N, int L, int S) {
  if (N <= 0 || L <= 0 || S <= 0) return 0;
  return (size_t)LOSS_BLOCKS * 8 * 8 + align_up((size_t)N * L, 256) + align_up((size_t)N * S, 256) + align_up((size_t)N, 256) * 2 + 2048;
}

// kind 0: sparse focal, dual-softmax (positives only)      1: sparse focal, Sinkhorn (conf = conf_matrix_with_bin)
// kind 2: dense focal                                       3: dense cross-entropy
// sums[0] = sum of the positive terms over the M ground-truth entries (times mask0 x mask1)
// sums[2] = kind 1: sum over the supervised dustbin entries, sums[3] = their number;
//           kind 2 / 3: sum of the NEGATIVE term over ALL entries of the volume, sums[3] = the same term gathered at the
//           positives (the caller subtracts: negatives = all - positives; their number is N L S - M)
extern "C" int loftr_coarse_loss_sums(const float* conf, int N, int L, int S, int kind, const int64_t* gt_b, const int64_t* gt_i,
                                      const int64_t* gt_j, long M, const uint8_t* mask0, const uint8_t* mask1, float alpha,
                                      float gamma, double* sums, void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(conf && sums && ws && N > 0 && L > 0 && S > 0 && M >= 0 && kind >= 0 && kind <= 3 && (M == 0 || (gt_b && gt_i && gt_j)));
  LOFTR_CHECK_ARG((mask0 == nullptr) == (mask1 == nullptr));
  hipStream_t st = (hipStream_t)stream;
  WsAlloc wa(ws, ws_bytes);
  double* part = wa.take<double>((size_t)LOSS_BLOCKS * 4);
  uint8_t* has0 = wa.take<uint8_t>((size_t)N * L);
  uint8_t* has1 = wa.take<uint8_t>((size_t)N * S);
  uint8_t* any0 = wa.take<uint8_t>(N);
  uint8_t* any1 = wa.take<uint8_t>(N);
  if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
  (void)hipMemsetAsync(sums, 0, sizeof(double) * 4, st);
  const bool bins = kind == 1;                     // conf is conf_matrix_with_bin [N, L+1, S+1]
  const long ldL = bins ? L + 1 : L, ldS = bins ? S + 1 : S;
  const int pos_mode = kind == 3 ? 2 : 0, neg_mode = kind == 3 ? 3 : 1;
  const int gb = (int)((M + 255) / 256) < LOSS_BLOCKS ? (int)((M + 255) / 256) : LOSS_BLOCKS;
  if (M > 0) {
    hipLaunchKernelGGL(loss_gather_kernel, dim3(gb), dim3(256), 0, st, conf, ldL, ldS, gt_b, gt_i, gt_j, M, L, S, mask0, mask1, pos_mode,
                       (double)alpha, (double)gamma, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, part, gb, sums);
  }
  if (kind == 1) {
    (void)hipMemsetAsync(has0, 0, (size_t)N * L, st);
    (void)hipMemsetAsync(has1, 0, (size_t)N * S, st);
    if (M > 0) hipLaunchKernelGGL(mark_gt_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, gt_b, gt_i, gt_j, M, L, S, has0, has1);
    if (mask0) hipLaunchKernelGGL(any_mask_kernel, dim3(N), dim3(256), 0, st, mask0, mask1, L, S, any0, any1);
    const long tot = (long)N * (L + S);
    const int nb = (int)((tot + 255) / 256) < LOSS_BLOCKS ? (int)((tot + 255) / 256) : LOSS_BLOCKS;
    hipLaunchKernelGGL(loss_bins_kernel, dim3(nb), dim3(256), 0, st, conf, N, L, S, has0, has1, mask0, mask1, any0, any1, (double)alpha,
                       (double)gamma, part + LOSS_BLOCKS, part + 2 * LOSS_BLOCKS);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, part + LOSS_BLOCKS, nb, sums + 2);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, part + 2 * LOSS_BLOCKS, nb, sums + 3);
  } else if (kind >= 2) {
    // negatives = all entries - positives: one pass over the volume with the negative formula, minus the same formula
    // gathered at the positives
    hipLaunchKernelGGL(loss_dense_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, conf, N, L, S, mask0, mask1, neg_mode, (double)alpha,
                       (double)gamma, part + LOSS_BLOCKS);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, part + LOSS_BLOCKS, LOSS_BLOCKS, sums + 2);
    if (M > 0) {
      hipLaunchKernelGGL(loss_gather_kernel, dim3(gb), dim3(256), 0, st, conf, ldL, ldS, gt_b, gt_i, gt_j, M, L, S, mask0, mask1, neg_mode,
                         (double)alpha, (double)gamma, part + 2 * LOSS_BLOCKS);
      hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, part + 2 * LOSS_BLOCKS, gb, sums + 3);     // to subtract (host side)
    }
  }
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// sums[0] = sum of the (weighted) squared offsets over the correct entries, sums[1] = their number,
// sums[2] = sum of 1 / clamp(std, 1e-10) over all M entries (l2_with_std) or M (l2)
extern "C" int loftr_fine_loss_sums(const float* expec_f, int ld, const float* expec_f_gt, long M, int with_std, float correct_thr,
                                    double* sums, void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(sums && ws && M >= 0 && (M == 0 || (expec_f && expec_f_gt)) && ld >= (with_std ? 3 : 2));
  hipStream_t st = (hipStream_t)stream;
  WsAlloc wa(ws, ws_bytes);
  double* part = wa.take<double>((size_t)LOSS_BLOCKS * 3);
  if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
  (void)hipMemsetAsync(sums, 0, sizeof(double) * 3, st);
  if (M == 0) return LOFTR_OK;
  const int nb = (int)((M + 255) / 256) < LOSS_BLOCKS ? (int)((M + 255) / 256) : LOSS_BLOCKS;
  hipLaunchKernelGGL(fine_loss_kernel, dim3(nb), dim3(256), 0, st, expec_f, ld, expec_f_gt, M, with_std, correct_thr, part,
                     part + LOSS_BLOCKS, part + 2 * LOSS_BLOCKS);
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, part + k * LOSS_BLOCKS, nb, sums + k);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
