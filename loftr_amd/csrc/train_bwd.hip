// Backward passes of the matching heads and of the losses that read them (second half):
//   * d loss_c / d conf_matrix      LoFTRLoss.compute_coarse_loss,  src/losses/loftr_loss.py:22-99   (all four kinds of train.hip)
//   * d loss_f / d expec_f          LoFTRLoss._compute_fine_loss_*, src/losses/loftr_loss.py:108-157
//   * d expec_f / d feat_f0, feat_f1   FineMatching.forward,        src/loftr/utils/fine_matching.py:43-57
// (d conf_matrix / d sim_matrix of the dual-softmax and the Sinkhorn backward are loftr_dual_softmax_bwd / loftr_sinkhorn_bwd in
// coarse_match.hip: they share the forward's descriptor staging, score sweeps and iteration kernels.)  What torch.autograd does for the reference as a chain of a dozen ATen backward nodes
// is ONE kernel per head here: each recomputes the forward quantities it needs from the head's inputs (nothing but the
// inputs is kept alive between forward and backward) and writes the gradient in a single pass.
// The chain stops at the heads' inputs -- the transformer outputs: the transformers, FinePreprocess and the backbone
// have no backward.
#include "common.h"

namespace {

__device__ __forceinline__ bool clamp_open(float c) { return c >= 1e-6f && c <= 1.f - 1e-6f; }   // torch.clamp passes the gradient inside [min, max]
__device__ __forceinline__ double clampd(float c) { return (double)fminf(fmaxf(c, 1e-6f), 1.f - 1e-6f); }

// d term / d p of train.hip's loss_term (same modes)
__device__ __forceinline__ double loss_term_grad(double p, int mode, double alpha, double gamma) {
  switch (mode) {
    case 0: return alpha * (gamma * pow(1.0 - p, gamma - 1.0) * log(p) - pow(1.0 - p, gamma) / p);
    case 1: return -alpha * (gamma * pow(p, gamma - 1.0) * log(1.0 - p) - pow(p, gamma) / (1.0 - p));
    case 2: return -1.0 / p;
    default: return 1.0 / (1.0 - p);
  }
}

// dense kinds: every entry of the volume first receives the NEGATIVE term's gradient (times mask0 x mask1) ...
__global__ __launch_bounds__(256) void loss_grad_dense_kernel(const float* __restrict__ conf, int N, int L, int S,
                                                              const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                              int mode, double alpha, double gamma, double scale,
                                                              float* __restrict__ grad) {
  const long rows = (long)N * L;
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = (int)(row / L);
    const bool r_ok = !mask0 || mask0[row];
    const float* cr = conf + row * S;
    float* gr = grad + row * S;
    for (int j = threadIdx.x; j < S; j += 256) {
      float g = 0.f;
      if (r_ok && (!mask1 || mask1[(long)n * S + j])) {
        const float c = cr[j];
        if (clamp_open(c)) g = (float)(scale * loss_term_grad(clampd(c), mode, alpha, gamma));
      }
      gr[j] = g;
    }
  }
}

// ... then the supervised entries are OVERWRITTEN with the positive term's gradient (ids are unique: they come from a
// boolean mask, loftr_loss.py:29).  Sparse kind 0: the volume was zero-filled before.
__global__ __launch_bounds__(256) void loss_grad_gather_kernel(const float* __restrict__ conf, long ldL, long ldS, const int64_t* __restrict__ b,
                                                               const int64_t* __restrict__ i, const int64_t* __restrict__ j, long M, int L,
                                                               int S, const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                               int mode, double alpha, double gamma, double scale,
                                                               float* __restrict__ grad) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const long bb = b[m], ii = i[m], jj = j[m];
  const bool w = !mask0 || (mask0[bb * L + ii] != 0 && mask1[bb * S + jj] != 0);
  const long o = (bb * ldL + ii) * ldS + jj;
  const float c = conf[o];
  grad[o] = (w && clamp_open(c)) ? (float)(scale * loss_term_grad(clampd(c), mode, alpha, gamma)) : 0.f;
}

// sparse Sinkhorn 'negatives' (loftr_loss.py:63-79, train.hip:loss_bins_kernel): the dustbin entry of every row / column without a
// ground-truth match, kept iff the row / column carries some loss weight; same focal form as the positives (mode 0).
__global__ void mark_gt_bwd_kernel(const int64_t* __restrict__ b, const int64_t* __restrict__ i, const int64_t* __restrict__ j, long M, int L,
                                   int S, uint8_t* __restrict__ has0, uint8_t* __restrict__ has1) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  has0[b[m] * L + i[m]] = 1;
  has1[b[m] * S + j[m]] = 1;
}
__global__ void any_mask_bwd_kernel(const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1, int L, int S,
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
__global__ __launch_bounds__(256) void loss_grad_bins_kernel(const float* __restrict__ conf_bin, int N, int L, int S,
                                                             const uint8_t* __restrict__ has0, const uint8_t* __restrict__ has1,
                                                             const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                             const uint8_t* __restrict__ any0, const uint8_t* __restrict__ any1,
                                                             double alpha, double gamma, double scale, float* __restrict__ grad) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * (L + S)) return;
  const int n = (int)(idx / (L + S)), r = (int)(idx - (long)n * (L + S));
  long o;
  if (r < L) {
    if (has0[(long)n * L + r] || (mask0 && !(mask0[(long)n * L + r] && any1[n]))) return;
    o = ((long)n * (L + 1) + r) * (S + 1) + S;
  } else {
    const int jj = r - L;
    if (has1[(long)n * S + jj] || (mask1 && !(mask1[(long)n * S + jj] && any0[n]))) return;
    o = ((long)n * (L + 1) + L) * (S + 1) + jj;
  }
  const float c = conf_bin[o];
  if (clamp_open(c)) grad[o] = (float)(scale * loss_term_grad(clampd(c), 0, alpha, gamma));
}

// d loss_f / d expec_f.  sums = loftr_fine_loss_sums' output of the forward (device): [1] = number of correct entries,
// [2] = sum of 1 / clamp(std) over all M entries.  The std column receives no gradient: the weight is .detach()ed (:131).
__global__ __launch_bounds__(256) void fine_loss_grad_kernel(const float* __restrict__ expec_f, int ld, const float* __restrict__ gt, long M,
                                                             int with_std, float thr, int training, const double* __restrict__ sums,
                                                             float upstream, float* __restrict__ grad) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const double n_ok = sums[1];
  float gx = 0.f, gy = 0.f;
  const float tx = gt[m * 2], ty = gt[m * 2 + 1];
  if (n_ok > 0) {
    if (fmaxf(fabsf(tx), fabsf(ty)) < thr) {
      const double w = with_std ? (1.0 / (double)fmaxf(expec_f[m * ld + 2], 1e-10f)) * ((double)M / sums[2]) : 1.0;
      const double k = -2.0 * w / n_ok * (double)upstream;
      gx = (float)(k * ((double)tx - (double)expec_f[m * ld]));
      gy = (float)(k * ((double)ty - (double)expec_f[m * ld + 1]));
    }
  } else if (training && !with_std && m == 0) {      // plain l2 without a correct match: the false supervision of entry 0 (:113-117)
    gx = -2.f * upstream * (tx - expec_f[0]);        // (l2_with_std: weight[0] = 0, no gradient, :138-143)
    gy = -2.f * upstream * (ty - expec_f[1]);
  }
  grad[m * ld] = gx; grad[m * ld + 1] = gy;
  for (int k = 2; k < ld; ++k) grad[m * ld + k] = 0.f;
}

// FineMatching backward: one wave per match, lane r < WW owns window position r (the forward kernel's layout, fine.hip).
//   expec_f = (E[x], E[y], sum_a sqrt(clamp(Var_a, 1e-10))) of heat = softmax(<f0[centre], f1[r]> / sqrt(C))
//   g = d L / d expec_f [M, 3]  ->  d L / d feat_f1 [M, WW, C] = dsim_r f0[centre],
//                                   d L / d feat_f0 [M, WW, C] = (sum_r dsim_r f1[r]) at the centre row, 0 elsewhere
//   grid (ceil(M / 4)), 256 threads.
__global__ __launch_bounds__(256) void fine_match_bwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1, int M, int WW,
                                                             int W, int C, const float* __restrict__ g, float* __restrict__ d0,
                                                             float* __restrict__ d1) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long m = (long)blockIdx.x * 4 + wave;
  if (m >= M) return;
  const float* p = f0 + (m * WW + WW / 2) * C;
  const float* q = f1 + m * WW * C;
  float sim = 0.f;
  for (int r = 0; r < WW; ++r) {                           // the forward's arithmetic, same order (fine.hip:fine_match_kernel)
    float part = 0.f;
    for (int c = lane; c < C; c += 64) part += p[c] * q[(long)r * C + c];
    part = wave_sum(part);
    if (lane == r) sim = part;
  }
  const bool act = lane < WW;
  const float temp = 1.f / sqrtf((float)C);
  const float t = act ? sim * temp : -3.0e38f;
  const float mx = wave_max(t);
  const float e = act ? expf(t - mx) : 0.f;
  const float heat = e / wave_sum(e);
  const float gx = act ? 2.f * (float)(lane % W) / (float)(W - 1) - 1.f : 0.f;
  const float gy = act ? 2.f * (float)(lane / W) / (float)(W - 1) - 1.f : 0.f;
  const float cx = wave_sum(heat * gx), cy = wave_sum(heat * gy);
  const float vx = wave_sum(heat * gx * gx) - cx * cx;
  const float vy = wave_sum(heat * gy * gy) - cy * cy;
  // std = sqrt(clamp(vx)) + sqrt(clamp(vy)): d std / d v = 1 / (2 sqrt(v)) where the clamp is open (v >= 1e-10)
  const float g0 = g[m * 3], g1 = g[m * 3 + 1], g2 = g[m * 3 + 2];
  const float hx = vx >= 1e-10f ? g2 * 0.5f / sqrtf(vx) : 0.f;
  const float hy = vy >= 1e-10f ? g2 * 0.5f / sqrtf(vy) : 0.f;
  // d L / d heat_r = (g0 - 2 cx hx) gx + hx gx^2 + (g1 - 2 cy hy) gy + hy gy^2
  const float dh = (g0 - 2.f * cx * hx) * gx + hx * gx * gx + (g1 - 2.f * cy * hy) * gy + hy * gy * gy;
  const float dot = wave_sum(heat * dh);
  const float ds = act ? temp * heat * (dh - dot) : 0.f;  // softmax backward, times softmax_temp
  float* o0 = d0 + m * WW * C;
  float* o1 = d1 + m * WW * C;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    const float pc = c < C ? p[c] : 0.f;
    float acc = 0.f;
    for (int r = 0; r < WW; ++r) {
      const float dsr = __shfl(ds, r, 64);
      if (c < C) {
        o1[(long)r * C + c] = dsr * pc;
        acc += dsr * q[(long)r * C + c];
        if (r != WW / 2) o0[(long)r * C + c] = 0.f;
      }
    }
    if (c < C) o0[(long)(WW / 2) * C + c] = acc;
  }
}

}  // namespace

// grad_conf = d (pos_scale * sum_pos + neg_scale * sum_neg) / d conf, the sums being loftr_coarse_loss_sums' (same kind, ids and
// masks): [N, L, S] for kinds 0, 2, 3; [N, L+1, S+1] (conf_matrix_with_bin) for kind 1.  The caller folds the means, loss weights,
// corner cases and the upstream gradient into the scales: pos_scale = upstream c_pos_w / M (0 without ground truth), neg_scale =
// upstream c_neg_w / (N L S - M) (dense kinds) or / sums[3] (kind 1: the number of supervised dustbin entries).
// Workspace: loftr_loss_workspace_bytes(N, L, S) (kind 1 only; may be null otherwise).
extern "C" int loftr_coarse_loss_grad(const float* conf, int N, int L, int S, int kind, const int64_t* gt_b, const int64_t* gt_i,
                                      const int64_t* gt_j, long M, const uint8_t* mask0, const uint8_t* mask1, float alpha,
                                      float gamma, double pos_scale, double neg_scale, float* grad_conf, void* ws, size_t ws_bytes,
                                      void* stream) {
  LOFTR_CHECK_ARG(conf && grad_conf && N > 0 && L > 0 && S > 0 && M >= 0 && kind >= 0 && kind <= 3 && (M == 0 || (gt_b && gt_i && gt_j)));
  LOFTR_CHECK_ARG((mask0 == nullptr) == (mask1 == nullptr));
  hipStream_t st = (hipStream_t)stream;
  const bool bins = kind == 1;
  const long ldL = bins ? L + 1 : L, ldS = bins ? S + 1 : S;
  const int pos_mode = kind == 3 ? 2 : 0, neg_mode = kind == 3 ? 3 : 1;
  if (kind <= 1) (void)hipMemsetAsync(grad_conf, 0, sizeof(float) * (size_t)N * ldL * ldS, st);
  else hipLaunchKernelGGL(loss_grad_dense_kernel, dim3(1024), dim3(256), 0, st, conf, N, L, S, mask0, mask1, neg_mode, (double)alpha,
                          (double)gamma, neg_scale, grad_conf);
  if (M > 0)
    hipLaunchKernelGGL(loss_grad_gather_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, conf, ldL, ldS, gt_b, gt_i, gt_j, M, L, S,
                       mask0, mask1, pos_mode, (double)alpha, (double)gamma, pos_scale, grad_conf);
  if (bins) {
    LOFTR_CHECK_ARG(ws != nullptr);
    WsAlloc wa(ws, ws_bytes);
    uint8_t* has0 = wa.take<uint8_t>((size_t)N * L);
    uint8_t* has1 = wa.take<uint8_t>((size_t)N * S);
    uint8_t* any0 = wa.take<uint8_t>(N);
    uint8_t* any1 = wa.take<uint8_t>(N);
    if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
    (void)hipMemsetAsync(has0, 0, (size_t)N * L, st);
    (void)hipMemsetAsync(has1, 0, (size_t)N * S, st);
    if (M > 0) hipLaunchKernelGGL(mark_gt_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, gt_b, gt_i, gt_j, M, L, S, has0, has1);
    if (mask0) hipLaunchKernelGGL(any_mask_bwd_kernel, dim3(N), dim3(256), 0, st, mask0, mask1, L, S, any0, any1);
    const long tot = (long)N * (L + S);
    hipLaunchKernelGGL(loss_grad_bins_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, conf, N, L, S, has0, has1, mask0, mask1,
                       any0, any1, (double)alpha, (double)gamma, neg_scale, grad_conf);
  }
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

extern "C" int loftr_fine_loss_grad(const float* expec_f, int ld, const float* expec_f_gt, long M, int with_std, float correct_thr,
                                    int training, const double* sums, float upstream, float* grad_expec, void* stream) {
  LOFTR_CHECK_ARG(M >= 0 && ld >= (with_std ? 3 : 2));
  if (M == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(expec_f && expec_f_gt && sums && grad_expec);
  hipLaunchKernelGGL(fine_loss_grad_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, expec_f, ld, expec_f_gt, M,
                     with_std, correct_thr, training, sums, upstream, grad_expec);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

extern "C" int loftr_fine_match_bwd(const float* feat_f0, const float* feat_f1, int M, int WW, int C, const float* grad_expec,
                                    float* grad_f0, float* grad_f1, void* stream) {
  LOFTR_CHECK_ARG(M >= 0 && WW > 0 && C > 0);
  if (M == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(feat_f0 && feat_f1 && grad_expec && grad_f0 && grad_f1);
  int W = 1;
  while (W * W < WW) ++W;
  if (W * W != WW || WW > 64 || W < 2) return LOFTR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fine_match_bwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, feat_f0, feat_f1, M, WW, W, C,
                     grad_expec, grad_f0, grad_f1);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
