// Backward of FinePreprocess.
//   reference: src/loftr/loftr_module/fine_preprocess.py:29-59 under torch.autograd --
//     win = unfold(feat_f, W, stride, pad W/2)[b_ids, ids]                      [M, WW, Cf]   per side
//     cwin = down_proj(cat[feat_c0[b, i], feat_c1[b, j]])                       [2M, Cf]
//     out = merge_feat(cat[cat[win0, win1], repeat(cwin, ww)], -1)              [2M, WW, Cf]
// From dY = d out:  dXcat = dY Wm,  dWm = dY^T Xcat,  dbm = sum dY;  d win = dXcat[..., :Cf] scattered back into the fine maps
// (overlapping 5 x 5 windows at stride 4: atomic adds, like torch's index / fold backward on a GPU);  d cwin = sum_ww dXcat[..., Cf:];
// d cc = d cwin Wd,  dWd = d cwin^T cc,  dbd = sum d cwin;  d cc scattered into feat_c0 / feat_c1 at (b, i) / (b, j).
// Windows and gathered coarse rows are recomputed.  GEMMs: loftr_linear_fwd / launch_wgrad (split-fp16 MFMA); the rest fp32 vector code.
#include "linear.h"
#include "head_grads.h"

extern "C" size_t loftr_linear_workspace_bytes(int M, int N, int K);
extern "C" int loftr_linear_fwd(const float* a, const float* w, float* out, int M, int N, int K, void* ws, size_t ws_bytes, void* stream);

namespace {
namespace fb {
// Xcat[(side M + m) WW + ww] = [window pixel (Cf), cwin (Cf)]; also used (cwin == null) to fill only the window half
__global__ void build_xcat_kernel(loftr_fmap f0, loftr_fmap f1, const int64_t* __restrict__ b_ids, const int64_t* __restrict__ i_ids,
                                  const int64_t* __restrict__ j_ids, int M, int w0c, int w1c, int stride, int W, int Cf,
                                  const float* __restrict__ cwin, float* __restrict__ xcat) {
  const int m = blockIdx.x, side = blockIdx.y;
  const loftr_fmap f = side ? f1 : f0;
  const int wc = side ? w1c : w0c;
  const long cell = side ? j_ids[m] : i_ids[m];
  const long b = b_ids[m];
  const int cy = (int)(cell / wc) * stride, cx = (int)(cell % wc) * stride, r = W / 2, WW = W * W;
  float* out = xcat + ((long)side * M + m) * WW * 2 * Cf;
  const float* cw = cwin + ((long)side * M + m) * Cf;
  for (int i = threadIdx.x; i < WW * Cf; i += blockDim.x) {
    const int ww = i / Cf, c = i - ww * Cf;
    const int y = cy + ww / W - r, x = cx + ww % W - r;
    float v = 0.f;
    if (y >= 0 && y < f.H && x >= 0 && x < f.W) v = f.data[b * f.sn + (long)c * f.sc + (long)y * f.sh + (long)x * f.sw];
    out[(long)ww * 2 * Cf + c] = v;
    out[(long)ww * 2 * Cf + Cf + c] = cw[c];
  }
}
// cc[(side M + m)] = feat_c{side}[b, id]
__global__ void gather_cc_kernel(const float* __restrict__ fc0, const float* __restrict__ fc1, const int64_t* __restrict__ b_ids,
                                 const int64_t* __restrict__ i_ids, const int64_t* __restrict__ j_ids, int M, int L, int S, int Cc,
                                 float* __restrict__ cc) {
  const long m = blockIdx.x;
  const int side = blockIdx.y;
  const long row = side ? b_ids[m] * S + j_ids[m] : b_ids[m] * L + i_ids[m];
  const float* src = (side ? fc1 : fc0) + row * Cc;
  for (int c = threadIdx.x; c < Cc; c += blockDim.x) cc[((long)side * M + m) * Cc + c] = src[c];
}
__global__ void add_bias_kernel(float* __restrict__ x, const float* __restrict__ b, long rows, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * C) x[i] += b[i % C];
}
__global__ void stack2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {   // out = [a; b]
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * n) out[i] = i < n ? a[i] : b[i - n];
}
// dcwin[r][c] = sum_ww dxcat[(r WW + ww)][Cf + c]   (ascending ww)
__global__ void sum_ctx_kernel(const float* __restrict__ dxcat, long rows, int WW, int Cf, float* __restrict__ dcwin) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Cf) return;
  const long r = i / Cf; const int c = (int)(i - r * Cf);
  float s = 0.f;
  for (int ww = 0; ww < WW; ++ww) s += dxcat[(r * WW + ww) * 2 * Cf + Cf + c];
  dcwin[i] = s;
}
// d feat_f{side}[b, c, y, x] += dxcat[(side M + m) WW + ww][c]   for window pixels inside the map
__global__ void scatter_windows_kernel(loftr_fmap g0, loftr_fmap g1, const int64_t* __restrict__ b_ids, const int64_t* __restrict__ i_ids,
                                       const int64_t* __restrict__ j_ids, int M, int w0c, int w1c, int stride, int W, int Cf,
                                       const float* __restrict__ dxcat) {
  const int m = blockIdx.x, side = blockIdx.y;
  const loftr_fmap f = side ? g1 : g0;
  const int wc = side ? w1c : w0c;
  const long cell = side ? j_ids[m] : i_ids[m];
  const long b = b_ids[m];
  const int cy = (int)(cell / wc) * stride, cx = (int)(cell % wc) * stride, r = W / 2, WW = W * W;
  const float* src = dxcat + ((long)side * M + m) * WW * 2 * Cf;
  for (int i = threadIdx.x; i < WW * Cf; i += blockDim.x) {
    const int ww = i / Cf, c = i - ww * Cf;
    const int y = cy + ww / W - r, x = cx + ww % W - r;
    if (y >= 0 && y < f.H && x >= 0 && x < f.W)
      atomicAdd(const_cast<float*>(f.data) + b * f.sn + (long)c * f.sc + (long)y * f.sh + (long)x * f.sw, src[(long)ww * 2 * Cf + c]);
  }
}
__global__ void scatter_cc_kernel(const float* __restrict__ dcc, const int64_t* __restrict__ b_ids, const int64_t* __restrict__ i_ids,
                                  const int64_t* __restrict__ j_ids, int M, int L, int S, int Cc, float* __restrict__ g0, float* __restrict__ g1) {
  const long m = blockIdx.x;
  const int side = blockIdx.y;
  const long row = side ? b_ids[m] * S + j_ids[m] : b_ids[m] * L + i_ids[m];
  float* dst = (side ? g1 : g0) + row * Cc;
  for (int c = threadIdx.x; c < Cc; c += blockDim.x) atomicAdd(dst + c, dcc[((long)side * M + m) * Cc + c]);
}
__global__ void transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int R, int Cc) {      // wt[c][r] = w[r][c]
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)R * Cc) return;
  const int r = (int)(i / Cc), c = (int)(i - (long)r * Cc);
  wt[(long)c * R + r] = w[i];
}
inline dim3 g1d(long n) { return dim3((unsigned)((n + 255) / 256)); }

struct Ws { float *cc, *cwin, *xcat, *dy, *dxcat, *dcwin, *dcc, *wmT, *wdT, *wpart, *cpart; void* lin; size_t lin_bytes, wpart_floats; };
Ws carve(WsAlloc& wa, int M, int WW, int Cf, int Cc) {
  Ws w{};
  const size_t R = (size_t)2 * M, T = R * WW;
  w.cc = wa.take<float>(R * Cc); w.cwin = wa.take<float>(R * Cf);
  w.xcat = wa.take<float>(T * 2 * Cf); w.dy = wa.take<float>(T * Cf); w.dxcat = wa.take<float>(T * 2 * Cf);
  w.dcwin = wa.take<float>(R * Cf); w.dcc = wa.take<float>(R * Cc);
  w.wmT = wa.take<float>((size_t)2 * Cf * Cf); w.wdT = wa.take<float>((size_t)Cc * Cf);
  w.wpart_floats = wgrad_part_floats((long)T, Cf, 2 * Cf) + wgrad_part_floats((long)R, Cf, Cc);
  w.wpart = wa.take<float>(w.wpart_floats);
  w.cpart = wa.take<float>(colsum_part_floats((long)T, Cf) + 64);
  w.lin_bytes = loftr_linear_workspace_bytes((int)T, 2 * Cf, Cc > 2 * Cf ? Cc : 2 * Cf);
  w.lin = wa.take<char>(w.lin_bytes);
  return w;
}
}  // namespace fb
}  // namespace

extern "C" size_t loftr_fine_preprocess_bwd_workspace_bytes(int M, int W, int Cf, int Cc) {
  if (M <= 0 || W <= 0 || Cf <= 0 || Cc <= 0) return 0;
  WsAlloc wa(nullptr, ~(size_t)0);
  (void)fb::carve(wa, M, W * W, Cf, Cc);
  return wa.off + 256;
}

// grad_f0 / grad_f1: maps in the layout of feat_f0 / feat_f1 (any strides), grad_c0 [N,L,Cc], grad_c1 [N,S,Cc]: ALL FOUR ARE ADDED TO
// (the caller zero-fills them; windows and matches overlap).  grad_down_w [Cf,Cc], grad_down_b [Cf], grad_merge_w [Cf,2Cf],
// grad_merge_b [Cf]: written.
extern "C" int loftr_fine_preprocess_bwd(const loftr_fmap* feat_f0, const loftr_fmap* feat_f1, const float* feat_c0, const float* feat_c1,
                                         int L, int S, int Cc, const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids, int M,
                                         int w0c, int w1c, int stride, int W, int Cf, const float* down_w, const float* down_b,
                                         const float* merge_w, const float* grad_out0, const float* grad_out1, const loftr_fmap* grad_f0,
                                         const loftr_fmap* grad_f1, float* grad_c0, float* grad_c1, float* grad_down_w, float* grad_down_b,
                                         float* grad_merge_w, float* grad_merge_b, void* ws, size_t ws_bytes, void* stream) {
  using namespace fb;
  LOFTR_CHECK_ARG(M >= 0);
  if (M == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(feat_f0 && feat_f1 && feat_f0->data && feat_f1->data && feat_c0 && feat_c1 && b_ids && i_ids && j_ids && down_w && down_b &&
                  merge_w && grad_out0 && grad_out1 && grad_f0 && grad_f1 && grad_f0->data && grad_f1->data && grad_c0 && grad_c1 &&
                  grad_down_w && grad_down_b && grad_merge_w && grad_merge_b && ws);
  LOFTR_CHECK_ARG(w0c > 0 && w1c > 0 && stride > 0 && W > 0 && (W & 1));
  if (Cf % 32 != 0 || Cc % 32 != 0 || 2 * Cf > 256 || Cc > 256) return LOFTR_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  WsAlloc wa(ws, ws_bytes);
  Ws a = carve(wa, M, W * W, Cf, Cc);
  if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
  const int WW = W * W, C2 = 2 * Cf;
  const long R = 2L * M, T = R * WW;
  int rc;
#define LIN(A_, W_, OUT_, M_, N_, K_) if ((rc = loftr_linear_fwd(A_, W_, OUT_, (int)(M_), N_, K_, a.lin, a.lin_bytes, stream))) return rc
  // ---- recompute: gathered coarse rows, their down projection, Xcat
  hipLaunchKernelGGL(gather_cc_kernel, dim3(M, 2), dim3(256), 0, st, feat_c0, feat_c1, b_ids, i_ids, j_ids, M, L, S, Cc, a.cc);
  LIN(a.cc, down_w, a.cwin, R, Cf, Cc);
  hipLaunchKernelGGL(add_bias_kernel, g1d(R * Cf), dim3(256), 0, st, a.cwin, down_b, R, Cf);
  hipLaunchKernelGGL(build_xcat_kernel, dim3(M, 2), dim3(256), 0, st, *feat_f0, *feat_f1, b_ids, i_ids, j_ids, M, w0c, w1c, stride, W, Cf,
                     a.cwin, a.xcat);
  hipLaunchKernelGGL(stack2_kernel, g1d(T * Cf), dim3(256), 0, st, grad_out0, grad_out1, a.dy, (long)M * WW * Cf);
  // ---- merge_feat
  hipLaunchKernelGGL(transpose_kernel, g1d((long)Cf * C2), dim3(256), 0, st, merge_w, a.wmT, Cf, C2);      // [Cf, 2Cf] -> [2Cf, Cf]
  LIN(a.dy, a.wmT, a.dxcat, T, C2, Cf);
  if ((rc = launch_wgrad(a.dy, Cf, a.xcat, C2, T, grad_merge_w, a.wpart, a.wpart_floats, st))) return rc;
  if ((rc = launch_colsum(a.dy, T, Cf, grad_merge_b, a.cpart, st))) return rc;
  // ---- windows back into the fine maps; coarse context
  hipLaunchKernelGGL(scatter_windows_kernel, dim3(M, 2), dim3(256), 0, st, *grad_f0, *grad_f1, b_ids, i_ids, j_ids, M, w0c, w1c, stride, W, Cf,
                     a.dxcat);
  hipLaunchKernelGGL(sum_ctx_kernel, g1d(R * Cf), dim3(256), 0, st, a.dxcat, R, WW, Cf, a.dcwin);
  // ---- down_proj
  hipLaunchKernelGGL(transpose_kernel, g1d((long)Cf * Cc), dim3(256), 0, st, down_w, a.wdT, Cf, Cc);       // [Cf, Cc] -> [Cc, Cf]
  LIN(a.dcwin, a.wdT, a.dcc, R, Cc, Cf);
  if ((rc = launch_wgrad(a.dcwin, Cf, a.cc, Cc, R, grad_down_w, a.wpart, a.wpart_floats, st))) return rc;
  if ((rc = launch_colsum(a.dcwin, R, Cf, grad_down_b, a.cpart, st))) return rc;
  hipLaunchKernelGGL(scatter_cc_kernel, dim3(M, 2), dim3(256), 0, st, a.dcc, b_ids, i_ids, j_ids, M, L, S, Cc, grad_c0, grad_c1);
#undef LIN
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
