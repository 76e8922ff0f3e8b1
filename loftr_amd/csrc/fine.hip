// Fine level: window gather + coarse-context merge (FinePreprocess) and the soft-argmax
// refinement (FineMatching).
//   reference: src/loftr/loftr_module/fine_preprocess.py:29-59, src/loftr/utils/fine_matching.py:15-74
#include "linear.h"

namespace {

// One block per (match, side): copies the W x W window of the fine map centred on the matched
// coarse cell into a dense [W*W, Cf] tile (zeros outside the map) -- the only rows of
// F.unfold(kernel=W, stride, padding=W//2) the reference ever uses (fine_preprocess.py:40-47).
// Thread <-> channel, so a channels-last map is read coalesced; the tile is written in the SP
// GEMM-operand format (gemm.h) because its only consumer is the merge_feat GEMM.
//   grid (M, 2), Cf threads (Cf even, a multiple of 32).
__global__ void gather_windows_kernel(loftr_fmap f0, loftr_fmap f1, const int64_t* __restrict__ b_ids,
                                      const int64_t* __restrict__ i_ids, const int64_t* __restrict__ j_ids,
                                      int M, int w0c, int w1c, int stride, int W, int Cf,
                                      sp_t* __restrict__ win0, sp_t* __restrict__ win1) {
  const int m = blockIdx.x, side = blockIdx.y;
  const loftr_fmap f = side ? f1 : f0;
  const int wc = side ? w1c : w0c;
  const long cell = side ? j_ids[m] : i_ids[m];
  const long b = b_ids[m];
  const int cy = (int)(cell / wc) * stride, cx = (int)(cell % wc) * stride;
  const int r = W / 2;
  sp_t* out = (side ? win1 : win0) + (long)m * W * W * Cf;
  for (int c = threadIdx.x; c < Cf; c += blockDim.x) {
    const float* base = f.data + b * f.sn + (long)c * f.sc;
    for (int wy = 0; wy < W; ++wy) {
      const int y = cy + wy - r;
      for (int wx = 0; wx < W; ++wx) {
        const int x = cx + wx - r;
        float v = 0.f;
        if (y >= 0 && y < f.H && x >= 0 && x < f.W) v = base[(long)y * f.sh + (long)x * f.sw];
        sp_store(out + (wy * W + wx) * Cf, c, v, true);
      }
    }
  }
}

// Matched coarse features feat_c0[b, i, :] / feat_c1[b, j, :] (fine_preprocess.py:51-52) gathered into
// dense [M, Cc] SP tiles (A operand of down_proj).   grid (M, 2), Cc threads.
__global__ void gather_coarse_kernel(const float* __restrict__ fc0, const float* __restrict__ fc1,
                                     const int64_t* __restrict__ b_ids, const int64_t* __restrict__ i_ids,
                                     const int64_t* __restrict__ j_ids, int L, int S, int Cc,
                                     sp_t* __restrict__ cg0, sp_t* __restrict__ cg1) {
  const long m = blockIdx.x;
  const int side = blockIdx.y;
  const long row = side ? b_ids[m] * S + j_ids[m] : b_ids[m] * L + i_ids[m];
  const float* src = (side ? fc1 : fc0) + row * Cc;
  sp_t* dst = (side ? cg1 : cg0) + m * Cc;
  for (int c = threadIdx.x; c < Cc; c += blockDim.x) sp_store(dst, c, src[c], true);
}

// FineMatching: one wave per match.  Lane r < WW owns window position r.
//   grid (ceil(M/4)), 256 threads.
__global__ __launch_bounds__(256) void fine_match_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                         int M, int WW, int W, int C,
                                                         const float* __restrict__ mkpts1_c,
                                                         const int64_t* __restrict__ b_ids, float scale,
                                                         const float* __restrict__ scale1,
                                                         float* __restrict__ expec_f, float* __restrict__ mkpts1_f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long m = (long)blockIdx.x * 4 + wave;
  if (m >= M) return;
  const float* p = f0 + (m * WW + WW / 2) * C;            // centre feature of window 0     :43
  const float* q = f1 + m * WW * C;
  float sim = 0.f;
  for (int r = 0; r < WW; ++r) {
    float part = 0.f;
    for (int c = lane; c < C; c += 64) part += p[c] * q[(long)r * C + c];
    part = wave_sum(part);                                 // einsum('mc,mrc->mr')           :44
    if (lane == r) sim = part;
  }
  const bool act = lane < WW;
  const float t = act ? sim * (1.f / sqrtf((float)C)) : -3.0e38f;       // softmax_temp * sim  :45-46
  const float mx = wave_max(t);
  const float e = act ? expf(t - mx) : 0.f;
  const float heat = e / wave_sum(e);
  // normalised grid of kornia.create_meshgrid(W, W, True): x fastest, values -1 .. 1
  const float gx = act ? 2.f * (float)(lane % W) / (float)(W - 1) - 1.f : 0.f;
  const float gy = act ? 2.f * (float)(lane / W) / (float)(W - 1) - 1.f : 0.f;
  const float cx = wave_sum(heat * gx), cy = wave_sum(heat * gy);       // spatial_expectation2d :49
  const float vx = wave_sum(heat * gx * gx) - cx * cx;                   // :53
  const float vy = wave_sum(heat * gy * gy) - cy * cy;
  const float sd = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f)); // :54
  if (lane == 0) {
    expec_f[m * 3 + 0] = cx; expec_f[m * 3 + 1] = cy; expec_f[m * 3 + 2] = sd;
    float sx = scale, sy = scale;
    if (scale1) { const long b = b_ids[m]; sx = scale * scale1[b * 2]; sy = scale * scale1[b * 2 + 1]; }
    const float half = (float)(W / 2);
    mkpts1_f[m * 2 + 0] = mkpts1_c[m * 2 + 0] + (cx * half) * sx;       // :69
    mkpts1_f[m * 2 + 1] = mkpts1_c[m * 2 + 1] + (cy * half) * sy;
  }
}

}  // namespace

extern "C" size_t loftr_fine_preprocess_workspace_bytes(int M, int W, int Cf) {
  if (M <= 0) return 0;
  const size_t Cc = 2 * (size_t)Cf;                               // coarse width (d_model_c = 256 for Cf = 128)
  size_t b = 0;
  b += 2 * align_up((size_t)M * W * W * Cf * 4, 256);             // windows (SP)
  b += 2 * align_up((size_t)M * Cc * 4 * 2, 256);                 // gathered coarse features (SP), generous
  b += 4 * align_up((size_t)M * Cf * 4, 256);                     // down-projected ctx (SP), merged ctx (fp32), x2 sides
  b += align_up((size_t)Cf * Cc * 4 * 2, 256) + 2 * align_up((size_t)Cf * Cf * 4, 256);   // weights (SP)
  return b + 8192;
}

extern "C" int loftr_fine_preprocess(const loftr_fmap* feat_f0, const loftr_fmap* feat_f1,
                                     const float* feat_c0, const float* feat_c1, int L, int S, int Cc,
                                     const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids,
                                     int M, int w0c, int w1c, int stride, int W, int Cf,
                                     const float* down_w, const float* down_b, const float* merge_w,
                                     const float* merge_b, float* out0, float* out1, void* ws,
                                     size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(M >= 0);
  if (M == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(feat_f0 && feat_f1 && feat_f0->data && feat_f1->data && b_ids && i_ids && j_ids && out0 && out1);
  LOFTR_CHECK_ARG(w0c > 0 && w1c > 0 && stride > 0 && W > 0 && (W & 1) && Cf > 0 && Cf <= 1024);
  hipStream_t st = (hipStream_t)stream;
  const int WW = W * W;
  if (!down_w) {                                   // fine_concat_coarse_feat = False: windows only (fp32 out)
    return LOFTR_ERR_UNSUPPORTED;                  // no shipped config uses it (cvpr_ds_config.py:13, default.py:13)
  }
  LOFTR_CHECK_ARG(feat_c0 && feat_c1 && down_b && merge_w && merge_b && ws);
  if (Cf % 32 != 0 || Cc % 32 != 0 || Cc > 2 * Cf * 2) return LOFTR_ERR_UNSUPPORTED;
  WsAlloc wa(ws, ws_bytes);
  sp_t* win0 = wa.take<sp_t>((size_t)M * WW * Cf);
  sp_t* win1 = wa.take<sp_t>((size_t)M * WW * Cf);
  sp_t* cg0 = wa.take<sp_t>((size_t)M * Cc);
  sp_t* cg1 = wa.take<sp_t>((size_t)M * Cc);
  sp_t* cdn0 = wa.take<sp_t>((size_t)M * Cf);
  sp_t* cdn1 = wa.take<sp_t>((size_t)M * Cf);
  float* ctx0 = wa.take<float>((size_t)M * Cf);
  float* ctx1 = wa.take<float>((size_t)M * Cf);
  sp_t* down_sp = wa.take<sp_t>((size_t)Cf * Cc);
  sp_t* mwin_sp = wa.take<sp_t>((size_t)Cf * Cf);   // merge_feat.weight[:, :Cf]  (window half)
  sp_t* mctx_sp = wa.take<sp_t>((size_t)Cf * Cf);   // merge_feat.weight[:, Cf:]  (coarse-context half)
  if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
  int rc;
  {
    SpJobs j; j.n = 3;
    j.src[0] = down_w; j.dst[0] = down_sp; j.rows[0] = Cf; j.K[0] = Cc; j.ld[0] = Cc;
    j.src[1] = merge_w; j.dst[1] = mwin_sp; j.rows[1] = Cf; j.K[1] = Cf; j.ld[1] = 2 * Cf;
    j.src[2] = merge_w + Cf; j.dst[2] = mctx_sp; j.rows[2] = Cf; j.K[2] = Cf; j.ld[2] = 2 * Cf;
    if ((rc = launch_sp_convert(j, st))) return rc;
  }
  { TimedLaunch tl(LOFTR_T_GATHER, st);
    hipLaunchKernelGGL(gather_windows_kernel, dim3(M, 2), dim3(Cf < 64 ? 64 : Cf), 0, st, *feat_f0, *feat_f1, b_ids,
                       i_ids, j_ids, M, w0c, w1c, stride, W, Cf, win0, win1); }
  hipLaunchKernelGGL(gather_coarse_kernel, dim3(M, 2), dim3(Cc < 64 ? 64 : (Cc > 1024 ? 1024 : Cc)), 0, st, feat_c0,
                     feat_c1, b_ids, i_ids, j_ids, L, S, Cc, cg0, cg1);
  LOFTR_CHECK_LAUNCH();
  for (int side = 0; side < 2; ++side) {
    sp_t* cg = side ? cg1 : cg0;
    sp_t* cdn = side ? cdn1 : cdn0;
    float* ctx = side ? ctx1 : ctx0;
    sp_t* win = side ? win1 : win0;
    float* out = side ? out1 : out0;
    // feat_c_win = down_proj(feat_c[b_ids, ids])                       fine_preprocess.py:51-52
    LinearArgs d{asrc_plain(cg, Cc), down_sp, Cc, nullptr, cdn, Cf, M, Cf, Cc, down_b, 1, 1, false};
    if ((rc = launch_linear(d, st))) return rc;
    // merge_feat(cat[window, repeat(feat_c_win)]) = window @ Wm[:, :Cf]^T + (feat_c_win @ Wm[:, Cf:]^T + b):
    // the coarse half is constant over the window -> computed once per match      :53-56
    LinearArgs c{asrc_plain(cdn, Cf), mctx_sp, Cf, ctx, nullptr, Cf, M, Cf, Cf, merge_b, 1, 1, false};
    if ((rc = launch_linear(c, st))) return rc;
    LinearArgs w{asrc_plain(win, Cf), mwin_sp, Cf, out, nullptr, Cf, M * WW, Cf, Cf, ctx, 2, WW, false};
    if ((rc = launch_linear(w, st))) return rc;
  }
  return LOFTR_OK;
}

extern "C" int loftr_fine_match(const float* feat_f0, const float* feat_f1, int M, int WW, int C,
                                const float* mkpts1_c, const int64_t* b_ids, float scale,
                                const float* scale1, float* expec_f, float* mkpts1_f, void* stream) {
  LOFTR_CHECK_ARG(M >= 0);
  if (M == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(feat_f0 && feat_f1 && mkpts1_c && expec_f && mkpts1_f && C > 0 && WW > 0);
  LOFTR_CHECK_ARG(scale1 == nullptr || b_ids != nullptr);
  int W = 1;
  while (W * W < WW) ++W;
  if (W * W != WW || WW > 64 || W < 2) return LOFTR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fine_match_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, feat_f0, feat_f1, M,
                     WW, W, C, mkpts1_c, b_ids, scale, scale1, expec_f, mkpts1_f);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
