/* libloftr_hip -- C-ABI of the MI355X-native LoFTR matching path (gfx950 / CDNA4).
 *
 * Drop-in boundary.  The reference (zju3dv/LoFTR) is pure Python: it has no FFI / plugin
 * registry; the seam is nn.Module composition in src/loftr/loftr.py:20-27,56-75.  Each entry
 * point below replaces the ATen op sequence of one of those sub-modules and is what a ctypes
 * binding inside the reference's modules would call (INTEGRATION.md shows the stub).  The
 * Python mirror of the reference interface lives in loftr_amd/ and calls exactly these symbols.
 *
 * Conventions
 *   - every function is asynchronous on the caller-supplied hipStream_t (passed as void*);
 *   - all device buffers are caller-allocated (PyTorch owns memory); the library never
 *     allocates or frees device memory; `*_workspace_bytes` tells how much scratch to pass;
 *   - all floating point data is fp32 (GEMMs evaluate fp32 products as 3 fp16 MFMAs with fp32
 *     accumulation, csrc/gemm.h), ids are int64, masks are uint8 (0 = padded), as in the
 *     reference;
 *   - return value: 0 = ok, negative = loftr_status below (no exceptions);
 *   - no environment variable is read; the only process-global state are the debug switches, the timing mask and the
 *     range guard at the end of this header, all off / at their defaults unless set through their entry points.
 */
#ifndef LOFTR_HIP_H_
#define LOFTR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  LOFTR_OK = 0,
  LOFTR_ERR_BAD_ARG = -1,       /* null pointer / non-positive or unsupported shape            */
  LOFTR_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels are built for (C, H, D, W)   */
  LOFTR_ERR_WORKSPACE = -3,     /* workspace smaller than *_workspace_bytes()                  */
  LOFTR_ERR_LAUNCH = -4,        /* HIP reported a launch error                                 */
  LOFTR_ERR_NO_DEVICE = -5,     /* no gfx950 device visible                                    */
  LOFTR_ERR_COMM = -6,          /* RCCL unavailable or a collective / communicator call failed */
  LOFTR_ERR_RANGE = -7          /* range guard (loftr_hip_range_check_enable): |activation| >= 65504 or not finite */
} loftr_status;

/* 13: prepared transformer weights, RCCL entry points, scaled activations, pose estimation;
 * 14: training-side consumers (loftr_spvs_coarse / _fine, loftr_coarse_loss_sums, loftr_fine_loss_sums);
 * 16: backward of the matching heads and their losses (loftr_*_grad, loftr_dual_softmax_bwd, loftr_sinkhorn_bwd,
 *     loftr_fine_match_bwd);
 * 17: loftr_head_feat_grads (the feature-gradient GEMMs of both coarse heads);
 * 18: loftr_encoder_layer_bwd;
 * 19: loftr_fine_preprocess_bwd;
 * 20: loftr_conv_wgrad (backbone training: weight gradient of a convolution);
 * 21: training-mode glue of the backbone (loftr_bn_train_fwd / _bwd, loftr_act_fwd / _bwd, loftr_upsample2x_bilinear_fwd / _bwd);
 * 22: the persistent coarse transformer (loftr_coarse_plan_bytes / _build / _signature, loftr_transformer_fwd_planned) and the
 *     debug switches (loftr_hip_debug_set / _get) that replace the library's environment variables;
 * 23: loftr_conv_scratch_bytes / loftr_conv_bn_act_prepared_scratch (the 196-channel layers' remainder channels as a tap-decomposed product);
 * 24: loftr_transformer_fwd_padded (padding masks: 128-token tiles without a valid token are not computed) */
#define LOFTR_HIP_ABI_VERSION 24

int loftr_hip_abi_version(void);
const char* loftr_hip_status_string(int status);
/* 0 if the current HIP device is a gfx950 part, LOFTR_ERR_NO_DEVICE otherwise. */
int loftr_hip_device_check(void);

/* A 4-D feature map [N,C,H,W] addressed through element strides (sn,sc,sh,sw), so that both NCHW
 * and channels-last (what the MIOpen backbone produces fastest) storage work without a copy. */
typedef struct {
  const float* data;
  long sn, sc, sh, sw;    /* element strides */
  int H, W;               /* spatial size    */
} loftr_fmap;

/* ---- position encoding + flatten ---------------------------------------------------------
 * Replaces: PositionEncodingSine.forward (src/loftr/utils/position_encoding.py:37-42) followed
 * by rearrange 'n c h w -> n (h w) c' (src/loftr/loftr.py:58-59).
 *   feat: [N,C,H,W] map (any strides), pe [C,pe_h,pe_w] (the module's constant table, H<=pe_h,
 *   W<=pe_w), out [N,H*W,C] contiguous. */
int loftr_pos_encode_flatten(const loftr_fmap* feat, const float* pe, int pe_h, int pe_w,
                             float* out, int N, int C, void* stream);

/* ---- LoFTREncoderLayer / LocalFeatureTransformer ------------------------------------------
 * Weights of one LoFTREncoderLayer (src/loftr/loftr_module/transformer.py:7-33); every matrix
 * is the nn.Linear weight as stored in the state_dict: [out_features, in_features] row-major. */
typedef struct {
  const float* q_proj;   /* [C, C]   */
  const float* k_proj;   /* [C, C]   */
  const float* v_proj;   /* [C, C]   */
  const float* merge;    /* [C, C]   */
  const float* mlp0;     /* [2C, 2C] */
  const float* mlp2;     /* [C, 2C]  */
  const float* norm1_w;  /* [C] */
  const float* norm1_b;  /* [C] */
  const float* norm2_w;  /* [C] */
  const float* norm2_b;  /* [C] */
} loftr_layer_weights;

/* Scratch needed by loftr_encoder_layer_fwd / loftr_transformer_fwd for `nb` sequences of
 * length L attending to sequences of length S (transformer: pass nb = 2*N). */
size_t loftr_encoder_workspace_bytes(int nb, int L, int S, int C);

/* Replaces: LoFTREncoderLayer.forward (transformer.py:35-58) incl. LinearAttention.forward
 * (src/loftr/loftr_module/linear_attention.py:20-47).
 *   x [nb,L,C], source [nb,S,C], x_mask [nb,L] / source_mask [nb,S] uint8 or NULL,
 *   out [nb,L,C] (may alias x).  C in {128, 256}, H = 8. */
int loftr_encoder_layer_fwd(const float* x, const float* source, const uint8_t* x_mask,
                            const uint8_t* source_mask, const loftr_layer_weights* w, float* out,
                            int nb, int L, int S, int C, int H, void* ws, size_t ws_bytes,
                            void* stream);

/* Backward of loftr_encoder_layer_fwd (what torch.autograd derives from transformer.py:35-58 + linear_attention.py:20-47):
 * from grad_out = dL/d out [nb,L,C]:  grad_x [nb,L,C], grad_source [nb,S,C] (WRITTEN, not accumulated: a self layer's caller adds
 * the two) and the gradients of the ten weight tensors (written into the caller's buffers, shapes as loftr_layer_weights).
 * The layer is recomputed from (x, source) in the workspace; C in {128, 256}, head dimension 16 or 32. */
typedef struct {
  float* q_proj; float* k_proj; float* v_proj; float* merge; float* mlp0; float* mlp2;
  float* norm1_w; float* norm1_b; float* norm2_w; float* norm2_b;
} loftr_layer_grads;
size_t loftr_encoder_layer_bwd_workspace_bytes(int nb, int L, int S, int C, int H);
int loftr_encoder_layer_bwd(const float* x, const float* source, const uint8_t* x_mask, const uint8_t* source_mask,
                            const loftr_layer_weights* w, const float* grad_out, float* grad_x, float* grad_source,
                            const loftr_layer_grads* gw, int nb, int L, int S, int C, int H, void* ws, size_t ws_bytes,
                            void* stream);

/* Replaces: LocalFeatureTransformer.forward (transformer.py:80-101), in place on feat0/feat1.
 *   feat0 [N,L,C], feat1 [N,S,C]; layer_is_cross[i] = 0 for 'self', 1 for 'cross';
 *   cross layers keep the reference's sequential dependency (feat1 attends to the UPDATED
 *   feat0, :96-97).  When L == S and feat1 == feat0 + N*L*C the two self-attention calls of a
 *   layer run as one batch of 2N sequences. */
int loftr_transformer_fwd(float* feat0, float* feat1, const uint8_t* mask0, const uint8_t* mask1,
                          const loftr_layer_weights* layers, const int* layer_is_cross,
                          int n_layers, int N, int L, int S, int C, int H, const void* prepared,
                          size_t prepared_bytes, void* ws, size_t ws_bytes, void* stream);
/* loftr_transformer_fwd for a caller that does not read the features of PADDING tokens (mask byte 0; MegaDepth batches padded to a common
 * size, dataset.py:107-125).  The reference computes x + LayerNorm2(mlp([x, b1])) for them in every layer, and nothing in LoFTR.forward reads
 * it: their scores are filled (coarse_matching.py:115-118), as attention sources they are multiplied by zero (linear_attention.py:37-40) and
 * the fine stage only gathers at matched -- valid -- tokens.  With skip_padded_tiles != 0 and masks given, a 128-token tile (tokens
 * 128 t .. 128 t + 127 of a sequence) whose mask bytes are all zero keeps its INPUT values in feat0 / feat1; every other token comes out
 * bit-identical to loftr_transformer_fwd (a fully masked tile's K^T V / Ksum partial is exactly +0 and is written as such in either call).
 * skip_padded_tiles == 0 or no masks: the same as loftr_transformer_fwd. */
int loftr_transformer_fwd_padded(float* feat0, float* feat1, const uint8_t* mask0, const uint8_t* mask1,
                                 const loftr_layer_weights* layers, const int* layer_is_cross,
                                 int n_layers, int N, int L, int S, int C, int H, const void* prepared,
                                 size_t prepared_bytes, void* ws, size_t ws_bytes, int skip_padded_tiles, void* stream);
/* The same forward as ONE persistent launch (csrc/encoder_fused.hip: coarse_persistent_kernel).  The reference's schedule
 * (transformer.py:91-99) synchronises whole calls; its data dependency -- feat1 attends to the UPDATED feat0 -- is per pair, and
 * below that per 128-token tile.  A PLAN is the dependency graph of one forward's work items for a shape (n_layers of the pattern
 * [self, cross] * P, N pairs, L and S tokens), ordered by a list schedule on the host; 256 resident workgroups pull it in order and
 * wait on counters in the workspace.  Build it once per shape into a caller-owned DEVICE buffer of loftr_coarse_plan_bytes() bytes
 * (0 = shape not supported; loftr_coarse_plan_build is a set-up call: it synchronises the stream), then pass it here.
 *   order 0 = dependency-driven (critical path first), 1 = the reference's call order (same arithmetic item by item: results are
 *   bit-identical, tests/test_hip_parity.py); the same value must be given to _build and to _fwd_planned.
 *   diag: NULL, or a device buffer of diag_bytes >= 16: word 0 = error (0 ok; 2 = the plan was built for another shape / order and
 *   nothing was computed; odd = a workgroup gave up waiting for a dependency after ~1 s, value = 1 + 2 * item), words 1-3 reserved;
 *   with diag_bytes >= 16 + 32 * items (items = loftr_coarse_plan_bytes / 32 - 1) additionally per work item four 8-byte words
 *   {popped, dependencies met, done} in 10 ns ticks and the workgroup id (profiling).  The caller zeroes word 0.
 * C = 256, H = 8, n_layers <= 8 even; anything else: LOFTR_ERR_UNSUPPORTED (use loftr_transformer_fwd). */
size_t loftr_coarse_plan_bytes(const int* layer_is_cross, int n_layers, int N, int L, int S);
int loftr_coarse_plan_build(const int* layer_is_cross, int n_layers, int N, int L, int S, int order, void* plan,
                            size_t plan_bytes, void* stream);
unsigned loftr_coarse_plan_signature(int n_layers, int N, int L, int S, int order);
int loftr_transformer_fwd_planned(float* feat0, float* feat1, const uint8_t* mask0, const uint8_t* mask1,
                                  const loftr_layer_weights* layers, const int* layer_is_cross,
                                  int n_layers, int N, int L, int S, int C, int H, const void* prepared,
                                  size_t prepared_bytes, void* ws, size_t ws_bytes, const void* plan,
                                  size_t plan_bytes, int plan_order, void* diag, size_t diag_bytes, void* stream);

/* Inference with constant weights: every layer matrix is re-encoded once (row-scaled split-fp16 operand format,
 * csrc/gemm.h) into a caller-owned buffer of loftr_transformer_prepared_bytes(n_layers, C) bytes; hand it to
 * loftr_transformer_fwd as `prepared` (NULL there = convert on every call into the workspace).  The LayerNorm vectors
 * are always read from `layers`.  The buffer must be rebuilt when a weight changes. */
size_t loftr_transformer_prepared_bytes(int n_layers, int C);
int loftr_transformer_prepare(const loftr_layer_weights* layers, int n_layers, int C, void* prepared,
                              size_t prepared_bytes, void* stream);

/* ---- CoarseMatching ------------------------------------------------------------------------
 * Geometry + selection parameters shared by the two match types. */
typedef struct {
  int N, h0c, w0c, h1c, w1c;     /* L = h0c*w0c, S = h1c*w1c                                    */
  int C;                         /* descriptor width (256)                                      */
  float thr;                     /* config['thr']            (strict >)                         */
  int border_rm;                 /* config['border_rm']                                          */
  float scale;                   /* hw0_i[0] / hw0_c[0]      (coarse_matching.py:242)           */
  const uint8_t* mask0;          /* [N,L] or NULL (MegaDepth padding masks)                     */
  const uint8_t* mask1;          /* [N,S] or NULL                                               */
  const float* scale0;           /* [N,2] or NULL                                               */
  const float* scale1;           /* [N,2] or NULL                                               */
} loftr_coarse_params;

/* Match outputs; capacity must be N*L rows (at most one match per row of conf_matrix).
 * counts[0] = M (total), counts[1+b] = matches of pair b.  Rows are in ascending (b, i). */
typedef struct {
  int64_t* b_ids;     /* [N*L] */
  int64_t* i_ids;     /* [N*L] */
  int64_t* j_ids;     /* [N*L] */
  float* mconf;       /* [N*L] */
  float* mkpts0_c;    /* [N*L,2] */
  float* mkpts1_c;    /* [N*L,2] */
  int32_t* counts;    /* [1+N]   */
} loftr_match_out;

size_t loftr_coarse_match_workspace_bytes(int N, int L, int S, int C);

/* Replaces: CoarseMatching.forward, match_type='dual_softmax' + get_coarse_match, eval branch
 * (src/loftr/utils/coarse_matching.py:105-119,150-196,238-261).
 *   feat_c0 [N,L,C], feat_c1 [N,S,C]; conf_out [N,L,S] or NULL (data['conf_matrix'] elided). */
int loftr_coarse_match_dual_softmax(const float* feat_c0, const float* feat_c1,
                                    const loftr_coarse_params* p, float temperature,
                                    float* conf_out, const loftr_match_out* out, void* ws,
                                    size_t ws_bytes, void* stream);

/* Replaces: CoarseMatching.forward, match_type='sinkhorn' (coarse_matching.py:121-143, calling
 * SuperGlue's log_optimal_transport) + get_coarse_match.
 *   conf_out [N,L,S] REQUIRED (used as the score store); assign_out [N,L+1,S+1] or NULL
 *   (data['conf_matrix_with_bin'] when config['sparse_spvs']). */
int loftr_coarse_match_sinkhorn(const float* feat_c0, const float* feat_c1,
                                const loftr_coarse_params* p, float bin_score, int iters,
                                int prefilter, float* conf_out, float* assign_out,
                                const loftr_match_out* out, void* ws, size_t ws_bytes,
                                void* stream);

/* ---- FinePreprocess ------------------------------------------------------------------------
 * Replaces: FinePreprocess.forward (src/loftr/loftr_module/fine_preprocess.py:29-59): 5x5
 * windows (stride = hf/hc, zero padded) of the fine maps at the matched cells, fused with the
 * down-projected coarse features.  No unfold volume is materialised.
 *   feat_f0/1: fine maps (loftr_fmap, any strides; channels-last reads are fully coalesced);
 *   feat_c0 [N,L,Cc], feat_c1 [N,S,Cc] (transformer outputs);
 *   out0/out1 [M,W*W,Cf].  down_w [Cf,Cc], down_b [Cf], merge_w [Cf,2Cf], merge_b [Cf];
 *   (fine_concat_coarse_feat = False -- down_w NULL -- is not supported: no shipped config uses it.) */

size_t loftr_fine_preprocess_workspace_bytes(int M, int W, int Cf);

int loftr_fine_preprocess(const loftr_fmap* feat_f0, const loftr_fmap* feat_f1,
                          const float* feat_c0, const float* feat_c1, int L, int S, int Cc,
                          const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids, int M,
                          int w0c, int w1c, int stride, int W, int Cf,
                          const float* down_w, const float* down_b, const float* merge_w,
                          const float* merge_b, float* out0, float* out1, void* ws,
                          size_t ws_bytes, void* stream);

/* Backward of loftr_fine_preprocess (fine_preprocess.py:29-59 under autograd) from grad_out0 / grad_out1 [M, W*W, Cf].
 *   grad_f0 / grad_f1 (maps laid out like feat_f0 / feat_f1), grad_c0 [N,L,Cc], grad_c1 [N,S,Cc]: ADDED TO (zero-fill them first:
 *   windows overlap and a cell may carry several matches; float atomics).  The four parameter gradients are written. */
size_t loftr_fine_preprocess_bwd_workspace_bytes(int M, int W, int Cf, int Cc);
int loftr_fine_preprocess_bwd(const loftr_fmap* feat_f0, const loftr_fmap* feat_f1, const float* feat_c0, const float* feat_c1,
                              int L, int S, int Cc, const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids, int M,
                              int w0c, int w1c, int stride, int W, int Cf, const float* down_w, const float* down_b,
                              const float* merge_w, const float* grad_out0, const float* grad_out1, const loftr_fmap* grad_f0,
                              const loftr_fmap* grad_f1, float* grad_c0, float* grad_c1, float* grad_down_w, float* grad_down_b,
                              float* grad_merge_w, float* grad_merge_b, void* ws, size_t ws_bytes, void* stream);

/* ---- FineMatching ---------------------------------------------------------------------------
 * Replaces: FineMatching.forward + get_fine_match (src/loftr/utils/fine_matching.py:15-74).
 *   feat_f0/1 [M,WW,C]; mkpts1_c [M,2]; b_ids [M]; scale = hw0_i[0]/hw0_f[0];
 *   scale1 [N,2] or NULL (applied iff the batch has 'scale0', fine_matching.py:68);
 *   expec_f [M,3] (x, y, std), mkpts1_f [M,2].  (mkpts0_f is mkpts0_c, :66.) */
int loftr_fine_match(const float* feat_f0, const float* feat_f1, int M, int WW, int C,
                     const float* mkpts1_c, const int64_t* b_ids, float scale,
                     const float* scale1, float* expec_f, float* mkpts1_f, void* stream);

/* ---- ResNet-FPN building blocks (the caller side of the path) -----------------
 * Activations are NHWC tensors in the library's SP GEMM-operand format: uint32 [B, H, W, Cp] with
 * Cp = C rounded up to a multiple of 32; per 32-channel group 16 dwords of fp16 "hi" halves then 16 dwords
 * of fp16 "lo" halves (x ~= hi + lo, csrc/gemm.h); pad channels are zero.  loftr_sp_from_f32 /
 * loftr_sp_to_f32 convert from / to plain fp32 NHWC.
 *
 * loftr_conv_bn_act replaces nn.Conv2d (bias=False) [+ eval-mode nn.BatchNorm2d] [+ residual add]
 * [+ ReLU / LeakyReLU] of src/loftr/backbone/resnet_fpn.py:5-40,100-118 as ONE implicit-GEMM kernel:
 *   x_sp [B,H,W,ceil32(Cin)], weight [Cout,Cin,KH,KW] fp32 as in the state_dict, addressed through its four
 *   element strides (contiguous or channels-last storage), bn_* [Cout] or all NULL,
 *   act: 0 none, 1 ReLU, 2 LeakyReLU(0.01); residual_sp [B,Ho,Wo,ceil32(Cout)] or NULL (added before act);
 *   outputs y_sp (SP) and / or y_f32 (fp32 [B,Ho,Wo,Cout]); Ho = (H + 2 pad - KH) / stride + 1.
 * loftr_upsample2x_add replaces F.interpolate(scale_factor=2, bilinear, align_corners=True) + add (:111-116). */
size_t loftr_conv_workspace_bytes(int Cin, int Cout, int KH, int KW);
int loftr_conv_bn_act(const uint32_t* x_sp, int B, int H, int W, int Cin, const float* weight,
                      const long* weight_strides, int Cout, int KH, int KW, int stride, int pad, const float* bn_weight, const float* bn_bias,
                      const float* bn_mean, const float* bn_var, float bn_eps, int act,
                      const uint32_t* residual_sp, uint32_t* y_sp, float* y_f32, void* ws, size_t ws_bytes,
                      const float* x_inv_scale, void* stream);
/* Inference with constant weights: fold BN + encode the filter once, then run any number of convolutions on it.
 * loftr_conv_prepare fills `prepared` (loftr_conv_workspace_bytes(Cin, Cout, KH, KW) bytes, caller-owned, must
 * stay untouched while in use); loftr_conv_bn_act_prepared is loftr_conv_bn_act (low_sp == NULL) or
 * loftr_conv1x1_upsample_add (low_sp != NULL: 1x1 / stride 1 / no act / SP output only) without the per-call
 * weight preparation (two launches and a memset per convolution).  loftr_conv_bn_act == prepare into ws + this. */
int loftr_conv_prepare(const float* weight, const long* weight_strides, int Cin, int Cout, int KH, int KW,
                       const float* bn_weight, const float* bn_bias, const float* bn_mean, const float* bn_var,
                       float bn_eps, void* prepared, size_t prepared_bytes, void* stream);
/* OR-ed into `act` of loftr_conv_bn_act / loftr_conv_bn_act_prepared: this launch shares the GPU with work on another
 * stream (e.g. the FPN fine branch next to the coarse matching stage).  The 3x3 kernel then launches one workgroup per
 * tile instead of one persistent workgroup per CU: a persistent workgroup holds its CU (151 KB of LDS) for the whole
 * kernel, so the other stream could only start between kernels; with per-tile workgroups it interleaves every ~50 us
 * (measured: +1 % end to end with the two-stream overlap, -1.2 % on the kernel when it runs alone). */
#define LOFTR_CONV_SHARED_GPU 0x100
int loftr_conv_bn_act_prepared(const uint32_t* x_sp, int B, int H, int W, int Cin, const void* prepared,
                               size_t prepared_bytes, int Cout, int KH, int KW, int stride, int pad, int act,
                               const uint32_t* residual_sp, const uint32_t* low_sp, uint32_t* y_sp, float* y_f32,
                               const float* x_inv_scale, void* stream);
/* The same with a SCRATCH buffer (round 6, ABI 23).  3x3 / stride-1 layers whose output width is 193 .. 199 channels (LoFTR's 196) pad to 224
 * columns: the 7th 32-column tile spends a whole tile's matrix work on <= 7 real channels.  With loftr_conv_scratch_bytes(..) bytes of scratch
 * (0 = this shape has no such form) those channels are computed as a TAP-DECOMPOSED product instead -- conv = sum over the nine taps of shifted
 * 1x1 convolutions, so all (tap, channel) pairs are the columns of ONE K = Cin product on the unshifted pixels, evaluated at the centre-tap
 * steps of the main kernel into the scratch (fp32 [pixels][9 R]) and summed, shifted, by a second small kernel (csrc/conv3x3_duo.h).  Same
 * products, another summation order for those channels (fp32 noise).  scratch == NULL or too small, y_f32 requested, debug switch
 * "conv_rem" = 0: exactly loftr_conv_bn_act_prepared. */
size_t loftr_conv_scratch_bytes(int B, int H, int W, int Cout, int KH, int KW, int stride);
int loftr_conv_bn_act_prepared_scratch(const uint32_t* x_sp, int B, int H, int W, int Cin, const void* prepared,
                                       size_t prepared_bytes, int Cout, int KH, int KW, int stride, int pad, int act,
                                       const uint32_t* residual_sp, const uint32_t* low_sp, uint32_t* y_sp, float* y_f32,
                                       const float* x_inv_scale, void* scratch, size_t scratch_bytes, void* stream);
/* Stem: nn.Conv2d(1, C0, 7, stride 2, padding 3, bias=False) + eval BatchNorm2d + ReLU (resnet_fpn.py:52-54,101),
 * direct convolution; x [B,1,H,W] fp32 through its element strides (sb, sc, sh, sw), y_sp [B,Ho,Wo,ceil32(C0)]. */
int loftr_stem_conv_bn_relu(const float* x, const long* x_strides, int B, int H, int W, const float* weight,
                            const long* weight_strides, int C0, const float* bn_weight, const float* bn_bias,
                            const float* bn_mean, const float* bn_var, float bn_eps, uint32_t* y_sp, void* stream);
int loftr_upsample2x_add(const uint32_t* low_sp, const uint32_t* lateral_sp, uint32_t* out_sp, int B, int Hl,
                         int Wl, int C, void* stream);
/* One FPN top-down step (resnet_fpn.py:110-112 / :115-117) in one launch:
 *   y = conv1x1(x, weight) + F.interpolate(low, scale_factor=2, mode='bilinear', align_corners=True)
 * x_sp [B,H,W,ceil32(Cin)], weight [Cout,Cin,1,1] (bias-free lateral `layerK_outconv`), low_sp
 * [B,H/2,W/2,ceil32(Cout)], y_sp [B,H,W,ceil32(Cout)]; H and W even.  The lateral map never reaches HBM
 * and is added in fp32 (the two-call form rounds it to SP first).  Workspace: loftr_conv_workspace_bytes(Cin,Cout,1,1). */
int loftr_conv1x1_upsample_add(const uint32_t* x_sp, int B, int H, int W, int Cin, const float* weight,
                               const long* weight_strides, int Cout, const uint32_t* low_sp, uint32_t* y_sp,
                               void* ws, size_t ws_bytes, void* stream);
int loftr_sp_from_f32(const float* src, uint32_t* dst_sp, long rows, int C, void* stream);
/* Operand scaling (csrc/gemm.h).  The fp16 (hi, lo) pair keeps 22 bits of a value only above 2^-3; GEMM operands are
 * therefore stored times a power of two that lifts their maximum to [2^13, 2^14).  Filters: per output channel, inside
 * loftr_conv_prepare.  Activations produced by the library are BatchNorm / LayerNorm bounded and stored as they are; an
 * fp32 activation tensor of arbitrary magnitude enters through loftr_sp_from_f32_scaled, which writes the INVERSE of the
 * power of two it applied to the whole tensor to *inv_scale_out (device float) -- pass that pointer as x_inv_scale to
 * loftr_conv_bn_act / loftr_conv_bn_act_prepared (NULL = the tensor is unscaled). */
int loftr_sp_from_f32_scaled(const float* src, uint32_t* dst_sp, long rows, int C, float* inv_scale_out, void* stream);
int loftr_sp_to_f32(const uint32_t* src_sp, float* dst, long rows, int C, void* stream);

/* ---- evaluation caller (the reference's test_step, src/lightning/lightning_loftr.py:205-229) ----------
 * Replaces compute_symmetrical_epipolar_errors (src/utils/metrics.py:50-68 with :31-47): squared symmetric
 * epipolar distance of every match under the ground-truth relative pose of its pair, fp32.
 *   mkpts0_f / mkpts1_f [M,2] f32 pixels, m_bids [M] i64 pair index, T_0to1 [N,4,4], K0 / K1 [N,3,3] f32
 *   (row-major), epi_errs [M] f32 in match order (= the reference's per-pair concatenation, because the matcher
 *   emits matches grouped by ascending pair).  A match whose pair index is outside [0,N) gets NaN. */
int loftr_epipolar_errors(const float* mkpts0_f, const float* mkpts1_f, const long* m_bids, const float* T_0to1,
                          const float* K0, const float* K1, long M, int N, float* epi_errs, void* stream);

/* ---- training-side consumers of the path's outputs, FORWARD VALUES ONLY --------------------
 * loftr_spvs_coarse replaces spvs_coarse (src/loftr/utils/supervision.py:22-109, with warp_kpts of
 * src/loftr/utils/geometry.py:5-54): both coarse grids are warped into the other image through the depth maps and the
 * relative pose, rounded to the nearest coarse cell, and the mutual-nearest pairs become the ground-truth matches.
 *   depth0 [N,dh0,dw0], depth1 [N,dh1,dw1] f32; T_0to1, T_1to0 [N,4,4]; K0, K1 [N,3,3]; scale0/1 [N,2] or NULL (both);
 *   mask0 [N,L], mask1 [N,S] uint8 coarse masks or NULL (both); H*, W* = image sizes, scale = RESOLUTION[0] (8).
 *   Outputs: w_pt0_i [N,L,2], pt1_i [N,S,2] (data['spv_w_pt0_i'], ['spv_pt1_i']); spv_b/i/j [capacity N*L] in ascending
 *   (b, i), *count = their number (device int32; 0 -> the caller substitutes the reference's single (0,0,0), :94-99);
 *   conf_gt [N,L,S] or NULL (data['conf_matrix_gt']; the losses below work from the id lists and do not need it).
 * loftr_spvs_fine replaces spvs_fine (:124-142): expec_f_gt = (w_pt0_i[b,i] - pt1_i[b,j]) / scale / radius, scale
 *   multiplied by scale1[b] when given (the reference applies it iff 'scale0' is in the batch).
 * loftr_coarse_loss_sums / loftr_fine_loss_sums produce the reduction sums of LoFTRLoss (src/losses/loftr_loss.py:22-157)
 *   in fp64, one pass each (see csrc/train.hip for the layout of `sums`); the means, weights and corner cases are
 *   finished by the caller (loftr_amd/training.py).  Their gradients: loftr_coarse_loss_grad / loftr_fine_loss_grad below. */
typedef struct {
  int N, H0, W0, H1, W1, scale;
  int dh0, dw0, dh1, dw1;
  const float* depth0; const float* depth1;
  const float* T_0to1; const float* T_1to0;
  const float* K0; const float* K1;
  const float* scale0; const float* scale1;
  const uint8_t* mask0; const uint8_t* mask1;
} loftr_spvs_params;
size_t loftr_spvs_coarse_workspace_bytes(int N, int L, int S);
int loftr_spvs_coarse(const loftr_spvs_params* p, float* w_pt0_i, float* pt1_i, int64_t* spv_b, int64_t* spv_i, int64_t* spv_j,
                      int32_t* count, float* conf_gt, void* ws, size_t ws_bytes, void* stream);
int loftr_spvs_fine(const float* w_pt0_i, const float* pt1_i, int L, int S, const int64_t* b_ids, const int64_t* i_ids,
                    const int64_t* j_ids, long M, float scale, float radius, const float* scale1, float* expec_f_gt, void* stream);
size_t loftr_loss_workspace_bytes(int N, int L, int S);
int loftr_coarse_loss_sums(const float* conf, int N, int L, int S, int kind, const int64_t* gt_b, const int64_t* gt_i,
                           const int64_t* gt_j, long M, const uint8_t* mask0, const uint8_t* mask1, float alpha, float gamma,
                           double* sums, void* ws, size_t ws_bytes, void* stream);
int loftr_fine_loss_sums(const float* expec_f, int ld, const float* expec_f_gt, long M, int with_std, float correct_thr,
                         double* sums, void* ws, size_t ws_bytes, void* stream);

/* ---- backward of the matching heads and of the losses that read them -----------------------------------------------
 * What torch.autograd derives for the reference between `loss` (src/lightning/lightning_loftr.py:112-133,
 * src/losses/loftr_loss.py:165-192) and the INPUTS OF THE TWO HEADS: feat_c0 / feat_c1 entering CoarseMatching
 * (src/loftr/utils/coarse_matching.py:105-119, dual-softmax) and feat_f0 / feat_f1 entering FineMatching
 * (src/loftr/utils/fine_matching.py:43-57).  One kernel per node; each recomputes the forward quantities it needs from the
 * node's inputs.  Upstream of the heads the chain continues with loftr_encoder_layer_bwd, loftr_fine_preprocess_bwd and
 * loftr_conv_wgrad (+ the forward convolutions on the transposed filter for the input gradient) declared further down.
 *
 * loftr_coarse_loss_grad: grad_conf = d(pos_scale * sum_pos + neg_scale * sum_neg) / d conf for the sums of
 *   loftr_coarse_loss_sums with the same kind, ids and masks ([N,L,S]; kind 1: [N,L+1,S+1] = conf_matrix_with_bin, workspace
 *   loftr_loss_workspace_bytes); the gradient of torch.clamp(conf, 1e-6, 1 - 1e-6) (:45,:54) is included.  The caller folds
 *   means, loss weights, corner cases (:31-42) and the upstream gradient into pos_scale = up * c_pos_w / M and neg_scale =
 *   up * c_neg_w / (N L S - M) (kind 1: / sums[3], the number of supervised dustbin entries).
 * loftr_fine_loss_grad: grad_expec [M,ld] = upstream * d loss_f / d expec_f (:108-157); `sums` is the DEVICE array the forward
 *   (loftr_fine_loss_sums) filled; the std column gets 0 (weight.detach(), :131); training = the module's .training (:113-117).
 * loftr_dual_softmax_bwd: dsim [N,L,S] = dL/d sim_matrix from grad_conf = dL/d conf_matrix (:110-119; 0 on the mask-filled
 *   entries).  sim = <feat_c0, feat_c1> / (C * temperature), so dL/dfeat_c0 = dsim . feat_c1 / (C T) and dL/dfeat_c1 =
 *   dsim^T . feat_c0 / (C T): loftr_head_feat_grads below (split-fp16 MFMA, csrc/head_grads.hip).
 *   Workspace: loftr_coarse_match_workspace_bytes(N, L, S, C).
 * loftr_fine_match_bwd: grad_f0 / grad_f1 [M,WW,C] from grad_expec [M,3] = dL/d expec_f (x, y, std) (:43-57; grad_f0 is
 *   non-zero at the centre row only, :43).
 * loftr_sinkhorn_bwd: the Sinkhorn head (coarse_matching.py:121-143 + SuperGlue log_optimal_transport) in reverse mode through
 *   the `iters` unrolled iterations: dZ [N,L+1,S+1] = dL/d couplings (scores padded with bin_score) from grad_assign =
 *   dL/d conf_matrix_with_bin, *dbin = dL/d bin_score (device float); z_scratch [N,L,S] receives the re-created scores.
 *   sim = <feat_c0, feat_c1> / C: the caller slices dZ[:, :L, :S] (zero on mask-filled entries) and finishes with two GEMMs. */
int loftr_coarse_loss_grad(const float* conf, int N, int L, int S, int kind, const int64_t* gt_b, const int64_t* gt_i,
                           const int64_t* gt_j, long M, const uint8_t* mask0, const uint8_t* mask1, float alpha, float gamma,
                           double pos_scale, double neg_scale, float* grad_conf, void* ws, size_t ws_bytes, void* stream);
int loftr_fine_loss_grad(const float* expec_f, int ld, const float* expec_f_gt, long M, int with_std, float correct_thr,
                         int training, const double* sums, float upstream, float* grad_expec, void* stream);
int loftr_dual_softmax_bwd(const float* feat_c0, const float* feat_c1, const loftr_coarse_params* p, float temperature,
                           const float* grad_conf, float* dsim, void* ws, size_t ws_bytes, void* stream);
int loftr_fine_match_bwd(const float* feat_f0, const float* feat_f1, int M, int WW, int C, const float* grad_expec,
                         float* grad_f0, float* grad_f1, void* stream);
/* The einsum behind both coarse heads in reverse (coarse_matching.py:110-114 / :122-123: sim = <feat_c0, feat_c1> * alpha):
 *   g0 [N,L,C] = alpha * dsim . feat_c1,   g1 [N,S,C] = alpha * dsim^T . feat_c0       (either output may be null)
 * dsim: N matrices of L x S floats with row pitch dsim_ld and batch stride dsim_bs (the Sinkhorn head hands in the interior of its
 * [L+1, S+1] gradient); fp32 in and out, split-fp16 MFMA products with fp32 accumulation (csrc/head_grads.hip).  C % 32 == 0, C <= 256. */
int loftr_head_feat_grads(const float* dsim, long dsim_ld, long dsim_bs, const float* feat_c0, const float* feat_c1,
                          int N, int L, int S, int C, float alpha, float* g0, float* g1, void* stream);
/* Weight gradient of a bias-free convolution (what autograd computes for every nn.Conv2d of resnet_fpn.py in a training step):
 *   dw_taps[ky * KW + kx][co][ci] = sum_{b,y,x} dy[b,y,x,co] * x[b, y stride + ky - pad, x stride + kx - pad, ci]   (zero outside the map)
 * dy [B,Ho,Wo,Cout], x [B,H,W,Cin] fp32 channels-last; the caller permutes dw_taps [KH*KW, Cout, Cin] to [Cout,Cin,KH,KW].  All taps in
 * one launch, each a split-K product over the output pixels on the split-fp16 MFMA path with fp32 accumulation and an ordered sum of the partials
 * (deterministic).  Cin % 4 == 0, Cin <= 256 (the one-channel stem: hand in the 7 x 7 patches as a 52-channel 1 x 1 problem).
 * The input gradient is loftr_conv_bn_act on the flipped, transposed filter (stride 2: on the zero-interleaved dy). */
size_t loftr_conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW);
int loftr_conv_wgrad(const float* dy, const float* x, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                     float* dw_taps, void* ws, size_t ws_bytes, void* stream);
size_t loftr_sinkhorn_bwd_workspace_bytes(int N, int L, int S, int C, int iters);
int loftr_sinkhorn_bwd(const float* feat_c0, const float* feat_c1, const loftr_coarse_params* p, float bin_score, int iters,
                       const float* grad_assign, float* z_scratch, float* dZ, float* dbin, void* ws, size_t ws_bytes, void* stream);

/* Replaces estimate_pose (src/utils/metrics.py:72-98: cv2.findEssentialMat(RANSAC) + cv2.recoverPose on intrinsics-
 * normalised key points), the pose step of compute_pose_errors (:101-136).  HOST function (cv2 is a CPU library too):
 * all pointers are host memory, the call is synchronous.  kpts0 / kpts1 [M,2] pixels, K0 / K1 [3,3] row-major,
 * thresh_px = RANSAC_PIXEL_THR (0.5), conf = RANSAC_CONF (0.99999), seed for the sampler.
 * Outputs: R [3,3], t [3] (unit norm), inliers [M] (RANSAC inliers that are in front of both cameras), *n_inliers =
 * their number, or -1 when the reference would return None (M < 5, no model, no point passes the cheirality test).
 * PARITY UNPINNED against OpenCV (absent from this image): published algorithms restated (five-point solver of Nister
 * 2004, Sampson-distance RANSAC with OpenCV's documented parameters), own sampling sequence -- csrc/pose.hip.
 * loftr_five_point exposes the minimal / least-squares solver: q0 / q1 [n,2] normalised points (double), up to 10
 * essential matrices (row-major, unit Frobenius norm) in E_out [10,9]. */
int loftr_estimate_pose(const float* kpts0, const float* kpts1, long M, const float* K0, const float* K1, float thresh_px,
                        float conf, unsigned seed, float* R_out, float* t_out, uint8_t* inliers_out, long* n_inliers);
int loftr_five_point(const double* q0, const double* q1, int n, double* E_out, int* n_solutions);

/* ---- input wire format (the step before the path; src/utils/dataset.py:78-89,111-118,149, megadepth.py:116-121) ----
 * From resized uint8 grayscale images to the tensors LoFTR.forward consumes: zero padding to [PH,PW] at the
 * bottom / right (pad_bottom_right), `float / 255`, the padding mask and its coarse version
 * (F.interpolate(mask, scale_factor=1/coarse_div, mode='nearest') = mask[d*y, d*x]).
 *   src [N, *, *] uint8 with byte pitches per image / per row; hw [N,2] int32 device: valid (h, w) <= (PH, PW);
 *   image [N,1,PH,PW] f32; mask [N,PH,PW] u8 or NULL; mask_c [N,PH/coarse_div,PW/coarse_div] u8 or NULL.
 * Decoding and cv2.resize stay with the caller (OpenCV; not reproducible without the library). */
/* ---- training-mode glue of the backbone (round 5; csrc/train_glue.hip) -------------------------------------------------------------
 * What sits between the convolutions of a TRAINING step of the ResNet-FPN (src/loftr/backbone/resnet_fpn.py:22-40,66-77,110-116) and was
 * PyTorch autograd until ABI 20.  fp32 tensors of logical shape [N, C, H, W] stored NCHW (channels_last = 0) or NHWC (channels_last = 1:
 * what the convolution nodes of the training path produce and consume -- no layout copy between a convolution and its BatchNorm; C % 4 == 0,
 * C <= 1024); HW = H * W; every reduction is a two-stage sum with float64 partials merged in a fixed order (deterministic).  Not used by the inference path (eval-mode BatchNorm is folded into the convolutions there).
 *
 * loftr_bn_train_fwd: nn.BatchNorm2d in .train() mode: mean / biased variance over (N, H, W) per channel, y = (x - mean) * invstd * gamma
 *   + beta (gamma / beta may be null: affine=False); mean [C], invstd [C] = 1 / sqrt(var + eps) are returned for the backward,
 *   var_unbiased [C] (may be null) is what the caller's running_var update takes (torch: momentum update with the UNBIASED variance).
 *   The reference trains with SyncBatchNorm (train.py:108): the same arithmetic over the union of the ranks' batches; one process here.
 * loftr_bn_train_bwd: dx, dgamma = sum dy * xhat, dbeta = sum dy (batch statistics: mean and variance depend on x).
 *   Workspace of both: loftr_bn_train_workspace_bytes(N, C, HW).
 * loftr_act_fwd: y = act(a + b) (b may be null; y may alias a): act 0 none, 1 ReLU, 2 LeakyReLU(slope) -- BasicBlock's relu(x + y)
 *   (resnet_fpn.py:40) and the heads' LeakyReLU (:70,:76).  loftr_act_bwd: dx = dy * act'(.) from the forward's OUTPUT y (the sign of the
 *   output is the sign of the input for a positive slope; dx is the gradient of a AND of b).
 * loftr_upsample2x_bilinear_fwd / _bwd: F.interpolate(x, scale_factor=2., mode='bilinear', align_corners=True) on N * C maps
 *   of H x W -> 2H x 2W (resnet_fpn.py:110,115) and its adjoint, evaluated as a gather (no atomics: deterministic, unlike torch's). */
size_t loftr_bn_train_workspace_bytes(int N, int C, long HW);
int loftr_bn_train_fwd(const float* x, int N, int C, long HW, int channels_last, const float* gamma, const float* beta, float eps, float* y,
                       float* mean, float* invstd, float* var_unbiased, void* ws, size_t ws_bytes, void* stream);
int loftr_bn_train_bwd(const float* dy, const float* x, int N, int C, long HW, int channels_last, const float* mean, const float* invstd,
                       const float* gamma, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);
int loftr_act_fwd(const float* a, const float* b, long n, int act, float slope, float* y, void* stream);
int loftr_act_bwd(const float* dy, const float* y, long n, int act, float slope, float* dx, void* stream);
int loftr_upsample2x_bilinear_fwd(const float* x, int N, int C, int H, int W, int channels_last, float* y, void* stream);
int loftr_upsample2x_bilinear_bwd(const float* dy, int N, int C, int H, int W, int channels_last, float* dx, void* stream);

/* cv2.resize(image_u8, (dw, dh)) with the default INTER_LINEAR (dataset.py:108,146) on the device, one grayscale image.
 * PARITY UNPINNED: restates OpenCV 4.x's fixed-point bilinear (11-bit coefficients, half-pixel centres); OpenCV is not
 * in this image, so it is verified against the numpy restatement only (oracle/input_oracle.py). */
int loftr_resize_linear_u8(const uint8_t* src, int sh, int sw, long src_pitch, uint8_t* dst, int dh, int dw,
                           long dst_pitch, void* stream);
int loftr_pack_gray_u8(const uint8_t* src, long src_image_pitch, long src_row_pitch, const int* hw, int N, int PH,
                       int PW, float* image, uint8_t* mask, uint8_t* mask_c, int coarse_div, void* stream);

/* ---- multi-GPU: the one data-path collective (SURVEY.md §8(b),(e)) -----------------------------------
 * Replaces the reference's result merge across DDP ranks (test.py:65 + src/lightning/data.py:315 shard the pairs,
 * src/utils/comm.py:113-219 gathers pickled results over gloo): pairs are independent, so all a rank needs from
 * the others is how many matches each of THEIR pairs produced -- an RCCL all-gather of int32[n] per rank over xGMI
 * on the caller's stream (latency bound; no host round trip).  counts_out[r*n + k] = counts_in[k] of rank r.
 * The communicator is an opaque handle bound to the HIP device current at creation; the 128-byte unique id comes
 * from loftr_rccl_unique_id on one rank and reaches the others through the caller's control plane.
 * librccl is dlopen'ed on first use: LOFTR_ERR_COMM if it is missing or any RCCL call fails. */
#define LOFTR_RCCL_ID_BYTES 128
int loftr_rccl_unique_id(char* id_out, size_t id_bytes);
int loftr_rccl_comm_create(const char* id, size_t id_bytes, int rank, int world, void** comm_out);
int loftr_rccl_comm_info(void* comm, int* rank_out, int* world_out);
int loftr_rccl_comm_destroy(void* comm);
int loftr_rccl_allgather_counts(void* comm, const int32_t* counts_in, int32_t* counts_out, int n, void* stream);

/* ---- debug / A-B switches (process-global; the library reads NO environment variable) -------------
 * Named integer switches that select an alternative schedule of the same arithmetic for A/B measurements and tests:
 *   "encoder_schedule"  1: loftr_transformer_fwd runs the coarse level as scheduled launches; 0: call by call in the reference's order
 *   "conv_persist_cap"  0: persistent convolution grids span the device's CUs; n >= 8: at most n workgroups (tests: many tiles each)
 *   "wgrad_chunk"       0: split-K chunk of the weight-gradient GEMMs chosen by shape; n > 0: forced
 *   "reduce_tall"       1: tall partial-sum reductions use the tall kernel; 0: the generic one
 *   "conv_duo"          1: 3x3 / stride-1 convolutions of 128 k / 192 / 224 output columns run conv3x3_duo_kernel; 0: the generic conv3x3_kernel
 *   "conv_rem"          1: loftr_conv_bn_act_prepared_scratch takes the remainder form where it applies; 0: never
 *   "conv_patch"        1: 3x3 / stride-1 convolutions run the patch kernels; 0: the implicit-GEMM kernel of the strided / 1x1 layers
 *   "pct_grid"          0: the persistent coarse transformer runs 256 workgroups (one per CU); n > 0: n workgroups
 *   "pct_quota"         0: its workgroups stay until the queue is empty; n > 0: a workgroup leaves after n work items (yielding variant)
 *   "pct_skip"          0; bit t: work items of type t (0 X, 1 K, 2 F) are popped and signalled but NOT executed (queue tests: wrong results)
 * Unknown key: LOFTR_ERR_BAD_ARG.  Results never depend on a switch beyond the last bits of a floating-point sum order. */
int loftr_hip_debug_set(const char* key, int value);
int loftr_hip_debug_get(const char* key, int* value, int* default_value);

/* ---- per-kernel timing (profiling aid; process-global like the debug switches) ---------------
 * When bit `id` of the mask is set, every launch of that kernel is bracketed by hipEvents
 * recorded on the launch stream (up to 4096 launches between reads).  Replaces the reference's
 * InferenceProfiler (src/utils/profiler.py:7-28: cuda.synchronize()-bracketed wall clocks).
 * loftr_hip_timing_read synchronises the recorded events and returns the accumulated GPU time
 * (ms) and launch count of kernel `id`; reset != 0 clears the accumulators. */
int loftr_hip_timing_enable(unsigned mask);
int loftr_hip_timing_kernel_count(void);
const char* loftr_hip_timing_kernel_name(int id);
int loftr_hip_timing_read(int id, double* total_ms, long long* launches, int reset);

/* ---- fp16-range guard (debug aid, process-global like the timing switch) -----------------------
 * The GEMM chain holds every operand as two fp16 numbers (csrc/gemm.h).  Weights, convolution filters and tensors entering through
 * loftr_linear_fwd / loftr_sp_from_f32_scaled are pre-scaled by powers of two and cannot overflow; the fine-level transformer
 * rescales its windows at run time (csrc/fine_fused.hip).  Activations that enter the chain UNSCALED -- the coarse feature maps after
 * the positional encoding, the coarse descriptors at the matcher, the fine preprocess inputs -- must stay below the fp16 maximum
 * 65504: beyond it the affected products are inf / NaN where the fp32 reference (src/loftr/loftr.py:56-75) is still finite.  With the
 * guard on, every such conversion is followed by a (synchronous) scan and the entry point returns LOFTR_ERR_RANGE instead. */
int loftr_hip_range_check_enable(int on);

/* ---- building block exposed for tests / profiling ------------------------------------------
 * out[M,N] = A[M,K] @ Wt[N,K]^T  (the fp32-accurate split-fp16 MFMA GEMM every linear layer above
 * is built on). */
size_t loftr_linear_workspace_bytes(int M, int N, int K);
int loftr_linear_fwd(const float* a, const float* w, float* out, int M, int N, int K,
                     void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOFTR_HIP_H_ */
