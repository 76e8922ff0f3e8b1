// Host-side launchers of the linear layers on the split-fp16 GEMM core (defined in linear.hip).
// Every A / W operand is an SP tensor (gemm.h); outputs are fp32 and / or SP.
#pragma once
#include "gemm.h"

// out = [relu](A @ W^T [+ bias]) written as fp32 (out_f32) and / or SP (out_sp), row pitch ldo.
//   bias_mode 0: none   1: bias[col]   2: bias[(row / group) * N + col]  (merge_feat: the coarse
//   half of the concatenation is constant per window, fine_preprocess.py:53-56)
struct LinearArgs {
  ASrc a;
  const sp_t* w; int ldw;       // [N, K] SP
  float* out_f32; sp_t* out_sp; int ldo;
  int M, N, K;
  const float* bias; int bias_mode; int group;
  bool relu;
  const float* wscale_inv;      // [N] or null: W row n was stored multiplied by a power of two, this is its inverse (gemm.h)
  const float* ascale_inv;      // [M] or null: the same for the rows of A (loftr_linear_fwd only)
};
int launch_linear(const LinearArgs& p, hipStream_t st);

// q/k/v projections with the linear-attention feature map fused
// (transformer.py:47-49 + linear_attention.py:31-42).  Up to three weight segments of C output
// columns each share the A operand; segment s writes out[s]:
//   kind 0 (Q): elu(v)+1, times q-mask.  With `kv` (coarse level, D == 32): additionally scaled by the
//               linear-attention normaliser z[l,h] = S / (Q[l,h,:] . Ksum[h,:] + eps) and written as SP
//               (the A operand of the fused attention+merge GEMM, linear_ln_kernel); else fp32.
//   kind 1 (K): elu(v)+1, times kv-mask -> fp32
//   kind 2 (V): v times kv-mask, divided by S -> fp32
// Rows are organised as nbatch batch elements of M rows (grid.z = nbatch) so that no tile straddles
// two batch elements when per-batch data (kv) is used; pass nbatch = 1, M = all rows otherwise.
struct ProjArgs {
  const sp_t* a; int M; int C; int nbatch;
  int nseg;
  const sp_t* w[3]; void* out[3]; int kind[3];
  const uint8_t* mask;                   // [nbatch * M] or null (same rows as A)
  float inv_s;                           // 1 / v_length
  const float* kv;                       // [nbatch, 8, 33, 32] (row 32 of each head = Ksum) or null
  float v_length, eps;
  const float* wsc[3];                   // per segment: [C] inverse row scales of w[seg] or null
};
int launch_proj(const ProjArgs& p, hipStream_t st);

// Coarse level (C = 256, 8 heads of 32): k / v projections of the source with the KV reduction of linear
// attention fused into the epilogue -- K and V never go to HBM.
//   w_kv: SP [2C, C], rows interleaved per head [K_h (32 rows) | V_h (32 rows)] (transformer.hip stages it so),
//   so a 128-column tile holds two heads and each wave's 64 columns are [K_h | V_h] of ONE head for its 64 rows;
//   the epilogue applies elu+1 / masks / 1/S and contracts over the rows with 32 fp32 MFMAs fed straight from the
//   accumulators (KV_h[d][v] += K[row][d] * V[row][v]), plus Ksum.  Output: per (batch element, head, row tile)
//   partials part[nb][8][splits][33][32] (rows 0..31 = KV[d][v], row 32 = Ksum[d]), splits = ceil(S / 128);
//   kv_finalize_kernel (attention.hip) sums them.
struct ProjKVArgs {
  const sp_t* a; int S; int C; int nbatch;       // source [nbatch, S, C] SP
  const sp_t* w_kv;
  const uint8_t* mask;                           // [nbatch * S] or null
  float inv_s;
  float* part; int splits;
  const float* wsc;                              // [2C] inverse row scales of w_kv (same interleaved row order) or null
};
int launch_proj_kv(const ProjKVArgs& p, hipStream_t st);

// out = [residual +] LayerNorm(A @ W^T) * gamma + beta     (N == C, one block spans the row)
//   merge + norm1 (transformer.py:51-52) and mlp.2 + norm2 + residual (:55-58).
//   Written as fp32 (out_f32) and / or SP (out_sp).  With w_batch_stride != 0 the rows are nbatch
//   batch elements of M rows each with its own B operand w + n * w_batch_stride (the per-pair
//   merged attention projection P_n, attention.hip); grid.y = nbatch.
struct LinearLNArgs {
  ASrc a;
  const sp_t* w; int ldw;
  const float* gamma; const float* beta;
  const float* residual;        // [rows, C] fp32 or null
  float* out_f32; sp_t* out_sp; // [rows, C]
  int M, C, K;
  float eps;
  int nbatch; long w_batch_stride;
  const float* wscale_inv;      // [C] inverse row scales of W or null
  float out_scale;              // 0 = none; else every accumulator is multiplied by it first (P of the merged attention
                                // projection is stored times a fixed power of two, attention.hip)
};
int launch_linear_ln(const LinearLNArgs& p, hipStream_t st);

// The whole x side of a coarse encoder layer (q projection + feature map + normaliser, merge with the per-sequence P +
// norm1, mlp.0 + ReLU, mlp.2 + norm2 + residual) in one launch with the tokens stationary in registers (encoder_fused.hip).
// C = 256 only; LOFTR_ERR_UNSUPPORTED otherwise (C = 128: the per-GEMM kernels of transformer.hip: encoder_layer).
struct EncoderXArgs {
  const sp_t* x_sp; const float* x_f32; float* out_f32; sp_t* out_sp;   // [nseq * T, C]; out may alias x
  int nseq, T, C;
  const sp_t* wq; const sp_t* pm; long pm_seq_stride; const sp_t* w0; const sp_t* w2;
  const float *wq_s, *w0_s, *w2_s;                    // inverse row scales of wq / w0 / w2
  const float* kv;                                    // [nseq, 8, 33, 32]
  const uint8_t* mask;                                // [nseq * T] or null
  const float *g1, *b1, *g2, *b2;
  float v_length, attn_eps, p_out_scale, ln_eps;
  int skip_padded;                                    // != 0 (in-place calls with a mask only): a 128-token tile whose mask bytes are all zero keeps its input
};
int launch_encoder_x(const EncoderXArgs& p, hipStream_t st);
// Workgroups of the job (0: not a shape the fused kernel takes), and a launch of workgroups [off0, off0 + n0) of job p0 followed by
// [off1, off1 + n1) of job p1 (offsets / n0 multiples of 8; n1 == 0: p0 alone): two independent calls share rounds of 256 workgroups
int encoder_x_workgroups(const EncoderXArgs& p);
int launch_encoder_x2(const EncoderXArgs& p0, int off0, int n0, const EncoderXArgs& p1, int off1, int n1, hipStream_t st);

// The whole fine-level transformer (layers [self, cross]) on M window pairs of T <= 32 tokens, C = 128, in one launch
// (fine_fused.hip).  [l] = layer 0 (self) / 1 (cross).  LOFTR_ERR_UNSUPPORTED for any other shape.
struct FinePairArgs {
  float* f0; float* f1; int M, T, C;                  // [M, T, C] fp32, updated in place
  const sp_t *wq[2], *wk[2], *wv[2], *wm[2], *w0[2], *w2[2];
  const float *sq[2], *sk[2], *sv[2], *sm[2], *s0[2], *s2[2];     // inverse row scales
  const float *g1[2], *b1[2], *g2[2], *b2[2];
  float attn_eps, ln_eps;
};
int launch_fine_pair(const FinePairArgs& p, hipStream_t st);
