// Host-side launchers of the fp32-MFMA linear layers (defined in linear.hip).
#pragma once
#include "gemm.h"

// Epilogue selector of linear_kernel.
enum LinearEpi {
  EPI_STORE = 0,       // out = acc
  EPI_RELU = 1,        // out = max(acc, 0)                         (mlp.1, transformer.py:28)
  EPI_BIAS = 2,        // out = acc + bias[col]                     (fine_preprocess down_proj)
  EPI_GROUP_BIAS = 3,  // out = acc + gbias[(row / group) , col]    (merge_feat: the coarse half of
                       //   the concatenation is constant per window, fine_preprocess.py:53-56)
};

struct LinearArgs {
  ASrc a;
  const float* w; int ldw;      // [N, K] row-major
  float* out; int ldo;
  int M, N, K;
  const float* bias;            // EPI_BIAS: [N]; EPI_GROUP_BIAS: [M/group, N]
  int group;
};
int launch_linear(const LinearArgs& p, LinearEpi epi, hipStream_t st);

// q/k/v projections with the linear-attention feature map fused
// (transformer.py:47-49 + linear_attention.py:31-42).  Up to three weight segments of C output
// columns each share the A operand; segment s writes out[s] [M, C]:
//   kind 0 (Q): elu(v)+1, times q-mask          kind 1 (K): elu(v)+1, times kv-mask
//   kind 2 (V): v times kv-mask, divided by S
struct ProjArgs {
  const float* a; int M; int C;          // A [M, C]
  int nseg;
  const float* w[3]; float* out[3]; int kind[3];
  const uint8_t* mask;                   // [M] or null (same rows as A)
  float inv_s;                           // 1 / v_length
};
int launch_proj(const ProjArgs& p, hipStream_t st);

// out = [residual +] LayerNorm(A @ W^T) * gamma + beta     (N == C, one block spans the row)
//   merge + norm1 (transformer.py:51-52) and mlp.2 + norm2 + residual (:55-58).
//   Batched-attention mode (attn_kv != null, C == 256): A is the feature-mapped Q [nb, L, C], the B
//   operand of batch element n is w + n*C*C (= P_n, attention.hip) and the linear-attention
//   normaliser is applied to A on the fly (gemm.h: AttnXform); M = L rows per batch element,
//   grid.y = nb so no tile straddles two batch elements.
struct LinearLNArgs {
  ASrc a;
  const float* w; int ldw;
  const float* gamma; const float* beta;
  const float* residual;        // [M, C] or null
  float* out;                   // [M, C]
  int M, C, K;
  float eps;
  const float* attn_kv;         // [nb, 8, 33, 32] or null
  int nb; float v_length; float attn_eps;
};
int launch_linear_ln(const LinearLNArgs& p, hipStream_t st);
