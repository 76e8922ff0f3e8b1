// Linear-attention core launchers (defined in attention.hip).
#pragma once
#include "gemm.h"

// P (the merged attention projection, SP) is stored multiplied by this power of two (attention.hip: kv_finalize_kernel)
constexpr float ATTN_P_SCALE = 32.f;

size_t attention_workspace_bytes(int nb, int S, int C);

// Coarse level (C = 256, 8 heads of 32): finalize of the linear-attention reduction.
//   part [nb,8,splits,33,32] (written by linear.hip: proj_kv_kernel, splits = ceil(S / 128)) lives at the start
//   of `ws` (attention_part_buffer);  outputs, also in `ws`:
//   kv [nb,8,33,32] fp32 (rows 0..31 = KV[d][v], row 32 = Ksum[d]) and
//   pm [nb,C,C] SP      (KV folded into the merge weight: the per-pair B operand of the fused
//                        attention + merge GEMM).
float* attention_part_buffer(void* ws, size_t ws_bytes, int nb, int S);
int launch_attention_finalize(const float* merge_w, int nb, int S, int C, int H, void* ws, size_t ws_bytes,
                              const float** kv_out, const sp_t** pm_out, hipStream_t st);

// Fine level (per-match windows, C = 128, 8 heads of 16): whole attention of one window per block.
//   Qf [nb,L,C], Kf/Vf [nb,S,C] fp32 -> msg [nb,L,C] SP.
int launch_attention_small(const float* Qf, const float* Kf, const float* Vf, sp_t* msg, int nb, int L, int S,
                           int C, int H, hipStream_t st);
