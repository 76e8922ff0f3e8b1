// Linear-attention core launchers (defined in attention.hip).
#pragma once
#include "common.h"

size_t attention_workspace_bytes(int nb, int S, int C);

// Coarse level: Kf/Vf [nb,S,C] -> kv [nb,8,33,32] (KV + Ksum) and pm [nb,C,C] (KV folded into the
// merge weight); both live in `ws`.
int launch_attention_kv(const float* Kf, const float* Vf, const float* merge_w, int nb, int S, int C, int H,
                        void* ws, size_t ws_bytes, const float** kv_out, const float** pm_out, hipStream_t st);

// Qf [nb,L,C], Kf/Vf [nb,S,C]: outputs of the projection kernel (feature map, masks and the
// 1/S scaling already applied).  msg [nb,L,C].
int launch_linear_attention(const float* Qf, const float* Kf, const float* Vf, float* msg, int nb,
                            int L, int S, int C, int H, void* ws, size_t ws_bytes, hipStream_t st);
