// Linear-attention core launchers (defined in attention.hip).
#pragma once
#include "common.h"

size_t attention_workspace_bytes(int nb, int S, int C);

// Qf [nb,L,C], Kf/Vf [nb,S,C]: outputs of the projection kernel (feature map, masks and the
// 1/S scaling already applied).  msg [nb,L,C].
int launch_linear_attention(const float* Qf, const float* Kf, const float* Vf, float* msg, int nb,
                            int L, int S, int C, int H, void* ws, size_t ws_bytes, hipStream_t st);
