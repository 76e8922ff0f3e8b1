// Internal launcher of the on-the-fly-split GEMM of head_grads.hip (used by the coarse heads' backward and by the weight
// gradients of encoder_bwd.hip): out[n][m][c] = alpha * sum_k A[n](m, k) * B[n][k][c], fp32 in / out, C % 32 == 0, C <= 256.
#pragma once
#include "common.h"
int launch_head_grad(const float* a, long a_ld, long a_bs, bool trans, const float* b, long b_ld, long b_bs, float* out, long o_ld,
                     long o_bs, int M, int K, int ktot, int C, int nbatch, float alpha, hipStream_t st);

// dW [O][I] = dy^T act as ordered split-K (part: part_cap >= wgrad_part_floats(T, O, I) floats of scratch, else LOFTR_ERR_WORKSPACE); ordered partial sums; column sums.
size_t wgrad_part_floats(long T, int O, int I);
int launch_wgrad(const float* dy, int O, const float* act, int I, long T, float* dW, float* part, size_t part_cap, hipStream_t st);
int launch_reduce_partials(const float* part, float* out, int P, long stride, long n, hipStream_t st);
size_t colsum_part_floats(long rows, int C);
int launch_colsum(const float* x, long rows, int C, float* out, float* part, hipStream_t st);
