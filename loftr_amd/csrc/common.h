// Shared device/host helpers for the LoFTR matching-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/loftr_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LOFTR_NEG_INF (-1e9f)   // reference: src/loftr/utils/coarse_matching.py:6 (finite "INF")

#define LOFTR_CHECK_ARG(cond) do { if (!(cond)) return LOFTR_ERR_BAD_ARG; } while (0)
#define LOFTR_CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return LOFTR_ERR_LAUNCH; } while (0)

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- optional per-kernel timing (hipEvents on the launch stream; see loftr_hip_timing_* in the header)
enum LoftrTimedKernel {
  LOFTR_T_SCORE_STATS = 0,   // coarse_match.hip: score_stats_kernel   (dual-softmax pass A)
  LOFTR_T_SCORE_CONF = 1,    // coarse_match.hip: score_conf_kernel    (pass B, writes conf_matrix)
  LOFTR_T_PROJ = 2,          // linear.hip: proj_kernel                (q/k/v + feature map)
  LOFTR_T_LINEAR = 3,        // linear.hip: linear_kernel              (mlp.0 + relu, fine merges)
  LOFTR_T_LINEAR_LN = 4,     // linear.hip: linear_ln_kernel           (merge+LN, mlp.2+LN+residual)
  LOFTR_T_KV = 5,            // attention.hip: kv_partial_kernel
  LOFTR_T_ATTN_APPLY = 6,    // attention.hip: attn_apply_kernel
  LOFTR_T_ATTN_SMALL = 7,    // attention.hip: attn_small_kernel       (fine level)
  LOFTR_T_GATHER = 8,        // fine.hip: gather_windows_kernel
  LOFTR_T_OT_STORE = 9,      // coarse_match.hip: score_store_kernel   (sinkhorn)
  LOFTR_T_COUNT = 10
};
extern unsigned g_loftr_timing_mask;
void loftr_timing_mark(int id, hipStream_t st, bool end);
struct TimedLaunch {          // RAII: records an event pair around the launches in its scope
  int id; hipStream_t st; bool on;
  TimedLaunch(int id_, hipStream_t st_) : id(id_), st(st_), on((g_loftr_timing_mask >> id_) & 1u) {
    if (on) loftr_timing_mark(id, st, false);
  }
  ~TimedLaunch() { if (on) loftr_timing_mark(id, st, true); }
};

// Bump allocator over the caller-supplied workspace (the library never mallocs device memory).
struct WsAlloc {
  char* base; size_t cap; size_t off;
  WsAlloc(void* p, size_t bytes) : base((char*)p), cap(bytes), off(0) {}
  template <typename T> T* take(size_t n) {
    off = align_up(off, 256);
    T* r = (T*)(base + off);
    off += n * sizeof(T);
    return r;
  }
  bool ok() const { return off <= cap; }
};

// ---- wave64 reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduce across the 32 lanes of one half-wave (lanes that share lane>>5)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float elu1(float x) {     // elu(x)+1, linear_attention.py:10-11
  return x > 0.f ? x + 1.f : __expf(x);               // expm1(x)+1 == exp(x)
}
