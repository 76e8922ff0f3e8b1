// Shared device/host helpers for the LoFTR matching-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/loftr_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define LOFTR_NEG_INF (-1e9f)   // reference: src/loftr/utils/coarse_matching.py:6 (finite "INF")

#define LOFTR_CHECK_ARG(cond) do { if (!(cond)) return LOFTR_ERR_BAD_ARG; } while (0)
#define LOFTR_CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return LOFTR_ERR_LAUNCH; } while (0)

// gfx950 erratum (found in round 5, tools/micro/pk_opsel_probe.hip, profiles/r05_pk_opsel_probe.txt): a packed-fp32 instruction
// (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) whose op_sel bit for SRC1 is set (low result <- high dword of src1) intermittently
// computes with a wrong operand while ANOTHER wave of the same SIMD has MFMAs in flight (2e-3 of the products wrong in the probe;
// src0 / src2 op_sel and every op_sel_hi form are fine).  hipcc emits that form for "pair x scalar" when the scalar sits in the odd
// register of a pair.  Kernels in which it shows up are compiled without packed fp32 arithmetic; tests/test_isa_audit.py scans the
// ISA of every translation unit for the form.
#ifdef __HIP_DEVICE_COMPILE__
#define LOFTR_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define LOFTR_NO_PACKED_FP32          /* (the host pass does not know the feature) */
#endif

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- optional per-kernel timing (hipEvents on the launch stream; see loftr_hip_timing_* in the header)
enum LoftrTimedKernel {
  LOFTR_T_SCORE_STATS = 0,   // coarse_match.hip: score_stats_kernel   (dual-softmax pass A)
  LOFTR_T_SCORE_CONF = 1,    // coarse_match.hip: score_conf_kernel    (pass B, writes conf_matrix)
  LOFTR_T_PROJ = 2,          // linear.hip: proj_kernel                (q/k/v + feature map)
  LOFTR_T_LINEAR = 3,        // linear.hip: linear_kernel              (mlp.0 + relu, fine merges)
  LOFTR_T_LINEAR_LN = 4,     // linear.hip: linear_ln_kernel           (merge+LN, mlp.2+LN+residual)
  LOFTR_T_KV = 5,            // linear.hip: proj_kv_kernel             (k/v projection + fused KV reduction)
  LOFTR_T_ATTN_APPLY = 6,    // attention.hip: attn_apply_kernel
  LOFTR_T_ATTN_SMALL = 7,    // attention.hip: attn_small_kernel       (fine level)
  LOFTR_T_GATHER = 8,        // fine.hip: gather_windows_kernel
  LOFTR_T_OT_STORE = 9,      // coarse_match.hip: score_store_kernel   (sinkhorn)
  LOFTR_T_CONV = 10,         // conv.hip: conv_kernel               (backbone implicit GEMM: strided / 1x1)
  LOFTR_T_CONV3 = 11,        // conv3x3_duo.h: conv3x3_duo_kernel<Cfg<4,2,4,4,1>> (3x3 stride-1, 128-column tiles), also Cfg<6,2,4,8,2> (192 columns,
                             //   with or without the remainder form) and the generic conv3x3_kernel fallback
  LOFTR_T_CONV3W = 12,       // conv3x3_duo.h: conv3x3_duo_kernel<Cfg<7,2,4,8,2>> (3x3 stride-1, 7 output column tiles: Cout 193..224)
  LOFTR_T_ENCODER_X = 13,   // encoder_fused.hip: encoder_x_kernel  (q proj -> merge + LN -> mlp.0 -> mlp.2 + LN + residual, one launch)
  LOFTR_T_FINE_PAIR = 14,   // fine_fused.hip: fine_pair_kernel     (the whole fine-level transformer of a match, one launch)
  LOFTR_T_COUNT = 15
};
// ---- debug / A-B switches (loftr_hip_debug_set in the header): the library reads no environment variable
enum LoftrDebugKey {
  LOFTR_DBG_ENCODER_SCHEDULE = 0,   // 1 (default): coarse transformer without a plan runs the scheduled launches; 0: the reference's call order
  LOFTR_DBG_CONV_PERSIST_CAP,       // 0 (default): persistent convolution grids ask the device for its CU count; n >= 8: cap them at n workgroups
  LOFTR_DBG_WGRAD_CHUNK,            // 0 (default): split-K chunk of the weight-gradient GEMM chosen by shape; n > 0: forced
  LOFTR_DBG_REDUCE_TALL,            // 1 (default): tall partial-sum reductions of the weight gradients use the tall kernel; 0: the generic one
  LOFTR_DBG_PCT_GRID,               // 0 (default): the persistent coarse transformer runs one workgroup per CU (256); n > 0: n workgroups
  LOFTR_DBG_PCT_SKIP,               // 0 (default); bit t set: work items of type t (0 X, 1 K, 2 F) are popped and signalled but not executed (queue tests: WRONG results)
  LOFTR_DBG_PCT_QUOTA,              // 0 (default): persistent workgroups stay until the queue is empty; n > 0: a workgroup leaves after n items (A/B: sharing the GPU with another stream)
  LOFTR_DBG_CONV_DUO,               // 1 (default): 3x3 / stride-1 convolutions with 128 k / 192 / 224 output columns run conv3x3_duo_kernel; 0: the generic conv3x3_kernel (any Cout)
  LOFTR_DBG_CONV_PATCH,             // 1 (default): 3x3 / stride-1 convolutions run the patch kernels; 0: the implicit-GEMM conv_kernel (the strided / 1x1 path)
  LOFTR_DBG_CONV_REM,               // 1 (default): Cout = 193 .. 199 3x3 / stride-1 layers with a scratch buffer run 192 columns + the tap-decomposed remainder; 0: 224 columns
  LOFTR_DBG_COUNT
};
int loftr_debug_value(int key);
extern unsigned g_loftr_timing_mask;
extern int g_loftr_range_check;      // loftr_hip_range_check_enable: fp16-range guard on unscaled operands (sp_convert.hip)
void loftr_timing_mark(int id, hipStream_t st, bool end);
struct TimedLaunch {          // RAII: records an event pair around the launches in its scope
  int id; hipStream_t st; bool on;
  TimedLaunch(int id_, hipStream_t st_) : id(id_), st(st_), on((g_loftr_timing_mask >> id_) & 1u) {
    if (on) loftr_timing_mark(id, st, false);
  }
  ~TimedLaunch() { if (on) loftr_timing_mark(id, st, true); }
};

// ---- XCD-aware tile order --------------------------------------------------------------------
// MI355X dispatches the workgroups of a launch round-robin over its 8 XCDs (linear id % 8) and every
// XCD has a private 4 MB L2.  Column tiles that share the same A rows must therefore sit on the SAME
// XCD at the same time or the A panel is fetched from HBM once per column tile (measured: 4-6x the
// algorithmic read traffic).  1-D launches of xcd_grid(tiles_m, tiles_n) workgroups are mapped as
//   xcd = id % 8, slot = id / 8  ->  tile_n = slot % tiles_n, tile_m = xcd * ceil(tiles_m / 8) + slot / tiles_n
// so each XCD owns one contiguous eighth of the row tiles and sweeps all column tiles of a row tile
// back to back; consecutive row tiles (which share the 3x3 halo rows in the implicit-GEMM
// convolutions) also stay on one XCD and run close in time.
constexpr int NUM_XCD = 8;
static inline unsigned xcd_grid(int tiles_m, int tiles_n) { return (unsigned)(NUM_XCD * ceil_div(tiles_m, NUM_XCD) * tiles_n); }
__device__ __forceinline__ bool xcd_tile(int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
  const int id = blockIdx.x, xcd = id % NUM_XCD, slot = id / NUM_XCD;
  const int chunk = ceil_div(tiles_m, NUM_XCD);
  tile_n = slot % tiles_n;
  const int local = slot / tiles_n;
  tile_m = xcd * chunk + local;
  return local < chunk && tile_m < tiles_m;
}

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
// Reductions across the 32 lanes of one half-wave (lanes that share lane>>5), result in ALL of them.
// Four DPP steps reduce inside each row of 16 lanes (quad_perm x2, row_half_mirror, row_mirror), one
// v_permlane16_swap (gfx950) exchanges the two rows: no LDS-crossbar ds_bpermute traffic (what
// __shfl_xor compiles to), which the GEMM epilogues would otherwise issue five times per register.
#define LOFTR_DPP(v_, ctrl_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), ctrl_, 0xF, 0xF, true))
#define LOFTR_DPPI(v_, ctrl_) __builtin_amdgcn_update_dpp(0, v_, ctrl_, 0xF, 0xF, true)
__device__ __forceinline__ float half_sum(float v) {
  v += LOFTR_DPP(v, 0xB1);        // quad_perm [1,0,3,2]
  v += LOFTR_DPP(v, 0x4E);        // quad_perm [2,3,0,1]
  v += LOFTR_DPP(v, 0x141);       // row_half_mirror
  v += LOFTR_DPP(v, 0x140);       // row_mirror            -> every lane: sum of its row of 16
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);      // rows (0,1) / (2,3) combined
}
__device__ __forceinline__ float half_max(float v) {
  v = fmaxf(v, LOFTR_DPP(v, 0xB1));
  v = fmaxf(v, LOFTR_DPP(v, 0x4E));
  v = fmaxf(v, LOFTR_DPP(v, 0x141));
  v = fmaxf(v, LOFTR_DPP(v, 0x140));
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  return fmaxf(__int_as_float(r[0]), __int_as_float(r[1]));
}
// first (smallest) index over the half-wave; INT_MAX lanes do not participate
__device__ __forceinline__ int half_min_i32(int v) {
  v = min(v, LOFTR_DPPI(v, 0xB1));
  v = min(v, LOFTR_DPPI(v, 0x4E));
  v = min(v, LOFTR_DPPI(v, 0x141));
  v = min(v, LOFTR_DPPI(v, 0x140));
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return min((int)r[0], (int)r[1]);
}
// The same for the 16 accumulator registers of an MFMA tile at once, stage by stage, so that the DPP
// steps of different registers issue back to back instead of each waiting out its VALU->DPP hazard.
__device__ __forceinline__ void half_sum16(f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += LOFTR_DPP(v[r], 0xB1);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += LOFTR_DPP(v[r], 0x4E);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += LOFTR_DPP(v[r], 0x141);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += LOFTR_DPP(v[r], 0x140);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_int(v[r]), __float_as_int(v[r]), false, false);
    v[r] = __int_as_float(q[0]) + __int_as_float(q[1]);
  }
}
__device__ __forceinline__ void half_max16(f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], LOFTR_DPP(v[r], 0xB1));
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], LOFTR_DPP(v[r], 0x4E));
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], LOFTR_DPP(v[r], 0x141));
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], LOFTR_DPP(v[r], 0x140));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_int(v[r]), __float_as_int(v[r]), false, false);
    v[r] = fmaxf(__int_as_float(q[0]), __int_as_float(q[1]));
  }
}
// value held by lane ^ 32 (the other half-wave): one v_permlane32_swap, no LDS crossbar
__device__ __forceinline__ float swap32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
#define LOFTR_DPP_SWAP32(v_) swap32(v_)

__device__ __forceinline__ float elu1(float x) {     // elu(x)+1, linear_attention.py:10-11
  return x > 0.f ? x + 1.f : __expf(x);               // expm1(x)+1 == exp(x)
}
