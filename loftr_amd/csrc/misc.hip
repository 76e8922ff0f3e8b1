// Position encoding + NCHW -> N(HW)C flatten, library entry points that are not kernels.
//   reference: src/loftr/utils/position_encoding.py:37-42, src/loftr/loftr.py:58-59
#include "common.h"
#include <string.h>

namespace {
// 32 x 32 tile transpose through LDS: read rows of [C][HW] (coalesced along HW), write rows of
// [HW][C] (coalesced along C).  The constant sine table is added on the way.
//   grid (ceil(HW/32), ceil(C/32), N), block (32, 8)
//   CL = channels-last input (sc == 1): only the sine table goes through the transpose, the
//   feature value is read in the output's own (coalesced along C) order.
template <bool CL>
__global__ void pos_encode_flatten_kernel(loftr_fmap f, const float* __restrict__ pe,
                                          int pe_h, int pe_w, float* __restrict__ out, int C) {
  __shared__ float tile[32][33];
  const int H = f.H, W = f.W, HW = H * W;
  const int n = blockIdx.z, hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const float* in = f.data + (long)n * f.sn;
  for (int k = threadIdx.y; k < 32; k += 8) {
    const int c = c0 + k, hw = hw0 + threadIdx.x;
    float v = 0.f;
    if (c < C && hw < HW) {
      const int y = hw / W, x = hw - y * W;
      v = pe[((long)c * pe_h + y) * pe_w + x];
      if (!CL) v += in[(long)c * f.sc + (long)y * f.sh + (long)x * f.sw];
    }
    tile[k][threadIdx.x] = v;
  }
  __syncthreads();
  for (int k = threadIdx.y; k < 32; k += 8) {
    const int hw = hw0 + k, c = c0 + threadIdx.x;
    if (hw < HW && c < C) {
      float v = tile[threadIdx.x][k];
      if (CL) { const int y = hw / W, x = hw - y * W; v += in[(long)y * f.sh + (long)x * f.sw + c]; }
      out[((long)n * HW + hw) * C + c] = v;
    }
  }
}
}  // namespace

extern "C" int loftr_pos_encode_flatten(const loftr_fmap* feat, const float* pe, int pe_h, int pe_w, float* out,
                                        int N, int C, void* stream) {
  LOFTR_CHECK_ARG(feat && feat->data && pe && out && N >= 0 && C > 0 && feat->H > 0 && feat->W > 0 &&
                  feat->H <= pe_h && feat->W <= pe_w);
  if (N == 0) return LOFTR_OK;
  const dim3 grid(ceil_div(feat->H * feat->W, 32), ceil_div(C, 32), N), block(32, 8);
  if (feat->sc == 1)
    hipLaunchKernelGGL((pos_encode_flatten_kernel<true>), grid, block, 0, (hipStream_t)stream, *feat, pe, pe_h, pe_w, out, C);
  else
    hipLaunchKernelGGL((pos_encode_flatten_kernel<false>), grid, block, 0, (hipStream_t)stream, *feat, pe, pe_h, pe_w, out, C);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ---- debug / A-B switches -----------------------------------------------------------------
namespace {
struct DebugSwitch { const char* name; int def; int value; };
DebugSwitch g_debug[LOFTR_DBG_COUNT] = {{"encoder_schedule", 1, 1}, {"conv_persist_cap", 0, 0}, {"wgrad_chunk", 0, 0}, {"reduce_tall", 1, 1},
                                        {"pct_grid", 0, 0}, {"pct_skip", 0, 0}, {"pct_quota", 0, 0}, {"conv_duo", 1, 1}, {"conv_patch", 1, 1}, {"conv_rem", 1, 1}};
}  // namespace
int loftr_debug_value(int key) { return (key >= 0 && key < LOFTR_DBG_COUNT) ? g_debug[key].value : 0; }
extern "C" int loftr_hip_debug_set(const char* key, int value) {
  LOFTR_CHECK_ARG(key != nullptr);
  for (auto& d : g_debug)
    if (strcmp(d.name, key) == 0) { d.value = value; return LOFTR_OK; }
  return LOFTR_ERR_BAD_ARG;
}
extern "C" int loftr_hip_debug_get(const char* key, int* value, int* default_value) {
  LOFTR_CHECK_ARG(key != nullptr && value != nullptr);
  for (auto& d : g_debug)
    if (strcmp(d.name, key) == 0) { *value = d.value; if (default_value) *default_value = d.def; return LOFTR_OK; }
  return LOFTR_ERR_BAD_ARG;
}

// ---- per-kernel timing ---------------------------------------------------------------------
unsigned g_loftr_timing_mask = 0;
int g_loftr_range_check = 0;
extern "C" int loftr_hip_range_check_enable(int on) { g_loftr_range_check = on ? 1 : 0; return LOFTR_OK; }
namespace {
constexpr int T_POOL = 4096;                 // event pairs per kernel id; recording stops when full
struct TimingSlot {
  hipEvent_t ev[T_POOL][2];
  int created, used;
  double total_ms; long long launches;
};
TimingSlot g_slots[LOFTR_T_COUNT];
const char* const T_NAMES[LOFTR_T_COUNT] = {"score_sweep_kernel<0>", "score_sweep_kernel<1>", "proj_kernel", "linear_kernel",
                                            "linear_ln_kernel", "proj_kv_kernel", "attn_apply_kernel",
                                            "attn_small_kernel", "gather_windows_kernel", "score_sweep_kernel<2>", "conv_kernel",
                                            "conv3x3_duo_kernel<Cfg<4,2,4,4,1>>", "conv3x3_duo_kernel<Cfg<7,2,4,8,2>>", "encoder_x_kernel", "fine_pair_kernel"};
void timing_drain(TimingSlot& s) {
  for (int i = 0; i < s.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(s.ev[i][1]) == hipSuccess && hipEventElapsedTime(&ms, s.ev[i][0], s.ev[i][1]) == hipSuccess) {
      s.total_ms += ms;
      s.launches += 1;
    }
  }
  s.used = 0;
}
}  // namespace

void loftr_timing_mark(int id, hipStream_t st, bool end) {
  TimingSlot& s = g_slots[id];
  if (!end) {
    if (s.used >= T_POOL) return;
    if (s.used >= s.created) {
      if (hipEventCreate(&s.ev[s.created][0]) != hipSuccess || hipEventCreate(&s.ev[s.created][1]) != hipSuccess) return;
      s.created++;
    }
    (void)hipEventRecord(s.ev[s.used][0], st);
  } else {
    if (s.used >= s.created) return;
    (void)hipEventRecord(s.ev[s.used][1], st);
    s.used++;
  }
}

extern "C" int loftr_hip_timing_enable(unsigned mask) {
  g_loftr_timing_mask = mask & ((1u << LOFTR_T_COUNT) - 1u);
  return LOFTR_OK;
}

extern "C" int loftr_hip_timing_kernel_count(void) { return LOFTR_T_COUNT; }

extern "C" const char* loftr_hip_timing_kernel_name(int id) {
  return (id >= 0 && id < LOFTR_T_COUNT) ? T_NAMES[id] : "";
}

extern "C" int loftr_hip_timing_read(int id, double* total_ms, long long* launches, int reset) {
  LOFTR_CHECK_ARG(id >= 0 && id < LOFTR_T_COUNT && total_ms && launches);
  TimingSlot& s = g_slots[id];
  timing_drain(s);
  *total_ms = s.total_ms;
  *launches = s.launches;
  if (reset) { s.total_ms = 0.0; s.launches = 0; }
  return LOFTR_OK;
}

extern "C" int loftr_hip_abi_version(void) { return LOFTR_HIP_ABI_VERSION; }

extern "C" const char* loftr_hip_status_string(int status) {
  switch (status) {
    case LOFTR_OK: return "ok";
    case LOFTR_ERR_BAD_ARG: return "bad argument (null pointer or invalid shape)";
    case LOFTR_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case LOFTR_ERR_WORKSPACE: return "workspace too small";
    case LOFTR_ERR_LAUNCH: return "HIP kernel launch failed";
    case LOFTR_ERR_NO_DEVICE: return "no gfx950 (MI355X) device";
    case LOFTR_ERR_COMM: return "RCCL unavailable or a communicator / collective call failed";
    case LOFTR_ERR_RANGE: return "an activation entering the split-fp16 GEMM chain is not below the fp16 maximum 65504 (or not finite)";
    default: return "unknown status";
  }
}

extern "C" int loftr_hip_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return LOFTR_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return LOFTR_ERR_NO_DEVICE;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? LOFTR_OK : LOFTR_ERR_NO_DEVICE;
}
