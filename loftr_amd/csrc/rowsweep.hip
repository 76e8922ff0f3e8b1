// K = 256 linear layers of the coarse encoder as a STATIONARY-WEIGHT sweep (round 2).
//   reference: src/loftr/loftr_module/transformer.py:47-52 (q_proj, merge + norm1), linear_attention.py:31-47
//
// The tiled GEMM core (gemm.h) gives a 128 x 128 tile of a K = 256 product eight k-tiles of work between a DMA prologue
// and an epilogue that touches every accumulator: the q projection (30 GFLOP executed) took 111 us and merge + LayerNorm
// 158 us at B = 8 -- 0.2-0.3 PF of the ~1.1 PF the same loop sustains on long k.  Here the loop is turned inside out, as
// for the score volume (score_sweep.h, namespace sweep): the WEIGHT matrix [256 out, 256 in] is the stationary operand
// -- wave w of eight keeps the (hi, lo) MFMA fragments of its 32 output features for the whole K in 128 VGPRs -- and the
// workgroup sweeps it over 32-token panels of the activation (32 KB each) that the eight waves share through a four-stage
// LDS ring filled by global_load_lds two panels ahead (counted s_waitcnt vmcnt, one s_barrier per panel = per 48 MFMAs).
// With the weights as the MFMA's A operand and the panel as B, a LANE owns one token and its 16 accumulator registers
// are 16 of the wave's 32 features, four consecutive ones per register quad (the other 16 sit in lane ^ 32):
//   * q projection: the wave IS a head (D = 32), so the linear-attention normaliser Q . Ksum is 16 lane-private FMAs and
//     one half-wave exchange; the SP output leaves as 8-byte stores;
//   * merge + LayerNorm: per-wave (mean, M2) of the token's 32 features go through LDS, the eight partials are combined
//     with the exact parallel-variance formula (two-pass quality, ONE exchange) and the normalised row is written one
//     panel later, after the next panel's barrier made every wave's partials visible -- no extra barrier.
// A workgroup handles PPU consecutive panels of one sequence; tokens are independent, so the result does not depend on
// the decomposition (bitwise batch invariance).  grid = 8 XCDs x ceil(nseq / 8) x units, 512 threads, one per CU.
#include "linear.h"

namespace {
namespace rsw {
constexpr int W = 8, PT = 32, KS = 16, STAGE = PT * 1024, NST = 4, MAXPPU = 32;
constexpr int OFF_TAB = NST * STAGE;                      // float [2][256] per-feature tables
constexpr int OFF_MASK = OFF_TAB + 2 * 256 * 4;           // uint8 [MAXPPU * PT] token masks of the unit
constexpr int OFF_STAT = OFF_MASK + MAXPPU * PT;          // float2 [2][W][PT] LayerNorm partials (mean, M2)
constexpr int LDS_BYTES = OFF_STAT + 2 * W * PT * 8;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
constexpr int DMA_PER_WAVE = PT * 8 / 8 / W;              // global_load_lds instructions per wave per panel (4)

struct Args {
  const sp_t* a; const sp_t* w; long w_seq_stride;        // tokens [nseq][T][256] SP; weights [256][256] SP (+ seq * stride)
  int nseq, T, PPU, units;
  const float* wsc;                                       // [256] inverse row scales of W or null
  const uint8_t* mask;                                    // MODE 0: [nseq * T] or null
  const float* kv;                                        // MODE 0: [nseq][8][33][32], row 32 of a head = Ksum
  float v_length, eps;
  const float* gamma; const float* beta;                  // MODE 1
  float out_scale, ln_eps;
  sp_t* out;                                              // [nseq * T][256] SP
};

// register r of a lane in half-wave g holds feature 32 wave + 8 (r >> 2) + 4 g + (r & 3) of the lane's token

// four consecutive features -> the 8-byte hi piece and the 8-byte lo piece of their SP group
__device__ __forceinline__ void sp_pack4(float x0, float x1, float x2, float x3, uint2& hi, uint2& lo) {
  const uint32_t a = sp_pack(x0), b = sp_pack(x1), c = sp_pack(x2), d = sp_pack(x3);
  hi = make_uint2((a & 0xffffu) | (b << 16), (c & 0xffffu) | (d << 16));
  lo = make_uint2((a >> 16) | (b & 0xffff0000u), (c >> 16) | (d & 0xffff0000u));
}

template <int MODE>      // 0: q projection + elu+1 + mask + attention normaliser   1: LayerNorm(out_scale * A W^T) * gamma + beta
__global__ __launch_bounds__(512, 2) void rowsweep_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  // ---- unit: sequences are dealt to the XCDs (a sequence's weights / P stay in one L2), its units run back to back there
  const int id = blockIdx.x, xcd = id % NUM_XCD, slot = id / NUM_XCD;
  const int seq = (slot / a.units) * NUM_XCD + xcd, unit = slot % a.units;
  if (seq >= a.nseq) return;
  const int T = a.T, NP = (T + PT - 1) / PT;
  const int p0 = unit * a.PPU, np = min(a.PPU, NP - p0);
  if (np <= 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  const sp_t* an = a.a + (long)seq * T * 256;
  const sp_t* wn = a.w + (long)seq * a.w_seq_stride;

  // ---- stationary operand: this lane's 16-byte MFMA fragments of weight row 32 wave + li, all 16 k-steps, hi and lo
  h16x8 sh[KS], sl[KS];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(wn + (long)(wave * 32 + li) * 256);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = (ks >> 1) * 8 + 2 * (ks & 1) + g;
      sh[ks] = __builtin_bit_cast(h16x8, src[c]);
      sl[ks] = __builtin_bit_cast(h16x8, src[c + 4]);
    }
  }
  // ---- per-feature tables and the unit's token masks -> LDS (ordinary loads / LDS stores only here, before any DMA)
  float* tab = reinterpret_cast<float*>(lds + OFF_TAB);
  uint8_t* mask_s = reinterpret_cast<uint8_t*>(lds + OFF_MASK);
  if (threadIdx.x < 256) {
    const int f = threadIdx.x;
    if (MODE == 0) {
      tab[f] = a.wsc ? a.wsc[f] : 1.f;
      tab[256 + f] = a.kv[((long)seq * 8 + (f >> 5)) * (33 * 32) + 32 * 32 + (f & 31)];
    } else {
      tab[f] = a.gamma[f];
      tab[256 + f] = a.beta[f];
    }
  }
  if (MODE == 0) {
    for (int t = threadIdx.x; t < np * PT; t += 512)
      mask_s[t] = a.mask ? a.mask[(long)seq * T + min(p0 * PT + t, T - 1)] : 1;
  }

#define RSW_DOFF(oct_) ((oct_) * 8 * 256 + (lane >> 3) * 256 + (((lane & 7) ^ (((oct_) * 4 + (lane >> 4)) & 7)) << 2) + wave * 32)
#define RSW_ISSUE(p_)                                                                                    \
  {                                                                                                      \
    const int tok0__ = (p0 + (p_)) * PT;                                                                 \
    char* st__ = lds + ((p_) & (NST - 1)) * STAGE + wave * 4096;                                         \
    if (tok0__ + PT <= T) {                                                                              \
      const sp_t* base__ = an + (long)tok0__ * 256;                                                      \
      _Pragma("unroll") for (int oct__ = 0; oct__ < 4; ++oct__)                                          \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base__ + RSW_DOFF(oct__)), (lds_ptr_t)(st__ + oct__ * 1024), 16, 0, 0); \
    } else {                  /* last panel of the sequence: rows beyond T re-read row T-1 (never stored) */ \
      _Pragma("unroll") for (int oct__ = 0; oct__ < 4; ++oct__) {                                        \
        const int r__ = oct__ * 8 + (lane >> 3);                                                         \
        const int gt__ = min(tok0__ + r__, T - 1);                                                       \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(an + (long)gt__ * 256 + (RSW_DOFF(oct__) - r__ * 256)), \
                                         (lds_ptr_t)(st__ + oct__ * 1024), 16, 0, 0);                    \
      }                                                                                                  \
    }                                                                                                    \
  }

  const int a_off = lds_chunk_off(li, g);          // hi chunk of the even k-step; odd k-step: ^ 32, lo: ^ 64
  const int fq = wave * 32 + 4 * g;                // first feature of register quad 0 (quad q: + 8 q)
  const unsigned stat_base = (unsigned)(size_t)(lds_ptr_t)(lds + OFF_STAT);
  float pend[16];                                  // MODE 1: the previous panel's values, normalised one panel later
#pragma unroll
  for (int r = 0; r < 16; ++r) pend[r] = 0.f;

  // Stores (8 x 8 bytes) of the token `tok_` of this lane: features fq + 8 q .. + 3 -> dwords (wave * 32 + 4 q + 2 g) [+ 16]
#define RSW_STORE(y_, tok_)                                                                              \
  if ((tok_) < T) {                                                                                      \
    sp_t* o__ = a.out + ((long)seq * T + (tok_)) * 256 + wave * 32 + 2 * g;                              \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
      uint2 hi__, lo__;                                                                                  \
      sp_pack4(y_[4 * q], y_[4 * q + 1], y_[4 * q + 2], y_[4 * q + 3], hi__, lo__);                      \
      *reinterpret_cast<uint2*>(o__ + 4 * q) = hi__;                                                     \
      *reinterpret_cast<uint2*>(o__ + 16 + 4 * q) = lo__;                                                \
    }                                                                                                    \
  }
  // MODE 1: normalise `pend` (panel pp_) with the eight waves' partials of stat slot (pp_ & 1) and store it
#define RSW_FINALIZE(pp_)                                                                                \
  {                                                                                                      \
    const float2* sp__ = reinterpret_cast<const float2*>(lds + OFF_STAT) + ((pp_) & 1) * W * PT + li;    \
    float2 e__[W];                                                                                       \
    _Pragma("unroll") for (int k = 0; k < W; ++k) e__[k] = sp__[k * PT];                                 \
    float mean__ = 0.f;                                                                                  \
    _Pragma("unroll") for (int k = 0; k < W; ++k) mean__ += e__[k].x;                                    \
    mean__ *= 1.f / W;                                                                                   \
    float m2__ = 0.f;                                                                                    \
    _Pragma("unroll") for (int k = 0; k < W; ++k) { const float d = e__[k].x - mean__; m2__ += e__[k].y + 32.f * d * d; } \
    const float rstd__ = rsqrtf(m2__ * (1.f / 256.f) + a.ln_eps);                                        \
    float y__[16];                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
      const f32x4 ga = *reinterpret_cast<const f32x4*>(tab + fq + 8 * q);                                \
      const f32x4 be = *reinterpret_cast<const f32x4*>(tab + 256 + fq + 8 * q);                          \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) y__[4 * q + e] = (pend[4 * q + e] - mean__) * rstd__ * ga[e] + be[e]; \
    }                                                                                                    \
    RSW_STORE(y__, (p0 + (pp_)) * PT + li)                                                               \
  }

  // Every ordinary load above must be COMPLETE before the first DMA is issued (see score_sweep.h: a first use inside
  // the panel loop would make hipcc drain the in-flight DMA every iteration).
  LOFTR_WAITCNT_VM(0);
  __syncthreads();                                 // tables visible; no DMA in flight yet
  RSW_ISSUE(0);
  if (np > 1) RSW_ISSUE(1);
  constexpr int ST = 8;                            // stores a wave issues per panel
  f32x16 acc0, acc1;
  for (int p = 0; p < np; ++p) {
    // panel p has landed once at most {DMA of panel p+1, the stores of the previous iteration} are outstanding
    // (VMEM operations retire in order).  Iteration p issues stores from p = 0 on (MODE 0) / from p = 1 on (MODE 1: it
    // writes panel p - 1), so the first iteration whose wait may count them is p = 1 / p = 2.
    if (p + 1 >= np) LOFTR_WAITCNT_VM(0);
    else if (p < 1 + MODE) LOFTR_WAITCNT_VM(DMA_PER_WAVE);
    else LOFTR_WAITCNT_VM(DMA_PER_WAVE + ST);
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LayerNorm partials (asm ds_write) are in LDS
    __builtin_amdgcn_s_barrier();                  // ... for every wave; and every wave is past the MFMAs of panel p-2
    if (p + 2 < np) RSW_ISSUE(p + 2);
    // ---- 48 MFMAs: D[feature][token] += W[feature][k] * X[token][k], two accumulator chains
    const char* st = lds + (p & (NST - 1)) * STAGE;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const char* sk = st + (ks >> 1) * 4096;
      const h16x8 xh = *reinterpret_cast<const h16x8*>(sk + (a_off ^ ((ks & 1) ? 32 : 0)));
      const h16x8 xl = *reinterpret_cast<const h16x8*>(sk + (a_off ^ ((ks & 1) ? 96 : 64)));
      if (ks & 1) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[ks], xl, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl[ks], xh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[ks], xh, acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[ks], xl, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl[ks], xh, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[ks], xh, acc0, 0, 0, 0);
      }
    }
    // ---- epilogue of panel p: lane = token p0 * 32 + 32 p + li, registers = features fq + jr-order
    const int tok = (p0 + p) * PT + li;
    if (MODE == 0) {
      const float mk = mask_s[p * PT + li] ? 1.f : 0.f;
      float v[16];
      float den = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ws = *reinterpret_cast<const f32x4*>(tab + fq + 8 * q);
        const f32x4 ks4 = *reinterpret_cast<const f32x4*>(tab + 256 + fq + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = (acc0[4 * q + e] + acc1[4 * q + e]) * ws[e];
          x = x > 0.f ? x + 1.f : __expf(x);                       // elu(x) + 1     linear_attention.py:10-11,31
          x *= mk;                                                 // Q * q_mask      :35-36
          v[4 * q + e] = x;
          den = fmaf(x, ks4[e], den);
        }
      }
      den += swap32(den);                                          // the head's other 16 channels
      const float z = a.v_length * __builtin_amdgcn_rcpf(den + a.eps);   // Z = 1 / (Q . Ksum + eps), times v_length   :44-45
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] *= z;
      RSW_STORE(v, tok)
    } else {
      if (p > 0) RSW_FINALIZE(p - 1)
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { pend[r] = (acc0[r] + acc1[r]) * a.out_scale; s += pend[r]; }
      s += swap32(s);
      const float mw = s * (1.f / 32.f);
      float m2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = pend[r] - mw; m2 = fmaf(d, d, m2); }
      m2 += swap32(m2);
      // post (mean, M2) of this wave's 32 features of token li: an asm LDS store (a compiler-visible one would make hipcc
      // drain the in-flight DMA first), made visible by the lgkmcnt(0) + barrier at the top of the next iteration
      if (g == 0) {
        const unsigned addr = stat_base + (unsigned)((((p & 1) * W + wave) * PT + li) * 8);
        const f32x2 pr = {mw, m2};
        asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(pr) : "memory");
      }
    }
  }
  if (MODE == 1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    RSW_FINALIZE(np - 1)
  }
#undef RSW_ISSUE
#undef RSW_DOFF
#undef RSW_STORE
#undef RSW_FINALIZE
}

int plan(int nseq, int T, Args& a) {
  const int NP = ceil_div(T, PT);
  int ppu = ceil_div(nseq * NP, 256);               // about one workgroup per CU
  ppu = ppu < 1 ? 1 : (ppu > MAXPPU ? MAXPPU : ppu);
  a.nseq = nseq; a.T = T; a.PPU = ppu; a.units = ceil_div(NP, ppu);
  return NUM_XCD * ceil_div(nseq, NUM_XCD) * a.units;
}
}  // namespace rsw
}  // namespace

// LOFTR_ROWSWEEP=0 keeps the tiled kernels (A/B)
static bool rowsweep_enabled() {
  static const bool on = []() { const char* e = getenv("LOFTR_ROWSWEEP"); return !(e && atoi(e) == 0); }();
  return on;
}

// q projection of the coarse level (ProjArgs with kv: one segment, kind 0, C = 256).  LOFTR_ERR_UNSUPPORTED -> the caller
// uses proj_kernel.
int launch_rowsweep_q(const ProjArgs& p, hipStream_t st) {
  if (!rowsweep_enabled() || p.C != 256 || p.nseg != 1 || p.kind[0] != 0 || !p.kv || p.M <= 0 || p.nbatch <= 0) return LOFTR_ERR_UNSUPPORTED;
  rsw::Args a{};
  const int grid = rsw::plan(p.nbatch, p.M, a);
  a.a = p.a; a.w = p.w[0]; a.w_seq_stride = 0; a.wsc = p.wsc[0]; a.mask = p.mask; a.kv = p.kv; a.v_length = p.v_length; a.eps = p.eps;
  a.out = reinterpret_cast<sp_t*>(p.out[0]);
  TimedLaunch tl(LOFTR_T_PROJ, st);
  hipLaunchKernelGGL((rsw::rowsweep_kernel<0>), dim3(grid), dim3(512), 0, st, a);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// merge + norm1 of the coarse level (LinearLNArgs: plain A of width 256, per-batch B = P, SP output only).
int launch_rowsweep_ln(const LinearLNArgs& p, hipStream_t st) {
  if (!rowsweep_enabled() || p.C != 256 || p.K != 256 || p.ldw != 256 || p.a.ld0 != 256 || p.a.p1 || p.a.gather || p.residual || p.out_f32 ||
      !p.out_sp || p.wscale_inv || p.M <= 0 || p.nbatch <= 0 || (p.nbatch > 1 && p.w_batch_stride != 256L * 256))
    return LOFTR_ERR_UNSUPPORTED;
  rsw::Args a{};
  const int grid = rsw::plan(p.nbatch, p.M, a);
  a.a = p.a.p0; a.w = p.w; a.w_seq_stride = p.nbatch > 1 ? p.w_batch_stride : 0; a.gamma = p.gamma; a.beta = p.beta;
  a.out_scale = p.out_scale != 0.f ? p.out_scale : 1.f; a.ln_eps = p.eps; a.out = p.out_sp;
  TimedLaunch tl(LOFTR_T_LINEAR_LN, st);
  hipLaunchKernelGGL((rsw::rowsweep_kernel<1>), dim3(grid), dim3(512), 0, st, a);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
