// The whole fine-level LocalFeatureTransformer (layer "self", layer "cross") of one match in ONE kernel (round 3).
//   reference: src/loftr/loftr.py:71-72 (loftr_fine on the [M, WW, 128] window pairs), loftr_module/transformer.py:35-58,80-101,
//              linear_attention.py:20-47
//
// At the fine level attention never leaves a match: a window has WW = 25 tokens and the two windows of a match only ever
// attend to each other -- w0' = Enc1(w0; w0), w1' = Enc1(w1; w1), w0'' = Enc2(w0'; w1'), w1'' = Enc2(w1'; w0'') -- so the
// whole two-layer transformer of a match is local to ONE wave.  Until round 2 it ran as 15 launches (q/k/v projections written
// as fp32 and read back, attn_small_kernel, merge + LN, mlp.0, mlp.2 + LN, three times) moving ~6 GB per step through HBM.
// Here (same scheme as encoder_fused.hip): a workgroup is four waves, one per SIMD; a wave owns one match, holds both
// windows as MFMA fragments in registers (lane = token, 32 slots for the 25 tokens) and runs the four encoder calls back to
// back; the four waves share ONE stream of weight panels (16 KB = 4 blocks of 32 rows x 128 B) through a six-stage LDS
// ring filled by global_load_lds five panels ahead, one barrier per panel (24 MFMAs per wave).  Per call and match:
//   * K = elu(src Wk^T) + 1 and V = src Wv^T / S are computed in the TRANSPOSED orientation (activation as the MFMA's A operand:
//     lane = feature, registers = tokens), 32 features = two heads at a time; tokens beyond WW are zeroed; K and V tiles are split
//     into (hi, lo) halves IN their register order -- the contraction over tokens does not care about the order as long as both
//     operands use the same -- and KV = K^T V is six fp16 MFMAs per head pair (the off-diagonal head blocks are discarded);
//     Ksum goes through a 512-byte per-wave LDS scratch;
//   * Q = z (.) (elu(x Wq^T) + 1) per head pair, then message_h = KV_h^T Q_h (six MFMAs against the block-diagonal KV fragments)
//     and at once merge: msg += Wm[:, 32 t ..] message_t  -- neither Q nor the attention output ever exist as a whole;
//   * LayerNorm1, then per 32 hidden features hid = relu(W0 [x, msg]) -> out += W2[:, hp] hid, LayerNorm2, residual.
// HBM traffic: both windows in (fp32), both out, plus the fp32 residual copy below: ~100 KB per match instead of ~770 KB.
// Residual stream in fp32 (round 4): the GEMM operands are the (hi, lo) fragments (22 significant bits), but the residual
// `x + LayerNorm2(..)` (transformer.py:58) adds the window's fp32 rows -- every call writes its output rows to f0 / f1 in place
// (the wave owns them) and the call that updates the window next re-reads them (L2-resident, issued ahead of the LayerNorm
// statistics).  Round 3 rebuilt x from its fragments there: 22 bits in the residual path, measured 4.1e-4 px from the
// reference's fp64 run on e2e_synth where the reference's own fp32 run sits at 2.5e-4 (profiles/r04_parity_margins.txt).
#include "linear.h"

// timing probes (python -m loftr_amd.build --variant fprobe -DLOFTR_FINE_PROBE [-DFFX_PROBE_READS=0 ...]; tools/micro/fine_probe.py): per-wave
// phase sums of the 100 MHz wall clock; the 0 settings give WRONG results: fragment reads only in front of a panel / no weight DMA after the
// prologue / no per-panel barrier
#ifdef LOFTR_FINE_PROBE
__device__ long long* g_fine_probe = nullptr;   // set by loftr_fine_probe_buffer: per wave 6 phase sums, start, end
#define FFX_STAMP(i_) { const long long t__ = wall_clock64(); pacc[i_] += t__ - pt__; pt__ = t__; }
#else
#define FFX_STAMP(i_)
#endif
#ifndef FFX_PROBE_READS
#define FFX_PROBE_READS 1
#endif
#ifndef FFX_PROBE_DMA
#define FFX_PROBE_DMA 1
#endif
#ifndef FFX_PROBE_BARRIER
#define FFX_PROBE_BARRIER 1
#endif
namespace {
namespace ffx {
constexpr int W = 4, STAGE = 16 * 1024, BLK = 4096, NST = 6, DMA_PER_WAVE = 4;
constexpr int PPC = 8 + 8 + 24, NCALL = 4, NPANEL = PPC * NCALL;     // per call: 4 x (Wk, Wv), 4 x (Wq, Wm), 8 x (W0a, W0b, W2)
// per-layer tables (floats)
constexpr int T_QS = 0, T_KS = 128, T_VS = 256, T_MS = 384, T_W0S = 512, T_W2S = 768, T_G1 = 896, T_B1 = 1024, T_G2 = 1152,
              T_B2 = 1280, T_LAYER = 1408;
constexpr int OFF_TAB = NST * STAGE, OFF_KSUM = OFF_TAB + 2 * T_LAYER * 4;     // per wave: Ksum[128] floats
constexpr int OFF_KV = OFF_KSUM + W * 512;                                     // per wave: 4 head pairs x (hi, lo) x 64 lanes x 16 B
constexpr int OFF_DESC = OFF_KV + W * 8192;                                    // per panel: {source offset (bytes, from the base of its
constexpr int LDS_BYTES = OFF_DESC + NPANEL * 16;                              //  matrix), row pitch (dwords), matrix id, K-type flag}
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");

struct Args {
  float* f0; float* f1;                                  // [M][T][128] fp32, updated in place
  int M, T;
  const sp_t* wq[2]; const sp_t* wk[2]; const sp_t* wv[2]; const sp_t* wm[2]; const sp_t* w0[2]; const sp_t* w2[2];   // SP row-major
  const float* sq[2]; const float* sk[2]; const float* sv[2]; const float* sm[2]; const float* s0[2]; const float* s2[2];
  const float* g1[2]; const float* b1[2]; const float* g2[2]; const float* b2[2];
  float attn_eps, ln_eps;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b) {     // see encoder_fused.hip
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void sp_pack4(float x0, float x1, float x2, float x3, uint2& hi, uint2& lo) {
  sp_pack2(x0, x1, hi.x, lo.x);
  sp_pack2(x2, x3, hi.y, lo.y);
}
// D layout of a 32-row tile (register r of half-wave g = row 8 (r >> 2) + 4 g + (r & 3), lane = column) -> the lane's MFMA
// fragments over the ROW index (k-step s, element e = row 16 s + 8 g + e), as in encoder_fused.hip
__device__ __forceinline__ void pack_panel(const float (&v)[16], h16x8 (&fh)[2], h16x8 (&fl)[2]) {
  uint2 H[4], L[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) sp_pack4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], H[q], L[q]);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    swap_halves(H[2 * s].x, H[2 * s + 1].x); swap_halves(H[2 * s].y, H[2 * s + 1].y);
    swap_halves(L[2 * s].x, L[2 * s + 1].x); swap_halves(L[2 * s].y, L[2 * s + 1].y);
    fh[s] = __builtin_bit_cast(h16x8, u32x4{H[2 * s].x, H[2 * s].y, H[2 * s + 1].x, H[2 * s + 1].y});
    fl[s] = __builtin_bit_cast(h16x8, u32x4{L[2 * s].x, L[2 * s].y, L[2 * s + 1].x, L[2 * s + 1].y});
  }
}
// the inverse (the exchange is an involution): fragments of one 32-feature panel -> its 16 D-layout values hi + lo
__device__ __forceinline__ void unpack_panel(const h16x8 (&fh)[2], const h16x8 (&fl)[2], float (&v)[16]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const u32x4 Hv = __builtin_bit_cast(u32x4, fh[s]), Lv = __builtin_bit_cast(u32x4, fl[s]);
    uint32_t H[4] = {Hv[0], Hv[1], Hv[2], Hv[3]}, L[4] = {Lv[0], Lv[1], Lv[2], Lv[3]};
    swap_halves(H[0], H[2]); swap_halves(H[1], H[3]); swap_halves(L[0], L[2]); swap_halves(L[1], L[3]);
    // quad 2 s = dwords 0, 1; quad 2 s + 1 = dwords 2, 3; dword = two consecutive values
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int r = 8 * s + 2 * d;
      v[r] = (float)__builtin_bit_cast(_Float16, (uint16_t)(H[d] & 0xffffu)) + (float)__builtin_bit_cast(_Float16, (uint16_t)(L[d] & 0xffffu));
      v[r + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(H[d] >> 16)) + (float)__builtin_bit_cast(_Float16, (uint16_t)(L[d] >> 16));
    }
  }
}
// power of two that lifts `absmax` into [2^13, 2^14) (1 for ~0 / non-finite), and its inverse: the exact operand scaling of gemm.h,
// applied at run time to operands produced inside the kernel whose magnitude is not bounded by a LayerNorm (V / S, K^T V, z Q
// reach 1e-2 .. 1e3 with backbone-sized window features; unscaled their fp16 halves are subnormal at one end and overflow at the other)
__device__ __forceinline__ float pow2_lift(float absmax, float& inv) {
  const int e = (int)((__float_as_uint(absmax) >> 23) & 0xffu) - 126;      // absmax = m 2^e, m in [0.5, 1)
  int sh = 14 - e;
  sh = (absmax > 1e-30f && absmax < 1e30f) ? sh : 0;
  inv = __uint_as_float((unsigned)(127 - sh) << 23);
  return __uint_as_float((unsigned)(127 + sh) << 23);
}
__device__ __forceinline__ float wave_absmax16(const float (&v)[16]) {
  float m = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(v[r]));
  m = half_max(m);
  return fmaxf(m, swap32(m));
}
// eight values in REGISTER order -> one (hi, lo) fragment pair (element e = value e)
__device__ __forceinline__ void pack8(const float (&v)[8], h16x8& fh, h16x8& fl) {
  uint32_t hi[4], lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) sp_pack2(v[2 * e], v[2 * e + 1], hi[e], lo[e]);
  fh = __builtin_bit_cast(h16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
  fl = __builtin_bit_cast(h16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
}

// One head-pair tile of the source side: K = elu(k) + 1 and V = v / S (tokens beyond the window zeroed) in the transposed layout
// (lane = feature, register r of half-wave g = token 8 (r >> 2) + 4 g + (r & 3)) -> Ksum to the wave's LDS scratch, KV = K^T V on
// the matrix cores, its block-diagonal fragments (one (hi, lo) pair per lane) to the wave's LDS scratch.  LDS stores are inline
// asm: a compiler-visible LDS store makes hipcc drain the in-flight weight DMA first.  Returns what undoes the tile's scales.
__device__ __forceinline__ float kv_tile(const f32x16& kacc, const f32x16& vacc, float ksc, float vsc, int T, int g, int li,
                                         unsigned ksum_ad, unsigned kv_ad) {
  float kk[16], vv[16], ksum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool ok = (r & 3) + 8 * (r >> 2) + 4 * g < T;             // tokens beyond the window contribute nothing
    const float kx = kacc[r] * ksc;
    kk[r] = ok ? (kx > 0.f ? kx + 1.f : __expf(kx)) : 0.f;          // elu + 1     linear_attention.py:31-33
    vv[r] = ok ? vacc[r] * vsc : 0.f;                               // values / v_length   :41-42
    ksum += kk[r];
  }
  ksum += swap32(ksum);
  float v_inv;
  const float v_sc = pow2_lift(wave_absmax16(vv), v_inv);           // one exponent per tile: factors out of K^T V
#pragma unroll
  for (int r = 0; r < 16; ++r) vv[r] *= v_sc;
  if (g == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(ksum_ad), "v"(ksum) : "memory");
  // K^T V over the tokens: both tiles split in REGISTER order (k-step s = registers 8 s .. 8 s + 7)
  h16x8 kfh[2], kfl[2], vfh[2], vfl[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float k8[8] = {kk[8 * s], kk[8 * s + 1], kk[8 * s + 2], kk[8 * s + 3], kk[8 * s + 4], kk[8 * s + 5], kk[8 * s + 6], kk[8 * s + 7]};
    const float v8[8] = {vv[8 * s], vv[8 * s + 1], vv[8 * s + 2], vv[8 * s + 3], vv[8 * s + 4], vv[8 * s + 5], vv[8 * s + 6], vv[8 * s + 7]};
    pack8(k8, kfh[s], kfl[s]);
    pack8(v8, vfh[s], vfl[s]);
  }
  f32x16 kv;
#pragma unroll
  for (int r = 0; r < 16; ++r) kv[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {                                      // kv[d][v]: lane = v, register = d
    kv = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[s], vfl[s], kv, 0, 0, 0);
    kv = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[s], vfh[s], kv, 0, 0, 0);
    kv = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[s], vfh[s], kv, 0, 0, 0);
  }
  float kvv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) kvv[r] = kv[r];
  float kv_inv;
  const float kv_sc = pow2_lift(wave_absmax16(kvv), kv_inv);
#pragma unroll
  for (int r = 0; r < 16; ++r) kvv[r] *= kv_sc;
  h16x8 fh[2], fl[2];
  pack_panel(kvv, fh, fl);                                           // A fragments of KV^T: lane = v, k = d = 16 s + 8 g + e
  // head 2 t = rows / columns 0 .. 15 of the tile, head 2 t + 1 = 16 .. 31: a lane keeps the k-step of ITS head only
  const u32x4 mh = __builtin_bit_cast(u32x4, li < 16 ? fh[0] : fh[1]), ml = __builtin_bit_cast(u32x4, li < 16 ? fl[0] : fl[1]);
  asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024" :: "v"(kv_ad), "v"(mh), "v"(ml) : "memory");
  return v_inv * kv_inv;
}

__global__ __launch_bounds__(W * 64, 1) void fine_pair_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  const int T = a.T;
  const long m = (long)blockIdx.x * W + wave;
  const bool live = m < a.M;                             // wave-uniform
  const long mc = live ? m : a.M - 1;
  float* tab = reinterpret_cast<float*>(lds + OFF_TAB);
  const unsigned ksum_addr = (unsigned)(size_t)(lds_ptr_t)(lds + OFF_KSUM + wave * 512);
  const float* ksum_tab = reinterpret_cast<const float*>(lds + OFF_KSUM + wave * 512);
  const unsigned kv_addr = (unsigned)(size_t)(lds_ptr_t)(lds + OFF_KV + wave * 8192);
  const char* kv_tab = lds + OFF_KV + wave * 8192 + lane * 16;

  // ---- tables of both layers -> LDS (before any DMA)
  for (int f = threadIdx.x; f < 256; f += W * 64) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      float* tl = tab + l * T_LAYER;
      tl[T_W0S + f] = a.s0[l][f];
      if (f < 128) {
        tl[T_QS + f] = a.sq[l][f]; tl[T_KS + f] = a.sk[l][f]; tl[T_VS + f] = a.sv[l][f]; tl[T_MS + f] = a.sm[l][f];
        tl[T_W2S + f] = a.s2[l][f];
        tl[T_G1 + f] = a.g1[l][f]; tl[T_B1 + f] = a.b1[l][f]; tl[T_G2 + f] = a.g2[l][f]; tl[T_B2 + f] = a.b2[l][f];
      }
    }
  }
  // ---- both windows of this wave's match -> fragments (lane = token li, element e of k-step ks = feature 16 ks + 8 g + e)
  // Each window carries ONE power-of-two scale (its largest entry lifted to [2^13, 2^14): gemm.h's operand scaling, chosen at run
  // time per window and per layer): every product that is linear in the window is evaluated on the scaled fragments and the
  // exact inverse is applied where the result leaves the linear part (feature map, LayerNorm, residual).
  h16x8 wah[8], wal[8], wbh[8], wbl[8];
  float wa_sc, wa_inv, wb_sc, wb_inv;
  {
    const long row = (mc * T + min(li, T - 1)) * 128;
    const f32x4* p0 = reinterpret_cast<const f32x4*>(a.f0 + row);
    const f32x4* p1 = reinterpret_cast<const f32x4*>(a.f1 + row);
    f32x4 ua[16], ub[16];
    float ma = 0.f, mb = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      ua[i] = p0[4 * (i >> 1) + 2 * g + (i & 1)]; ub[i] = p1[4 * (i >> 1) + 2 * g + (i & 1)];
#pragma unroll
      for (int e = 0; e < 4; ++e) { ma = fmaxf(ma, fabsf(ua[i][e])); mb = fmaxf(mb, fabsf(ub[i][e])); }
    }
    ma = half_max(ma); ma = fmaxf(ma, swap32(ma)); mb = half_max(mb); mb = fmaxf(mb, swap32(mb));
    wa_sc = pow2_lift(ma, wa_inv); wb_sc = pow2_lift(mb, wb_inv);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const f32x4 u0 = ua[2 * ks] * wa_sc, u1 = ua[2 * ks + 1] * wa_sc, v0 = ub[2 * ks] * wb_sc, v1 = ub[2 * ks + 1] * wb_sc;
      const float xa[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
      const float xb[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      pack8(xa, wah[ks], wal[ks]);
      pack8(xb, wbh[ks], wbl[ks]);
    }
  }

  // ---- the weight stream: panel p (16 KB = 4 blocks of 32 rows x 128 B) lives in ring stage p % NST
  //   R panel: block b = k-group b of the panel's 32 rows;  K panel: block b = rows 32 b .. + 31 of the panel's k-group.
  //   wave w issues block w (4 DMA instructions of 8 rows each).
  // DMA addressing: a lane's BYTE offset inside a panel block for the two row pitches that occur (128 / 256 dwords), so that an issue is
  // `global_load_lds v_offset, s[base]` -- SGPR base + 32-bit VGPR offset, no per-lane 64-bit address arithmetic (round 6: the address
  // chains were 135 of the ~300 VALU instructions of an mlp iteration)
  //   pitch 128: off128[oct];  pitch 256: off128[oct] + (lane >> 3) * 512 per lane, + oct * 4096 on the scalar side
  unsigned off128[4];
#pragma unroll
  for (int oct = 0; oct < 4; ++oct) {
    const unsigned ro = oct * 8 + (lane >> 3), ch = ((lane & 7) ^ ((oct * 4 + (lane >> 4)) & 7)) << 2;
    off128[oct] = (ro * 128 + ch) * 4;
  }
  const unsigned rl512 = (unsigned)(lane >> 3) * 512u;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  // where panel p comes from: decoded once per workgroup into an LDS table (the divisions by 40 and 3 per panel and wave were a
  // few hundred scalar instructions per barrier interval): {address of its first row / k-group (lo, hi), row pitch, K-type}
  for (int p = threadIdx.x; p < NPANEL; p += W * 64) {
    const int c = p / PPC, q = p - c * PPC, l = c >> 1;
    int off, pitch = 128, mat, kt = 0;
    if (q < 8) { mat = (q & 1) ? 2 : 1; off = (q >> 1) * 32 * 128; }                                  // Wk / Wv rows 32 t ..
    else if (q < 16) { const int t = (q - 8) >> 1; if (q & 1) { mat = 3; off = t * 32; kt = 1; } else { mat = 0; off = t * 32 * 128; } }
    else { const int hp = (q - 16) / 3, i = (q - 16) - 3 * hp; pitch = 256;
           if (i == 2) { mat = 5; off = hp * 32; kt = 1; } else { mat = 4; off = hp * 32 * 256 + i * 128; } }
    const sp_t* mp = mat == 0 ? a.wq[l] : mat == 1 ? a.wk[l] : mat == 2 ? a.wv[l] : mat == 3 ? a.wm[l] : mat == 4 ? a.w0[l] : a.w2[l];
    const unsigned long long ad = (unsigned long long)(size_t)(mp + off);
    reinterpret_cast<int4*>(lds + OFF_DESC)[p] = make_int4((int)(unsigned)ad, (int)(unsigned)(ad >> 32), pitch, kt);
  }
#define FFX_ISSUE(p_)                                                                                      \
  {                                                                                                        \
    const int p__ = (p_);                                                                                  \
    const int4 d__ = reinterpret_cast<const int4*>(lds + OFF_DESC)[p__];                                   \
    const unsigned lo__ = (unsigned)__builtin_amdgcn_readfirstlane(d__.x), hi__ = (unsigned)__builtin_amdgcn_readfirstlane(d__.y); \
    const int pitch__ = __builtin_amdgcn_readfirstlane(d__.z), kt__ = __builtin_amdgcn_readfirstlane(d__.w);        \
    const unsigned long long wo__ = (unsigned long long)(kt__ ? wave_s * 32 * pitch__ : wave_s * 32) * 4ull;   /* this wave's block, bytes (uniform) */ \
    const char* base__ = reinterpret_cast<const char*>((size_t)((((unsigned long long)hi__ << 32) | lo__) + wo__)); \
    char* st__ = lds + (p__ % NST) * STAGE + wave_s * BLK;                                                 \
    const unsigned w256__ = pitch__ == 256 ? 1u : 0u;                                                      \
    _Pragma("unroll") for (int oct__ = 0; oct__ < 4; ++oct__)                                              \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base__ + (size_t)(w256__ * (unsigned)(oct__ * 4096)) + (off128[oct__] + w256__ * rl512)), \
                                       (lds_ptr_t)(st__ + oct__ * 1024), 16, 0, 0);                        \
  }
  // panel p has landed once at most the NST - 2 newer panels' DMAs are outstanding; the barrier makes every wave's share
  // visible and proves every wave is past panel p - 1, whose stage panel p + NST - 1 then overwrites
#define FFX_BEGIN(p_)                                                                                      \
  {                                                                                                        \
    const int rem__ = NPANEL - 1 - (p_);                                                                   \
    if (!FFX_PROBE_DMA) LOFTR_WAITCNT_VM(0);                                                               \
    else if (rem__ >= 4) LOFTR_WAITCNT_VM(4 * DMA_PER_WAVE);                                                    \
    else if (rem__ == 3) LOFTR_WAITCNT_VM(3 * DMA_PER_WAVE);                                               \
    else if (rem__ == 2) LOFTR_WAITCNT_VM(2 * DMA_PER_WAVE);                                               \
    else if (rem__ == 1) LOFTR_WAITCNT_VM(1 * DMA_PER_WAVE);                                               \
    else LOFTR_WAITCNT_VM(0);                                                                              \
    if (FFX_PROBE_BARRIER) __builtin_amdgcn_s_barrier();                                                   \
    if (FFX_PROBE_DMA && (p_) + NST - 1 < NPANEL) FFX_ISSUE((p_) + NST - 1);                                                \
  }
  const int a_off = lds_chunk_off(li, g);
#define FFX_RD(st_, blk_, odd_, lo_) (*reinterpret_cast<const h16x8*>((st_) + (blk_) * BLK + (a_off ^ (((odd_) ? 32 : 0) | ((lo_) ? 64 : 0)))))
#define FFX_USE(a_, b_, c_, d_) asm volatile("" :: "v"(a_), "v"(b_), "v"(c_), "v"(d_))
  // One panel = four units of (four fragment reads, six MFMAs); the next unit's reads are issued behind the first two MFMAs.
  // FFX_UNITS(st_, BLK0_, ODD0_, BLK1_, ODD1_, M1 .. M6): unit u reads fragments (a: block BLK0_(u), k-step half ODD0_(u);
  // b: BLK1_(u), ODD1_(u)), hi and lo each; M1 .. M6 are the six MFMA statements over ah__, al__, bh__, bl__.
#define FFX_UNITS(st_, BLK0_, ODD0_, BLK1_, ODD1_, M1_, M2_, M3_, M4_, M5_, M6_)                           \
  {                                                                                                        \
    h16x8 ah__ = FFX_RD(st_, BLK0_(0), ODD0_(0), 0), al__ = FFX_RD(st_, BLK0_(0), ODD0_(0), 1);            \
    h16x8 bh__ = FFX_RD(st_, BLK1_(0), ODD1_(0), 0), bl__ = FFX_RD(st_, BLK1_(0), ODD1_(0), 1);            \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                        \
      FFX_USE(ah__, al__, bh__, bl__);                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      h16x8 nah__ = ah__, nal__ = al__, nbh__ = bh__, nbl__ = bl__;                                        \
      M1_;                                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (FFX_PROBE_READS && u + 1 < 4) { nah__ = FFX_RD(st_, BLK0_(u + 1), ODD0_(u + 1), 0); nal__ = FFX_RD(st_, BLK0_(u + 1), ODD0_(u + 1), 1); } \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      M2_;                                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (FFX_PROBE_READS && u + 1 < 4) { nbh__ = FFX_RD(st_, BLK1_(u + 1), ODD1_(u + 1), 0); nbl__ = FFX_RD(st_, BLK1_(u + 1), ODD1_(u + 1), 1); } \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      M3_; M4_; M5_; M6_;                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      ah__ = nah__; al__ = nal__; bh__ = nbh__; bl__ = nbl__;                                              \
    }                                                                                                      \
  }
#define FFX_MF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C_, 0, 0, 0)
#define FFX_ID(u_) (u_)
#define FFX_ZERO(u_) 0
#define FFX_ONE(u_) 1
  // R panel (32 weight rows x 128 k): unit u = k-group u = k-steps 2u (a), 2u + 1 (b).
  //   normal:      acc[feature][token] += W . X      (weights = A operand)
  //   transposed:  acc[token][feature] += X . W      (activation = A operand)
#define FFX_RPANEL(st_, xh_, xl_, acc_)                                                                    \
  FFX_UNITS(st_, FFX_ID, FFX_ZERO, FFX_ID, FFX_ONE,                                                        \
            FFX_MF(ah__, xl_[2 * u], acc_), FFX_MF(al__, xh_[2 * u], acc_), FFX_MF(ah__, xh_[2 * u], acc_), \
            FFX_MF(bh__, xl_[2 * u + 1], acc_), FFX_MF(bl__, xh_[2 * u + 1], acc_), FFX_MF(bh__, xh_[2 * u + 1], acc_))
#define FFX_TPANEL(st_, xh_, xl_, acc_)                                                                    \
  FFX_UNITS(st_, FFX_ID, FFX_ZERO, FFX_ID, FFX_ONE,                                                        \
            FFX_MF(xh_[2 * u], al__, acc_), FFX_MF(xl_[2 * u], ah__, acc_), FFX_MF(xh_[2 * u], ah__, acc_), \
            FFX_MF(xh_[2 * u + 1], bl__, acc_), FFX_MF(xl_[2 * u + 1], bh__, acc_), FFX_MF(xh_[2 * u + 1], bh__, acc_))
  // K panel (128 weight rows x 32 k): unit u = (output panels 2 (u >> 1) (a), 2 (u >> 1) + 1 (b), k-step u & 1)
#define FFX_KB0(u_) (2 * ((u_) >> 1))
#define FFX_KB1(u_) (2 * ((u_) >> 1) + 1)
#define FFX_KODD(u_) ((u_) & 1)
#define FFX_KPANEL(st_, fh_, fl_, out_)                                                                    \
  FFX_UNITS(st_, FFX_KB0, FFX_KODD, FFX_KB1, FFX_KODD,                                                     \
            FFX_MF(ah__, fl_[u & 1], out_[2 * (u >> 1)]), FFX_MF(bh__, fl_[u & 1], out_[2 * (u >> 1) + 1]), \
            FFX_MF(al__, fh_[u & 1], out_[2 * (u >> 1)]), FFX_MF(bl__, fh_[u & 1], out_[2 * (u >> 1) + 1]), \
            FFX_MF(ah__, fh_[u & 1], out_[2 * (u >> 1)]), FFX_MF(bh__, fh_[u & 1], out_[2 * (u >> 1) + 1]))

  LOFTR_WAITCNT_VM(0);                                  // windows, tables: complete before the first DMA
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NST - 1; ++p) FFX_ISSUE(p);

#ifdef LOFTR_FINE_PROBE
  long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long pstart__ = wall_clock64();
  long long pt__ = pstart__;
#endif
  const int fq = 4 * g;
  const float S = (float)T, inv_s = 1.f / (float)T;      // v_length = number of source tokens (linear_attention.py:41-45)
  const h16x8 zero8 = __builtin_bit_cast(h16x8, u32x4{0u, 0u, 0u, 0u});
  f32x16 acc, acc2;
  f32x16 big[4];

#pragma unroll 1
  for (int c = 0; c < NCALL; ++c) {
    const int p0 = c * PPC;
    const float* tl = tab + (c >> 1) * T_LAYER;
    const bool self = c < 2;
    // ============ source side: KV_h = K_h^T V_h, Ksum_h for the four head pairs ================================
    // (the source is this window in the self layer, the other one in the cross layer: the panel loop is instantiated for both
    //  register sets -- a per-call copy of the source cost 64 registers and the moves)
    const float src_sc = self ? wa_sc : wb_sc, src_inv = self ? wa_inv : wb_inv;
    float kvi[4] = {1.f, 1.f, 1.f, 1.f};                  // what undoes a tile's two power-of-two scales (V, then KV)
#define FFX_KV_LOOP(SH_, SL_)                                                                              \
    _Pragma("unroll 1") for (int t = 0; t < 4; ++t) {                                                      \
      const int p = p0 + 2 * t;                                                                            \
      FFX_BEGIN(p);                                                                                        \
      if (live) {                                                                                          \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                       \
        FFX_TPANEL(lds + (p % NST) * STAGE, SH_, SL_, acc);       /* K tile: lane = feature 32 t + li, register = token */ \
      }                                                                                                    \
      FFX_BEGIN(p + 1);                                                                                    \
      if (live) {                                                                                          \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc2[r] = 0.f;                                      \
        FFX_TPANEL(lds + ((p + 1) % NST) * STAGE, SH_, SL_, acc2);                       /* V tile */      \
        const float inv__ = kv_tile(acc, acc2, tl[T_KS + 32 * t + li] * src_inv, tl[T_VS + 32 * t + li] * inv_s * src_inv, T, g, li, \
                                    ksum_addr + (unsigned)((32 * t + li) * 4), kv_addr + (unsigned)(t * 2048 + lane * 16));  \
        kvi[0] = kvi[1]; kvi[1] = kvi[2]; kvi[2] = kvi[3]; kvi[3] = inv__;                                 \
      }                                                                                                    \
    }
    FFX_STAMP(0)
    if (self) { FFX_KV_LOOP(wah, wal) } else { FFX_KV_LOOP(wbh, wbl) }
#undef FFX_KV_LOOP
    FFX_STAMP(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the Ksum / KV stores (asm) have left this wave
    // ============ x side: Q_t -> attention of head pair t -> merge, accumulated over t ============================
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) big[j][r] = 0.f;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      const int p = p0 + 8 + 2 * t;
      FFX_BEGIN(p);
      h16x8 mah[2], mal[2];
      if (live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        FFX_RPANEL(lds + (p % NST) * STAGE, wah, wal, acc);           // Q tile: lane = token, register = feature of heads 2 t, 2 t + 1
        float v[16], den0 = 0.f, den1 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 ws = *reinterpret_cast<const f32x4*>(tl + T_QS + 32 * t + fq + 8 * q);
          const f32x4 ks4 = *reinterpret_cast<const f32x4*>(ksum_tab + 32 * t + fq + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[4 * q + e] * (ws[e] * wa_inv);
            x = x > 0.f ? x + 1.f : __expf(x);
            v[4 * q + e] = x;
            if (q < 2) den0 = fmaf(x, ks4[e], den0); else den1 = fmaf(x, ks4[e], den1);
          }
        }
        den0 += swap32(den0); den1 += swap32(den1);
        const float z0 = S * __builtin_amdgcn_rcpf(den0 + a.attn_eps), z1 = S * __builtin_amdgcn_rcpf(den1 + a.attn_eps);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= r < 8 ? z0 : z1;
        float qm = 0.f, q_inv;                                         // per TOKEN exponent (a column of the product: factors out per lane)
#pragma unroll
        for (int r = 0; r < 16; ++r) qm = fmaxf(qm, v[r]);
        const float q_sc = pow2_lift(fmaxf(qm, swap32(qm)), q_inv);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= q_sc;
        const float undo = kvi[0] * q_inv * (0.25f * src_sc);   // the attention output (a convex combination of V rows) stays in the source window's scale
        h16x8 qh[2], ql[2];
        pack_panel(v, qh, ql);                                         // k-step 0 = head 2 t (d 0 .. 15), k-step 1 = head 2 t + 1
        // message[v][token] = sum_d KV_h[d][v] Q_h[token][d]: block-diagonal A operand (a lane's fragment belongs to one k-step)
        const h16x8 kvh0 = *reinterpret_cast<const h16x8*>(kv_tab + t * 2048), kvl0 = *reinterpret_cast<const h16x8*>(kv_tab + t * 2048 + 1024);
        const h16x8 a0h = li < 16 ? kvh0 : zero8, a0l = li < 16 ? kvl0 : zero8;
        const h16x8 a1h = li < 16 ? zero8 : kvh0, a1l = li < 16 ? zero8 : kvl0;
        f32x16 at;
#pragma unroll
        for (int r = 0; r < 16; ++r) at[r] = 0.f;
        FFX_MF(a0h, ql[0], at); FFX_MF(a0l, qh[0], at); FFX_MF(a0h, qh[0], at);
        FFX_MF(a1h, ql[1], at); FFX_MF(a1l, qh[1], at); FFX_MF(a1h, qh[1], at);
        kvi[0] = kvi[1]; kvi[1] = kvi[2]; kvi[2] = kvi[3];
        float av[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) av[r] = at[r] * undo;
        pack_panel(av, mah, mal);                                      // features 32 t .. 32 t + 31 of the attention output
      }
      FFX_BEGIN(p + 1);
      if (live) FFX_KPANEL(lds + ((p + 1) % NST) * STAGE, mah, mal, big);   // merge: msg += Wm[:, 32 t ..] message_t
    }
    FFX_STAMP(2)
    // ---- message = LayerNorm1(merge output) -> fragments                                                     transformer.py:51-52
    h16x8 mh[8], ml[8];
    if (live) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 ws = *reinterpret_cast<const f32x4*>(tl + T_MS + 32 * j + fq + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) { big[j][4 * q + e] *= ws[e] * (4.f * src_inv); s += big[j][4 * q + e]; }
        }
      s += swap32(s);
      const float mean = s * (1.f / 128.f);
      float m2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = big[j][r] - mean; m2 = fmaf(d, d, m2); }
      m2 += swap32(m2);
      const float rstd = rsqrtf(m2 * (1.f / 128.f) + a.ln_eps);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float y[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 ga = *reinterpret_cast<const f32x4*>(tl + T_G1 + 32 * j + fq + 8 * q);
          const f32x4 be = *reinterpret_cast<const f32x4*>(tl + T_B1 + 32 * j + fq + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) y[4 * q + e] = (big[j][4 * q + e] - mean) * rstd * ga[e] + be[e];
        }
        h16x8 fh[2], fl[2];
        pack_panel(y, fh, fl);
        mh[2 * j] = fh[0]; mh[2 * j + 1] = fh[1]; ml[2 * j] = fl[0]; ml[2 * j + 1] = fl[1];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) big[j][r] = 0.f;
    FFX_STAMP(3)
    // ============ mlp: per 32 hidden features hid = relu(W0[hp] [x, message]) -> out += W2[:, hp] hid ================
#pragma unroll 1
    for (int hp = 0; hp < 8; ++hp) {
      const int p = p0 + 16 + 3 * hp;
      h16x8 hh[2], hl[2];
      FFX_BEGIN(p);
      if (live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        FFX_RPANEL(lds + (p % NST) * STAGE, wah, wal, acc);
      }
      FFX_BEGIN(p + 1);
      if (live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        FFX_RPANEL(lds + ((p + 1) % NST) * STAGE, mh, ml, acc2);      // the message is LayerNorm output: unscaled
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 ws = *reinterpret_cast<const f32x4*>(tl + T_W0S + 32 * hp + fq + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * q + e] = fmaxf(fmaf(acc[4 * q + e], wa_inv, acc2[4 * q + e]) * ws[e], 0.f);   // transformer.py:55 (ReLU)
        }
        pack_panel(v, hh, hl);
      }
      FFX_BEGIN(p + 2);
      if (live) FFX_KPANEL(lds + ((p + 2) % NST) * STAGE, hh, hl, big);
    }
    FFX_STAMP(4)
    // ============ x <- x + LayerNorm2(mlp output), in fp32; every call's rows go to HBM ======================     transformer.py:55-58
    if (live) {
      // the window's fp32 rows (the kernel's input in the self layer, the self layer's output in the cross layer): issued here,
      // consumed after the LayerNorm statistics
      float* orow = ((c & 1) ? a.f1 : a.f0) + (mc * T + min(li, T - 1)) * 128;
      f32x4 xres[16];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) xres[4 * j + q] = *reinterpret_cast<const f32x4*>(orow + 32 * j + fq + 8 * q);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 ws = *reinterpret_cast<const f32x4*>(tl + T_W2S + 32 * j + fq + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) { big[j][4 * q + e] *= ws[e]; s += big[j][4 * q + e]; }
        }
      s += swap32(s);
      const float mean = s * (1.f / 128.f);
      float m2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = big[j][r] - mean; m2 = fmaf(d, d, m2); }
      m2 += swap32(m2);
      const float rstd = rsqrtf(m2 * (1.f / 128.f) + a.ln_eps);
      const bool store = li < T;
      float y[4][16], ym = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 ga = *reinterpret_cast<const f32x4*>(tl + T_G2 + 32 * j + fq + 8 * q);
          const f32x4 be = *reinterpret_cast<const f32x4*>(tl + T_B2 + 32 * j + fq + 8 * q);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = xres[4 * j + q][e] + ((big[j][4 * q + e] - mean) * rstd * ga[e] + be[e]);
            y[j][4 * q + e] = o[e];
            ym = fmaxf(ym, fabsf(o[e]));
          }
          if (store) *reinterpret_cast<f32x4*>(orow + 32 * j + fq + 8 * q) = o;
        }
      }
      ym = half_max(ym); ym = fmaxf(ym, swap32(ym));
      wa_sc = pow2_lift(ym, wa_inv);                      // the updated window gets its own scale
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) y[j][r] *= wa_sc;
        h16x8 fh[2], fl[2];
        pack_panel(y[j], fh, fl);
        wah[2 * j] = fh[0]; wah[2 * j + 1] = fh[1]; wal[2 * j] = fl[0]; wal[2 * j + 1] = fl[1];
      }
    }
    // the updated window becomes "the other one" of the next call: (w0, w1) -> (w1, w0') -> (w0', w1') -> (w1', w0'') -> ..
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const h16x8 th = wah[ks], tlo = wal[ks];
      wah[ks] = wbh[ks]; wal[ks] = wbl[ks]; wbh[ks] = th; wbl[ks] = tlo;
    }
    { const float ts = wa_sc, ti = wa_inv; wa_sc = wb_sc; wa_inv = wb_inv; wb_sc = ts; wb_inv = ti; }
    FFX_STAMP(5)
  }
#ifdef LOFTR_FINE_PROBE
  if (g_fine_probe && lane == 0) {
    long long* o = g_fine_probe + ((long)blockIdx.x * W + wave) * 8;
    for (int i = 0; i < 6; ++i) o[i] = pacc[i];
    o[6] = pstart__; o[7] = wall_clock64();
  }
#endif
#undef FFX_ISSUE
#undef FFX_BEGIN
#undef FFX_RD
#undef FFX_USE
#undef FFX_UNITS
#undef FFX_MF
#undef FFX_ID
#undef FFX_ZERO
#undef FFX_ONE
#undef FFX_RPANEL
#undef FFX_TPANEL
#undef FFX_KB0
#undef FFX_KB1
#undef FFX_KODD
#undef FFX_KPANEL
}
}  // namespace ffx
}  // namespace

#ifdef LOFTR_FINE_PROBE
extern "C" int loftr_fine_probe_buffer(void* buf) {
  long long* b = (long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fine_probe), &b, sizeof(b)) == hipSuccess ? LOFTR_OK : LOFTR_ERR_LAUNCH;
}
#endif

int launch_fine_pair(const FinePairArgs& p, hipStream_t st) {
  if (p.C != 128 || p.T < 1 || p.T > 32 || p.M <= 0) return LOFTR_ERR_UNSUPPORTED;
  ffx::Args a{};
  a.f0 = p.f0; a.f1 = p.f1; a.M = p.M; a.T = p.T; a.attn_eps = p.attn_eps; a.ln_eps = p.ln_eps;
  for (int l = 0; l < 2; ++l) {
    a.wq[l] = p.wq[l]; a.wk[l] = p.wk[l]; a.wv[l] = p.wv[l]; a.wm[l] = p.wm[l]; a.w0[l] = p.w0[l]; a.w2[l] = p.w2[l];
    a.sq[l] = p.sq[l]; a.sk[l] = p.sk[l]; a.sv[l] = p.sv[l]; a.sm[l] = p.sm[l]; a.s0[l] = p.s0[l]; a.s2[l] = p.s2[l];
    a.g1[l] = p.g1[l]; a.b1[l] = p.b1[l]; a.g2[l] = p.g2[l]; a.b2[l] = p.b2[l];
    if (!a.wq[l] || !a.sq[l] || !a.sk[l] || !a.sv[l] || !a.sm[l] || !a.s0[l] || !a.s2[l]) return LOFTR_ERR_UNSUPPORTED;
  }
  TimedLaunch tl(LOFTR_T_FINE_PAIR, st);
  hipLaunchKernelGGL(ffx::fine_pair_kernel, dim3(ceil_div(p.M, ffx::W)), dim3(ffx::W * 64), 0, st, a);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
