// The x side of a coarse LoFTREncoderLayer as ONE kernel with the tokens stationary in registers (round 3).
//   reference: src/loftr/loftr_module/transformer.py:47-58 (q_proj, merge, norm1, mlp, norm2, residual),
//              linear_attention.py:31-36,44-45 (feature map, mask, normaliser)
//
//   Q' = z (.) (elu(x Wq^T) + 1) (.) mask      message = LayerNorm1(Q' P^T)        (P = KV folded into merge, attention.hip)
//   hidden = relu([x, message] W0^T)           out = x + LayerNorm2(hidden W2^T)
//
// Until round 2 these were four launches (q projection, merge + LN, mlp.0, mlp.2 + LN) that moved 14 tensor-sized
// streams through HBM per layer call (Q', message, the 512-wide hidden tensor written and read back).  Activations do not
// fit the LDS next to a weight stream (a 128-token tile of one SP tensor is 128 KB of the 160), but they fit the REGISTER
// file, which is three times the LDS: 32 tokens x 256 features as MFMA B-operand fragments are 128 VGPRs per lane.
//
// Design.  A workgroup is four waves, one per SIMD (up to 512 registers each); a wave owns 32 tokens for the whole layer:
//   * its tokens' activations are the MFMA's B operand (lane = token, 8 consecutive features per lane and k-step, hi and lo
//     halves), the weights are the A operand and stream through a four-stage LDS ring that the four waves share: panels
//     of 32 KB = 8 blocks of (32 rows x 128 B), filled by global_load_lds NST - 1 panels ahead, ONE barrier per panel
//     (48 MFMAs per wave);
//   * with D[feature][token] the lane that owns a token RECEIVES that token's outputs (16 features per 32-feature panel,
//     the other 16 in lane ^ 32): an output panel becomes the next GEMM's B fragments by a (hi, lo) split and eight
//     v_permlane32_swap -- activations never leave the wave, no LDS round trip, no cross-wave exchange, and LayerNorm is a
//     lane-private sum plus one half-wave exchange;
//   * two panel shapes cover all four GEMMs.  "R" = 32 output features x 256 k (Wq rows with x; W0 rows with x, then with
//     the message): one output panel, 16 k-steps.  "K" = 256 output features x 32 k (P with one head of Q'; W2 with one
//     32-wide slice of the hidden layer): all eight output panels advance by two k-steps.  So Q' and the hidden layer are
//     consumed 32 features at a time as they are produced and never exist as a whole (Q': 16 registers, hidden: 16).
//   Live state per lane: x 128 + message accumulators / message fragments 128 + output accumulators 128 + ~70.
// HBM traffic per call: x (SP) and x (fp32, residual) in, out (fp32 + SP) in place: 4 tensor streams instead of 13.
// Weights: 2 MB per workgroup (128 tokens) from the XCD's L2 -- one sequence's workgroups run on one XCD.
#include "linear.h"
#include "attention.h"
#include "coarse_plan.h"

namespace {
namespace efx {
#ifndef EFX_NST
#define EFX_NST 4
#endif
// timing probes (wrong results): 0 = no weight DMA after the prologue / no per-panel barrier / no epilogue arithmetic
#ifndef EFX_PROBE_DMA
#define EFX_PROBE_DMA 1
#endif
#ifndef EFX_PROBE_BARRIER
#define EFX_PROBE_BARRIER 1
#endif
#ifndef EFX_PROBE_EPI
#define EFX_PROBE_EPI 1
#endif
constexpr int W = 4, PT = 32, STAGE = 32 * 1024, NST = EFX_NST, BLK = 4096;     // ring stages: panels are fetched NST - 1 ahead
constexpr int NPANEL = 16 + 48;                         // 8 x (Wq_h, P_h) + 16 x (W0a, W0b, W2_k)
constexpr int DMA_PER_WAVE = 8;                         // global_load_lds per wave and panel: 2 blocks x 4 row octets
// per-feature tables (floats) behind the ring
constexpr int T_WQS = 0, T_KSUM = 256, T_G1 = 512, T_B1 = 768, T_W0S = 1024, T_W2S = 1536, T_G2 = 1792, T_B2 = 2048, T_N = 2304;
constexpr int OFF_TAB = NST * STAGE;
constexpr int LDS_BYTES = OFF_TAB + T_N * 4;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");

struct Args {
  const sp_t* x_sp; const float* x_f32;                 // [nseq][T][256]
  float* out_f32; sp_t* out_sp;                         // may alias x (a wave touches only its own tokens)
  const sp_t* wq; const sp_t* pm; long pm_seq_stride;   // [256][256] SP; [nseq][256][256] SP (row j, column (h, d))
  const sp_t* w0; const sp_t* w2;                       // [512][512], [256][512] SP
  const float *wq_s, *w0_s, *w2_s;                      // inverse power-of-two row scales (gemm.h)
  const float* kv;                                      // [nseq][8][33][32], row 32 of a head = Ksum
  const uint8_t* mask;                                  // [nseq * T] or null
  const float *g1, *b1, *g2, *b2;
  float v_length, attn_eps, p_out_scale, ln_eps;
  int nseq, T, groups;                                  // groups of W token blocks per sequence
  int xsplit, gpc;                                      // fewer sequences than XCDs: a sequence's groups are cut into xsplit chunks of gpc groups, one XCD each
  int skip_padded;                                      // launches only (EncoderXArgs): fully masked 128-token tiles keep their input
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// exchange with lane ^ 32: afterwards lanes 0..31 hold (own a, partner's a), lanes 32..63 (partner's b, own b)
__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void sp_pack4(float x0, float x1, float x2, float x3, uint2& hi, uint2& lo) {
  sp_pack2(x0, x1, hi.x, lo.x);
  sp_pack2(x2, x3, hi.y, lo.y);
}
// 16 outputs of one 32-feature panel (register r of a lane in half-wave g = feature 8 (r >> 2) + 4 g + (r & 3) of the lane's
// token) -> the lane's B-operand fragments of the panel's two k-steps (k-step s, element e = feature 16 s + 8 g + e).
// g = 0 keeps its quads 0 / 2 and receives the partner's (features +4), g = 1 keeps 1 / 3 and receives the partner's.
__device__ __forceinline__ void pack_panel(const float (&v)[16], h16x8 (&fh)[2], h16x8 (&fl)[2]) {
  if (!EFX_PROBE_EPI) {                                 // probe: the accumulator bits as they are
    fh[0] = __builtin_bit_cast(h16x8, f32x4{v[0], v[1], v[2], v[3]}); fl[0] = __builtin_bit_cast(h16x8, f32x4{v[4], v[5], v[6], v[7]});
    fh[1] = __builtin_bit_cast(h16x8, f32x4{v[8], v[9], v[10], v[11]}); fl[1] = __builtin_bit_cast(h16x8, f32x4{v[12], v[13], v[14], v[15]});
    return;
  }
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

// a pointer the caller knows to be wave-uniform, as a scalar-register pair (folds away when it already is one)
__device__ __forceinline__ const sp_t* uniform_ptr(const sp_t* p) {
  const unsigned long long v = (unsigned long long)(size_t)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return reinterpret_cast<const sp_t*>((size_t)(((unsigned long long)hi << 32) | lo));
}

// workgroup index inside a job -> (sequence, group of token blocks): a sequence's groups run back to back on one XCD (weights, P in
// its L2); with fewer sequences than XCDs -- 1 / 2 / 4 at the outdoor configuration's batch sizes -- a sequence is cut into xsplit
// chunks of groups that take one XCD each (pinned one sequence per XCD, 2 sequences used 64 of the 256 CUs)
__device__ __forceinline__ bool job_tile(const Args& a, const int id, int& seq, int& grp) {
  const int xcd = id % NUM_XCD, slot = id / NUM_XCD;
  const int vs = (slot / a.gpc) * NUM_XCD + xcd;        // virtual sequence = (sequence, chunk)
  // (a chunk takes every xsplit-th group, not a contiguous range: the padded tail of a masked sequence -- groups that return at once,
  //  EncoderXArgs.skip_padded -- is then shared out evenly between the sequence's XCDs)
  seq = vs / a.xsplit; grp = (slot % a.gpc) * a.xsplit + vs % a.xsplit;
  return seq < a.nseq && grp < a.groups;
}

// K / V partial of a 128-token tile (kv_tile_main, below): what an X item of the persistent kernel appends for the calls that take its
// OUTPUT tile as their source ("fold": the tile is still in registers as MFMA fragments)
struct KvTile {
  const sp_t* x_sp; const uint8_t* mask; int T;         // the source sequence (already offset to it), its mask or null
  const sp_t* wkv; const float* wkv_s;
  float inv_s;
  float* part; long head_stride;                        // partial of head h at part + h * head_stride
};
struct FoldArgs { int n; KvTile k[2]; };
__device__ __forceinline__ void kv_tile_main(const KvTile& a, char* const lds, const h16x8 (&xh)[16], const h16x8 (&xl)[16],
                                             const float (&mk)[16], const bool live, const int nlive);

// One workgroup of the layer: 128 tokens (group `grp` of sequence `seq`) through the whole x side
template <bool FOLD>
__device__ __forceinline__ void encoder_x_body(const Args& a, const int seq, const int grp, char* const lds, const FoldArgs* fa = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  const int T = a.T;
  // Padding tokens (mask 0) take no part in attention: their scores are filled (coarse_matching.py:115-118), as sources they are multiplied by
  // zero, the fine stage never selects them -- what the layer computes for them (x + LayerNorm2(mlp([x, b1]))) is read by nobody in LoFTR.forward.
  // A caller that says so (skip_padded: loftr_transformer_fwd_padded) gets the 128-token tiles without a single valid token back UNCHANGED
  // instead (in-place calls only); every other token is bit-identical.  Workgroup-uniform.
  if constexpr (!FOLD) {
    if (a.skip_padded && a.mask && a.out_f32 == a.x_f32 && a.out_sp == a.x_sp) {
      const int t = grp * W * PT + (int)threadIdx.x;
      const bool valid = (int)threadIdx.x < W * PT && t < T && a.mask[(long)seq * T + t] != 0;
      if (!__syncthreads_or(valid)) return;
    }
  }
  const int tok = (grp * W + wave) * PT + li;
  const bool live = (grp * W + wave) * PT < T;          // wave-uniform: a wave beyond the sequence only feeds the ring
  const long row = (long)seq * T + min(tok, T - 1);
  float* tab = reinterpret_cast<float*>(lds + OFF_TAB);

  // ---- per-feature tables -> LDS, this lane's token -> registers (ordinary loads / LDS stores: all BEFORE the first DMA)
  for (int f = threadIdx.x; f < 256; f += W * 64) {
    tab[T_WQS + f] = a.wq_s[f];
    tab[T_KSUM + f] = a.kv[((long)seq * 8 + (f >> 5)) * (33 * 32) + 32 * 32 + (f & 31)];
    tab[T_G1 + f] = a.g1[f]; tab[T_B1 + f] = a.b1[f];
    tab[T_W0S + f] = a.w0_s[f]; tab[T_W0S + 256 + f] = a.w0_s[256 + f];
    tab[T_W2S + f] = a.w2_s[f];
    tab[T_G2 + f] = a.g2[f]; tab[T_B2 + f] = a.b2[f];
  }
  h16x8 xh[16], xl[16];                                 // B fragments of x: k-step ks, element e = feature 16 ks + 8 g + e
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.x_sp + row * 256);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int c = (ks >> 1) * 8 + 2 * (ks & 1) + g;
      xh[ks] = __builtin_bit_cast(h16x8, src[c]);
      xl[ks] = __builtin_bit_cast(h16x8, src[c + 4]);
    }
  }
  const float mk = (a.mask && !a.mask[row]) ? 0.f : 1.f;

  // ---- the weight stream.  Panel p of NPANEL lives in ring stage p % 3.
  //   block b (4 KB) of a stage = 32 rows x 128 B (one 32-k group), 16-B chunk c of row r at slot c ^ ((r >> 1) & 7).
  //   R panel: block b = k-group b of the panel's 32 rows;  K panel: block b = rows 32 b .. + 31 of the panel's k-group.
  //   One DMA instruction = 8 rows x 128 B; wave w issues blocks w and w + 4.
  const sp_t* pm = a.pm + (long)seq * a.pm_seq_stride;
  // DMA addressing (round 6): SGPR base + a 32-bit per-lane BYTE offset -- `global_load_lds v_off, s[base]` -- instead of a 64-bit address chain
  // per lane and instruction.  doff[oct]: row octet oct of a block at row pitch 256 dwords; pitch 512 adds (lane >> 3) * 1024 per lane and
  // oct * 8192 on the scalar side.
  unsigned doff[4];
#pragma unroll
  for (int oct = 0; oct < 4; ++oct) doff[oct] = ((unsigned)(oct * 8 + (lane >> 3)) * 256u + ((((unsigned)lane & 7u) ^ ((unsigned)(oct * 4 + (lane >> 4)) & 7u)) << 2)) * 4u;
  const unsigned drl = (unsigned)(lane >> 3) * 1024u;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#define EFX_ISSUE_SRC(p_, base_, pitch_, kt_)                                                              \
  {                                                                                                        \
    char* st__ = lds + ((p_) % NST) * STAGE;                                                               \
    const unsigned w512__ = (pitch_) == 512 ? 1u : 0u;                                                     \
    _Pragma("unroll") for (int bi__ = 0; bi__ < 2; ++bi__) {                                               \
      const int b__ = wave_s + 4 * bi__;                                                                   \
      const char* bb__ = reinterpret_cast<const char*>(uniform_ptr((kt_) ? (base_) + (long)b__ * 32 * (pitch_) : (base_) + b__ * 32)); \
      _Pragma("unroll") for (int oct__ = 0; oct__ < 4; ++oct__) {                                          \
        const char* bo__ = bb__ + (size_t)(w512__ * (unsigned)(oct__ * 8192));                             \
        asm volatile("" : "+s"(bo__));                  /* the scalar part stays a scalar-register pair: saddr form */ \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bo__ + (doff[oct__] + w512__ * drl)),                 \
                                         (lds_ptr_t)(st__ + b__ * BLK + oct__ * 1024), 16, 0, 0);          \
      }                                                                                                    \
    }                                                                                                      \
  }
#define EFX_ISSUE(p_)                                                                                      \
  {                                                                                                        \
    const int p__ = (p_);                                                                                  \
    const sp_t* base__; int pitch__; bool kt__;                                                            \
    if (p__ < 16) {                                                                                        \
      const int h__ = p__ >> 1;                                                                            \
      if (p__ & 1) { base__ = pm + h__ * 32; pitch__ = 256; kt__ = true; }                                 \
      else { base__ = a.wq + (long)h__ * 32 * 256; pitch__ = 256; kt__ = false; }                          \
    } else {                                                                                               \
      const int q__ = p__ - 16, hp__ = q__ / 3, i__ = q__ - 3 * hp__;                                      \
      if (i__ == 2) { base__ = a.w2 + hp__ * 32; pitch__ = 512; kt__ = true; }                             \
      else { base__ = a.w0 + (long)hp__ * 32 * 512 + i__ * 256; pitch__ = 512; kt__ = false; }             \
    }                                                                                                      \
    EFX_ISSUE_SRC(p__, base__, pitch__, kt__)                                                              \
  }
  // panel p has landed once at most the DMAs of the NST - 2 newer panels are outstanding (VMEM operations retire in order); the
  // barrier makes every wave's share visible and proves every wave is past panel p - 1, whose stage panel p + NST - 1 then overwrites
#define EFX_BEGIN_G(p_, NP_, ISSUE_)                                                                       \
  {                                                                                                        \
    if (!EFX_PROBE_DMA) LOFTR_WAITCNT_VM(0);                                                               \
    else if (NST >= 4 && (p_) + 2 < (NP_)) LOFTR_WAITCNT_VM(2 * DMA_PER_WAVE);                             \
    else if ((p_) + 1 < (NP_)) LOFTR_WAITCNT_VM(DMA_PER_WAVE);                                             \
    else LOFTR_WAITCNT_VM(0);                                                                              \
    if (EFX_PROBE_BARRIER) __builtin_amdgcn_s_barrier();                                                   \
    if (EFX_PROBE_DMA && (p_) + NST - 1 < (NP_)) ISSUE_((p_) + NST - 1);                                   \
  }
#define EFX_BEGIN(p_) EFX_BEGIN_G(p_, NPANEL, EFX_ISSUE)
  const int a_off = lds_chunk_off(li, g);               // hi chunk of the even k-step; odd k-step: ^ 32, lo: ^ 64
  // LDS fragment reads run one UNIT (four 16-B fragments, six MFMAs = 192 matrix-pipe cycles) ahead of the MFMAs that
  // consume them -- one wave per SIMD: nobody else hides the ds_read latency.  With LDS-DMA in flight hipcc turns every
  // LDS wait into lgkmcnt(0); EFX_USE (an empty asm that names the current fragments) pins that wait BEFORE the next
  // unit's reads are issued, so it only ever waits for reads issued a whole unit earlier.
#define EFX_RD(st_, blk_, odd_, lo_) (*reinterpret_cast<const h16x8*>((st_) + (blk_) * BLK + (a_off ^ (((odd_) ? 32 : 0) | ((lo_) ? 64 : 0)))))
#define EFX_USE(a_, b_, c_, d_) asm volatile("" :: "v"(a_), "v"(b_), "v"(c_), "v"(d_))
  // R panel: acc0 += W[32 rows][256 k] . B fragments bh / bl; unit u = k-group u = k-steps 2u, 2u + 1.  ONE accumulator chain:
  // a dependent v_mfma_f32_32x32x16_f16 issues at the full rate (tools/micro/mfma_chain.hip: 32.7 cycles per MFMA with 1, 2, 3
  // or 4 chains).  The next unit's four fragment reads are issued in pairs BEHIND the first two MFMAs of this unit -- each pair
  // in the shadow of a 32-cycle MFMA instead of as a burst in front of the unit -- and still >= 4 MFMAs ahead of their wait.
#define EFX_RPANEL_G(st_, bh_, bl_, MF_)                                                                   \
  {                                                                                                        \
    h16x8 eh__ = EFX_RD(st_, 0, 0, 0), el__ = EFX_RD(st_, 0, 0, 1), oh__ = EFX_RD(st_, 0, 1, 0), ol__ = EFX_RD(st_, 0, 1, 1);\
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                        \
      EFX_USE(eh__, el__, oh__, ol__);                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      h16x8 neh__ = eh__, nel__ = el__, noh__ = oh__, nol__ = ol__;                                        \
      acc0 = MF_(eh__, bl_[2 * u], acc0);                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (u + 1 < 8) { neh__ = EFX_RD(st_, u + 1, 0, 0); nel__ = EFX_RD(st_, u + 1, 0, 1); }               \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      acc0 = MF_(el__, bh_[2 * u], acc0);                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (u + 1 < 8) { noh__ = EFX_RD(st_, u + 1, 1, 0); nol__ = EFX_RD(st_, u + 1, 1, 1); }               \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      acc0 = MF_(eh__, bh_[2 * u], acc0);                                                                  \
      acc0 = MF_(oh__, bl_[2 * u + 1], acc0);                                                              \
      acc0 = MF_(ol__, bh_[2 * u + 1], acc0);                                                              \
      acc0 = MF_(oh__, bh_[2 * u + 1], acc0);                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      eh__ = neh__; el__ = nel__; oh__ = noh__; ol__ = nol__;                                              \
    }                                                                                                      \
  }
  // operand order: N = weights (LDS) as A, activations (registers) as B -> D[feature][token];  T = activations as A -> D[token][feature]
#define EFX_MF_N(w_, x_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(w_, x_, c_, 0, 0, 0)
#define EFX_MF_T(w_, x_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(x_, w_, c_, 0, 0, 0)
#define EFX_RPANEL(st_, bh_, bl_) EFX_RPANEL_G(st_, bh_, bl_, EFX_MF_N)
  // K panel: out_[jp] += W[32 jp .. + 31][32 k] . the two k-step fragments fh / fl; unit u = (output panels 2 j2, 2 j2 + 1,
  // k-step s): six MFMAs, the next unit's reads behind the first two
#define EFX_KPANEL(st_, fh_, fl_, out_)                                                                    \
  {                                                                                                        \
    h16x8 ah__ = EFX_RD(st_, 0, 0, 0), al__ = EFX_RD(st_, 0, 0, 1), bh__ = EFX_RD(st_, 1, 0, 0), bl__ = EFX_RD(st_, 1, 0, 1); \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                        \
      const int j2 = u >> 1, s = u & 1;                                                                    \
      const int nj = (u + 1) >> 1, ns = (u + 1) & 1;                                                       \
      EFX_USE(ah__, al__, bh__, bl__);                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      h16x8 nah__ = ah__, nal__ = al__, nbh__ = bh__, nbl__ = bl__;                                        \
      out_[2 * j2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah__, fl_[s], out_[2 * j2], 0, 0, 0);          \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (u + 1 < 8) { nah__ = EFX_RD(st_, 2 * nj, ns, 0); nal__ = EFX_RD(st_, 2 * nj, ns, 1); }           \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      out_[2 * j2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh__, fl_[s], out_[2 * j2 + 1], 0, 0, 0);  \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (u + 1 < 8) { nbh__ = EFX_RD(st_, 2 * nj + 1, ns, 0); nbl__ = EFX_RD(st_, 2 * nj + 1, ns, 1); }   \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      out_[2 * j2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al__, fh_[s], out_[2 * j2], 0, 0, 0);          \
      out_[2 * j2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl__, fh_[s], out_[2 * j2 + 1], 0, 0, 0);  \
      out_[2 * j2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah__, fh_[s], out_[2 * j2], 0, 0, 0);          \
      out_[2 * j2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh__, fh_[s], out_[2 * j2 + 1], 0, 0, 0);  \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      ah__ = nah__; al__ = nal__; bh__ = nbh__; bl__ = nbl__;                                              \
    }                                                                                                      \
  }

  LOFTR_WAITCNT_VM(0);                                  // x fragments, tables: complete before the first DMA
  __syncthreads();
  EFX_ISSUE(0);
  EFX_ISSUE(1);
  if (NST >= 4) EFX_ISSUE(2);

  const int fq = 4 * g;                                 // first feature of register quad 0 inside a panel (quad q: + 8 q)
  f32x16 acc0;
  f32x16 big[8];                                        // message accumulators (pass 1), then output accumulators (pass 2)
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) big[j][r] = 0.f;

  // ================= pass 1: per head h,  Q'_h = f(Wq_h x)  ->  message += P[:, h] Q'_h ==========================
#pragma unroll 1
  for (int h = 0; h < 8; ++h) {
    const int p = 2 * h;
    EFX_BEGIN(p);
    h16x8 qh[2], ql[2];
    if (live) {
      const char* st = lds + (p % NST) * STAGE;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
      EFX_RPANEL(st, xh, xl);
      // elu + 1, mask, attention normaliser z = S / (Q . Ksum + eps): the wave's 32 features ARE head h       linear_attention.py:31-36,44-45
      float v[16];
      float den = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ws = *reinterpret_cast<const f32x4*>(tab + T_WQS + 32 * h + fq + 8 * q);
        const f32x4 ks4 = *reinterpret_cast<const f32x4*>(tab + T_KSUM + 32 * h + fq + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc0[4 * q + e] * ws[e];
          if (EFX_PROBE_EPI) x = x > 0.f ? x + 1.f : __expf(x);
          x *= mk;
          v[4 * q + e] = x;
          den = fmaf(x, ks4[e], den);
        }
      }
      den += swap32(den);
      const float z = a.v_length * __builtin_amdgcn_rcpf(den + a.attn_eps);
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] *= z;
      pack_panel(v, qh, ql);
    }
    EFX_BEGIN(p + 1);
    if (live) {
      const char* st = lds + ((p + 1) % NST) * STAGE;
      EFX_KPANEL(st, qh, ql, big);
    }
  }
  // ---- message = LayerNorm1(p_out_scale * acc) -> B fragments (features 256 .. 511 of the mlp.0 input)     transformer.py:51-52
  h16x8 mh[16], ml[16];
  if (live) {
    // the accumulators are only READ here (they stay where the MFMAs left them); p_out_scale is folded into the statistics
    const float ps = a.p_out_scale;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += big[j][r];
    s += swap32(s);
    const float mean_a = s * (1.f / 256.f);             // mean of the unscaled accumulators
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = big[j][r] - mean_a; m2 = fmaf(d, d, m2); }
    m2 += swap32(m2);
    const float rstd = ps * rsqrtf(ps * ps * m2 * (1.f / 256.f) + a.ln_eps);   // (ps v - ps mean) * rsqrt(var(ps v) + eps)
    const float mean = mean_a;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float y[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(tab + T_G1 + 32 * j + fq + 8 * q);
        const f32x4 be = *reinterpret_cast<const f32x4*>(tab + T_B1 + 32 * j + fq + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[4 * q + e] = (big[j][4 * q + e] - mean) * rstd * ga[e] + be[e];
      }
      h16x8 fh[2], fl[2];
      pack_panel(y, fh, fl);
      mh[2 * j] = fh[0]; mh[2 * j + 1] = fh[1]; ml[2 * j] = fl[0]; ml[2 * j + 1] = fl[1];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) big[j][r] = 0.f;

  // ================= pass 2: per 32 hidden features,  hid = relu(W0[hp] [x, message])  ->  out += W2[:, hp] hid =====
#define EFX_PASS2_HEAD(hp_)                                                                                \
    const int p = 16 + 3 * (hp_);                                                                          \
    h16x8 hh[2], hl[2];                                                                                    \
    EFX_BEGIN(p);                                                                                          \
    if (live) {                                                                                            \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc0[r] = 0.f;                     \
      EFX_RPANEL(lds + (p % NST) * STAGE, xh, xl);                                                                 \
    }                                                                                                      \
    EFX_BEGIN(p + 1);                                                                                      \
    if (live) {                                                                                            \
      EFX_RPANEL(lds + ((p + 1) % NST) * STAGE, mh, ml);                                                                 \
      float v[16];                                                                                         \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
        const f32x4 ws = *reinterpret_cast<const f32x4*>(tab + T_W0S + 32 * (hp_) + fq + 8 * q);           \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                      \
          v[4 * q + e] = fmaxf(acc0[4 * q + e] * ws[e], 0.f);   /* transformer.py:55 (ReLU) */ \
      }                                                                                                    \
      pack_panel(v, hh, hl);                                                                               \
    }                                                                                                      \
    EFX_BEGIN(p + 2);
#pragma unroll 1
  for (int hp = 0; hp < 16; ++hp) {
    EFX_PASS2_HEAD(hp)
    if (live) EFX_KPANEL(lds + ((p + 2) % NST) * STAGE, hh, hl, big);
  }
#undef EFX_PASS2_HEAD

  // ================= out = x + LayerNorm2(mlp.2 output), fp32 and SP                                          transformer.py:55-58
  if (!FOLD && !live) return;
  if (live) {
    // o = acc * w2 row scale, evaluated on the fly in each of the three passes (the accumulators are only read)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ws = *reinterpret_cast<const f32x4*>(tab + T_W2S + 32 * j + fq + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = fmaf(big[j][4 * q + e], ws[e], s);
      }
    s += swap32(s);
    const float mean = s * (1.f / 256.f);
    float m2 = 0.f;
    asm volatile("" ::: "memory");                      // re-read the scale table per pass: kept in registers across the three passes it
#pragma unroll                                          // (128 values) pushes the accumulators out to scratch
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ws = *reinterpret_cast<const f32x4*>(tab + T_W2S + 32 * j + fq + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = big[j][4 * q + e] * ws[e] - mean; m2 = fmaf(d, d, m2); }
      }
    m2 += swap32(m2);
    const float rstd = rsqrtf(m2 * (1.f / 256.f) + a.ln_eps);
    asm volatile("" ::: "memory");
    // (both lanes of a token -- lane, lane ^ 32 -- take the same branch, so the exchanges inside stay paired)
    // FOLD: every lane computes (lanes beyond the sequence on the clamped row: finite values the K / V tail masks out), stores stay guarded
    const bool in_seq = tok < T;
    if (FOLD || in_seq) {
      const float* xr = a.x_f32 + row * 256;
      float* of = a.out_f32 + row * 256;
      sp_t* os = a.out_sp + row * 256;
      f32x4 xn[4];                                      // residual rows one panel ahead of their use (their loads are older than
#pragma unroll                                          // the previous panel's stores: the wait for them is a counted vmcnt)
      for (int q = 0; q < 4; ++q) xn[q] = *reinterpret_cast<const f32x4*>(xr + fq + 8 * q);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 xc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xc[q] = xn[q];
        if (j + 1 < 8) {
#pragma unroll
          for (int q = 0; q < 4; ++q) xn[q] = *reinterpret_cast<const f32x4*>(xr + 32 * (j + 1) + fq + 8 * q);
        }
        float y[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = 32 * j + fq + 8 * q;
          const f32x4 ga = *reinterpret_cast<const f32x4*>(tab + T_G2 + f);
          const f32x4 be = *reinterpret_cast<const f32x4*>(tab + T_B2 + f);
          const f32x4 ws = *reinterpret_cast<const f32x4*>(tab + T_W2S + f);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = xc[q][e] + ((big[j][4 * q + e] * ws[e] - mean) * rstd * ga[e] + be[e]);
            y[4 * q + e] = o[e];
          }
          if (!FOLD || in_seq) *reinterpret_cast<f32x4*>(of + f) = o;
        }
        if (!FOLD && !a.out_sp) continue;                 // (loftr_encoder_layer_fwd: fp32 result only)
        h16x8 fh[2], fl[2];
        pack_panel(y, fh, fl);
        if (FOLD) { xh[2 * j] = fh[0]; xh[2 * j + 1] = fh[1]; xl[2 * j] = fl[0]; xl[2 * j + 1] = fl[1]; }   // the output tile as the next calls' source fragments
        if (!FOLD || in_seq) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            *reinterpret_cast<u32x4*>(os + j * 32 + (2 * s2 + g) * 4) = __builtin_bit_cast(u32x4, fh[s2]);
            *reinterpret_cast<u32x4*>(os + j * 32 + 16 + (2 * s2 + g) * 4) = __builtin_bit_cast(u32x4, fl[s2]);
          }
        }
      }
    }
  }
  // ================= fold: K / V partials of the calls whose source is this output tile (persistent kernel only) =================
  if constexpr (FOLD) {
    if (fa->n > 0) {
      const int tok0 = (grp * W + wave) * PT;
      const int nlive = min(W, (T - grp * W * PT + PT - 1) / PT);
      float mk[16];                                       // register r <-> token tok0 + (r & 3) + 8 (r >> 2) + 4 g   (D[token][feature])
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = tok0 + (r & 3) + 8 * (r >> 2) + 4 * g;
        mk[r] = (t < T && (!a.mask || a.mask[(long)seq * T + min(t, T - 1)])) ? 1.f : 0.f;
      }
#pragma unroll 1
      for (int s = 0; s < fa->n; ++s) kv_tile_main(fa->k[s], lds, xh, xl, mk, live, nlive);
    }
  }
}

__global__ __launch_bounds__(W * 64, 1) void encoder_x_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  int seq, grp;
  if (job_tile(a, blockIdx.x, seq, grp)) encoder_x_body<false>(a, seq, grp, lds);
}

// TWO jobs in one launch.  A call's time is set by whole rounds of 256 workgroups (one per CU: tools/gpu/r4_enc_sweep.sh -- 256
// workgroups 116 us, 300: 201 us, 512: 242 us), and the cross-attention calls of the batch-8 configuration are 304: the 208 slots
// their second round leaves idle take workgroups of the NEXT self-attention call on the other image (it depends on the same
// predecessor).  Workgroups [0, n0) are job 0's [off0, off0 + n0), the rest job 1's from off1 on; offsets are multiples of the XCD
// count, so that a workgroup's XCD (blockIdx % 8) is the one its index inside the job names.  Results do not depend on the split.
struct Args2 { Args j[2]; int n0, off0, off1; };
__global__ __launch_bounds__(W * 64, 1) void encoder_x2_kernel(Args2 m) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  const bool second = (int)blockIdx.x >= m.n0;          // workgroup-uniform
  const Args* a = second ? &m.j[1] : &m.j[0];           // (one copy of the body: its arguments come from a uniform kernarg offset)
  int seq, grp;
  if (job_tile(*a, second ? (int)blockIdx.x - m.n0 + m.off1 : (int)blockIdx.x + m.off0, seq, grp)) encoder_x_body<false>(*a, seq, grp, lds);
}

// ------------------------------------------------------------------------------------------------------------------
// The PERSISTENT coarse transformer (round 6): one launch per LocalFeatureTransformer.forward (transformer.py:80-101).
//
// Until round 5 a transformer call was 12 encoder_x launches + 15 proj_kv + 15 kv_finalize launches, every one a grid-wide barrier: a
// launch's time is whole rounds of 256 workgroups (one per CU, ~110 us each), and at the BASELINE size (8 pairs x 4800 tokens = 300 /
// 600 workgroups per call) the x side ran 25 rounds for 19 rounds of work (profiles/r05_encoder_rounds.txt).  But the dependency
// `feat1 attends to the UPDATED feat0` (transformer.py:96-97) is PER PAIR and, below that, per 128-token tile:
//     K(c, p, t)   k / v projection of tile t of pair p's source + its partial K^T V      needs the X item that last wrote that tile
//     F(c, p, h)   fixed-order sum of the partials of head h, KV folded into the merge weight (P)      needs all K(c, p, .)
//     X(c, p, g)   the whole x side of the layer for 128 tokens (encoder_x_body)          needs all F(c, p, .), the X item that last
//                  wrote its tile, and (write-after-read) the F items of every other call that reads the tile's OLD content
// so here 256 resident workgroups pull these items from ONE queue whose order the host planned (coarse_plan in transformer.hip: a
// list schedule of the dependency graph, critical path first) and wait on per-item / per-(call, pair) counters in HBM.  The queue
// order is a topological order and items are popped in order, so an item's dependencies are always held by running workgroups: no
// deadlock whatever the residency.  Visibility between workgroups: agent-scope release (buffer_wbl2 sc1) by one lane after the
// workgroup's stores have drained, relaxed counter add; consumer: relaxed poll by one lane, agent-scope acquire (buffer_inv sc1),
// barrier, plain loads -- placement independent (guide: "inter-workgroup communication").
// Results do not depend on the queue order: every item computes from the same inputs in the same internal order
// (tests: dependency order vs call order bit-identical).
struct PctCall { int layer, x_img, s_img, pad, fold[2], pad2[2]; };   // fold: calls whose K / V partials this call's X items compute in their tails (-1: none)
struct PctArgs {
  float* f32[2]; sp_t* sp[2]; const uint8_t* mask[2]; int T[2];
  int N, n_calls, n_items, splits;                      // splits: row tiles per sequence the partial buffers are strided by (max over the images)
  const PctItem* items; unsigned* cnt;                  // cnt[0] = queue head, cnt[1] = error word, cnt[2] = plan signature check, cnt[PCT_CNT0 ..] dependency counters
  float* part; float* kv; sp_t* pm;                     // per (call, pair): [8][splits][33][32], [8][33][32], [256][256] SP
  unsigned long long* trace;                            // null, or per item {popped, ready, done, workgroup} (wall_clock64: 100 MHz)
  unsigned* status;                                     // null, or where the error word is copied to
  unsigned signature; int skip;
  int quota;                                            // 0: a workgroup pulls items until the queue is empty; n > 0: it leaves after n items (its CU goes back to the dispatcher: another stream's workgroups get a turn)
  PctLayerPtrs layer[PCT_MAX_LAYERS]; PctCall call[PCT_MAX_CALLS];
};
constexpr int OFF_RED = OFF_TAB + T_N * 4, RED_FLOATS = 33 * 32, OFF_CTRL = OFF_RED + W * RED_FLOATS * 4, LDS_BYTES_P = OFF_CTRL + 64;
static_assert(LDS_BYTES_P <= 160 * 1024, "one workgroup per CU");

// ---- K item: k / v projection of 128 source tokens + their partial K^T V and K sum (replaces proj_kv_kernel, linear.hip) ------------
//   linear_attention.py:31-32 (elu + 1), :37-42 (masks, values / v_length), :43-44 (einsum nshd,nshv->nhdv; K.sum)
// Same frame as encoder_x_body: a wave owns 32 tokens, their SP fragments stay in registers, the interleaved [K_h | V_h] weight rows
// (transformer.hip: stage_layer) stream through the LDS ring as 16 R panels.  Here the activations are the MFMA's A operand, so
// D[token][feature]: lane = feature d (resp. v) of the head, registers = 16 of the wave's tokens -- exactly the operands of the fp32
// MFMA that contracts over the tokens (KV_h[d][v] += K[t][d] V[t][v], 16 v_mfma_f32_32x32x2_f32 per head as in proj_kv_kernel).
// The four waves' results are summed through LDS in a fixed order: one partial per (tile, head), [33][32] (row 32 = K sum).
// kv_tile_body: a K item (loads the tile's fragments and token mask, then kv_tile_main); kv_tile_main: the 16 weight panels, feature map,
// K^T V on the fp32 matrix cores and the cross-wave reduction -- also the tail of an X item whose output tile is the source ("fold").
__device__ __forceinline__ void kv_tile_body(const KvTile& a, const int tile, char* const lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  const int T = a.T;
  const int tok0 = (tile * W + wave) * PT;
  const bool live = tok0 < T;                           // wave-uniform
  const int nlive = min(W, (T - tile * W * PT + PT - 1) / PT);
  h16x8 xh[16], xl[16];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.x_sp + (long)min(tok0 + li, T - 1) * 256);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int c = (ks >> 1) * 8 + 2 * (ks & 1) + g;
      xh[ks] = __builtin_bit_cast(h16x8, src[c]);
      xl[ks] = __builtin_bit_cast(h16x8, src[c + 4]);
    }
  }
  float mk[16];                                         // register r <-> token tok0 + (r & 3) + 8 (r >> 2) + 4 g
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = tok0 + (r & 3) + 8 * (r >> 2) + 4 * g;
    mk[r] = (t < T && (!a.mask || a.mask[min(t, T - 1)])) ? 1.f : 0.f;
  }
  kv_tile_main(a, lds, xh, xl, mk, live, nlive);
}

__device__ __forceinline__ void kv_tile_main(const KvTile& a, char* const lds, const h16x8 (&xh)[16], const h16x8 (&xl)[16],
                                             const float (&mk)[16], const bool live, const int nlive) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  float* tab = reinterpret_cast<float*>(lds + OFF_TAB);
  // (entered with no DMA in flight and every wave past its last use of the ring and of the scale tables: the barrier below)
  LOFTR_WAITCNT_VM(0);
  __syncthreads();
  for (int f = threadIdx.x; f < 512; f += W * 64) tab[T_W0S + f] = a.wkv_s[f];
  unsigned doff[4];                                     // see encoder_x_body
#pragma unroll
  for (int oct = 0; oct < 4; ++oct) doff[oct] = ((unsigned)(oct * 8 + (lane >> 3)) * 256u + ((((unsigned)lane & 7u) ^ ((unsigned)(oct * 4 + (lane >> 4)) & 7u)) << 2)) * 4u;
  [[maybe_unused]] const unsigned drl = 0u;             // (the K / V weights have one pitch)
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int a_off = lds_chunk_off(li, g);
  const sp_t* const wkv = a.wkv;
#define KVX_ISSUE(p_) EFX_ISSUE_SRC((p_), wkv + (long)(p_) * 32 * 256, 256, false)
  // reduction buffer: wave w's [33][32] block; element (d, v) of the fp32 MFMA's D: row d = (r & 3) + 8 (r >> 2) + 4 g, column v = li
  const unsigned red_w = (unsigned)(size_t)(lds_ptr_t)(lds + OFF_RED) + (unsigned)((wave * RED_FLOATS + 4 * g * 32 + li) * 4);
  const char* red_r = lds + OFF_RED;
  // the sum of head h's four blocks (live waves only, fixed order) -> the tile's partial; done by wave h & 3 one barrier after the blocks were written
#define KVX_REDUCE(h_)                                                                                     \
  if (wave == ((h_) & 3)) {                                                                                \
    float* out__ = a.part + (long)(h_) * a.head_stride;                                                    \
    _Pragma("unroll") for (int u = 0; u < 5; ++u) {                                                        \
      const int v__ = lane + u * 64;                                                                       \
      if (v__ < RED_FLOATS / 4) {                                                                          \
        f32x4 s__ = *reinterpret_cast<const f32x4*>(red_r + v__ * 16);                                     \
        if (nlive > 1) s__ += *reinterpret_cast<const f32x4*>(red_r + 1 * RED_FLOATS * 4 + v__ * 16);     \
        if (nlive > 2) s__ += *reinterpret_cast<const f32x4*>(red_r + 2 * RED_FLOATS * 4 + v__ * 16);     \
        if (nlive > 3) s__ += *reinterpret_cast<const f32x4*>(red_r + 3 * RED_FLOATS * 4 + v__ * 16);     \
        reinterpret_cast<f32x4*>(out__)[v__] = s__;                                                        \
      }                                                                                                    \
    }                                                                                                      \
  }
  LOFTR_WAITCNT_VM(0);                                  // x fragments, tables: complete before the first DMA
  __syncthreads();
  KVX_ISSUE(0);
  KVX_ISSUE(1);
  if (NST >= 4) KVX_ISSUE(2);
  f32x16 acc0, acck;
#pragma unroll 1
  for (int h = 0; h < 8; ++h) {
    const int p = 2 * h;
    EFX_BEGIN_G(p, 16, KVX_ISSUE);
    if (h > 0) KVX_REDUCE(h - 1)
    if (live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
      EFX_RPANEL_G(lds + (p % NST) * STAGE, xh, xl, EFX_MF_T);
      acck = acc0;
    }
    EFX_BEGIN_G(p + 1, 16, KVX_ISSUE);
    if (live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
      EFX_RPANEL_G(lds + ((p + 1) % NST) * STAGE, xh, xl, EFX_MF_T);
      const float wk = tab[T_W0S + 64 * h + li], wv = tab[T_W0S + 64 * h + 32 + li];
      float ksum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float k = acck[r] * wk;
        acck[r] = (k > 0.f ? k + 1.f : __expf(k)) * mk[r];            // elu(k) + 1, masked
        acc0[r] = acc0[r] * (wv * (mk[r] * a.inv_s));                  // values * mask / v_length
        ksum += acck[r];
      }
      f32x16 kv;
#pragma unroll
      for (int r = 0; r < 16; ++r) kv[r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) kv = __builtin_amdgcn_mfma_f32_32x32x2f32(acck[r], acc0[r], kv, 0, 0, 0);
      ksum += swap32(ksum);
      // (asm LDS stores: a compiler-visible one would drain the weight DMA in flight)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(red_w), "v"(kv[r]), "i"(((r & 3) + 8 * (r >> 2)) * 128) : "memory");
      if (g == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(red_w), "v"(ksum), "i"(32 * 128) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __builtin_amdgcn_s_barrier();
  KVX_REDUCE(7)
#undef KVX_ISSUE
#undef KVX_REDUCE
}

// ---- F item: head h of a (call, pair): sum of the row-tile partials in a fixed order, KV folded into the merge projection ----------
// The arithmetic of kv_finalize_kernel (attention.hip), thread for thread: P[j][h * 32 + d] = sum_v KV[d][v] Wm[j][h * 32 + v], SP.
__device__ __forceinline__ void kv_final_body(const float* part, const int splits, float* kvout, const float* wm, const int h,
                                              sp_t* pm, char* const lds) {
  float* skv = reinterpret_cast<float*>(lds);
  float* sgrp = reinterpret_cast<float*>(lds + 8192);   // [4][33 * 32]
  {
    constexpr int NV = 33 * 32 / 4;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 a[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = w; k < splits; k += 4) {
      const f32x4* pk = reinterpret_cast<const f32x4*>(part + (long)k * (33 * 32));
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int v = lane + u * 64;
        if (v < NV) a[u] += pk[v];
      }
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int v = lane + u * 64;
      if (v < NV) reinterpret_cast<f32x4*>(sgrp + w * (33 * 32))[v] = a[u];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 33 * 32; e += 256) {
    const float s = ((sgrp[e] + sgrp[33 * 32 + e]) + sgrp[2 * 33 * 32 + e]) + sgrp[3 * 33 * 32 + e];
    kvout[e] = s;
    skv[e] = s;
  }
  __syncthreads();
  const int q = threadIdx.x & 3;
#pragma unroll 1
  for (int z = 0; z < 4; ++z) {
    const int j = z * 64 + (threadIdx.x >> 2);
    f32x4 w4[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w4[i] = reinterpret_cast<const f32x4*>(wm + (long)j * 256 + h * 32)[i];
    float r[8];
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      const float* kvrow = &skv[(q * 8 + dd) * 32];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(kvrow + i * 4);
        s += k4.x * w4[i].x + k4.y * w4[i].y + k4.z * w4[i].z + k4.w * w4[i].w;
      }
      r[dd] = s * ATTN_P_SCALE;
    }
    u32x4 hi, lo;
    sp_pack8(r, hi, lo);
    sp_t* dst = pm + (long)j * 256 + h * 32 + q * 4;
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 16) = lo;
  }
}

typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned pct_load(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(W * 64, 1) void coarse_persistent_kernel(PctArgs P) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES_P];
  int* const ctrl = reinterpret_cast<int*>(lds + OFF_CTRL);
  if (P.items[-1].what != P.signature || P.items[-1].signal != (unsigned)P.n_items) {     // a plan built for another shape: refuse (error word 2)
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_or(P.cnt + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (P.status) __hip_atomic_fetch_or(P.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // Everything between the barriers that only one lane needs to do runs WAVE-UNIFORMLY in wave 0 (scalar branches, values broadcast
  // with readfirstlane; single-lane regions hold one atomic each and no barrier).  A plain `if (threadIdx.x == 0)` around the
  // pop / poll and the publish makes hipcc rotate those two blocks into an outer loop run by lane 0 alone while lanes 1..63 of its
  // wave go round the barriers without it: the workgroup then re-reads the same queue entry for ever (round 6, first build).
  const int lane0 = (threadIdx.x & 63) == 0;
  const bool wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;
  for (int mine = 0; P.quota <= 0 || mine < P.quota; ++mine) {
    unsigned long long t_pop = 0, t_rdy = 0;
    if (wave0) {
      unsigned itv = 0;
      if (lane0) itv = __hip_atomic_fetch_add(P.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned it = __builtin_amdgcn_readfirstlane(itv);
      int go = it < (unsigned)P.n_items ? 1 : 0;
      if (P.trace) t_pop = wall_clock64();
      if (go) {
        // wait for the item's dependencies: relaxed polls of monotonic counters (one address per poll), then ONE agent-scope acquire
        const PctItem* pi = P.items + it;
        bool any = false;
#pragma unroll 1
        for (int d = 0; d < 4 && go; ++d) {
          const uint32_t dep = __builtin_amdgcn_readfirstlane(pi->dep[d]);
          if (dep == PCT_NODEP) continue;
          any = true;
          unsigned* c = P.cnt + (dep & 0xfffffu);
          const unsigned target = dep >> 20;
          unsigned spins = 0;
          while (__builtin_amdgcn_readfirstlane(pct_load(c)) < target) {
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 1023u) == 0 && (__builtin_amdgcn_readfirstlane(pct_load(P.cnt + 1)) != 0 || spins > (1u << 20))) {   // someone gave up, or ~1 s without progress: give up too
              if (lane0) {
                __hip_atomic_fetch_or(P.cnt + 1, 1u + (it << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (P.status) __hip_atomic_fetch_or(P.status, 1u + (it << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              go = 0;
              break;
            }
          }
        }
        if (any) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (P.trace) t_rdy = wall_clock64();
      }
      if (lane0) ctrl[0] = go ? (int)it : -1;
    }
    __syncthreads();
    const int it = __builtin_amdgcn_readfirstlane(ctrl[0]);
    if (it < 0) break;
    const PctItem* pi = P.items + it;
    const unsigned what = __builtin_amdgcn_readfirstlane(pi->what), signal = __builtin_amdgcn_readfirstlane(pi->signal);
    const unsigned sig2a = __builtin_amdgcn_readfirstlane(pi->sig2[0]), sig2b = __builtin_amdgcn_readfirstlane(pi->sig2[1]);
    const int type = what & 15, c = (what >> 4) & 255, pair = (what >> 12) & 255, idx = what >> 20;
    const PctCall call = P.call[c];
    const PctLayerPtrs& lw = P.layer[call.layer];
    const int Tx = P.T[call.x_img], Ts = P.T[call.s_img];
    const long cp = (long)c * P.N + pair;
    if ((P.skip >> type) & 1) {
    } else if (type == PCT_X) {
      Args a;
      a.x_sp = P.sp[call.x_img]; a.x_f32 = P.f32[call.x_img]; a.out_f32 = P.f32[call.x_img]; a.out_sp = P.sp[call.x_img];
      a.wq = lw.wq; a.pm = P.pm + (long)c * P.N * 65536; a.pm_seq_stride = 65536; a.w0 = lw.w0; a.w2 = lw.w2;
      a.wq_s = lw.wq_s; a.w0_s = lw.w0_s; a.w2_s = lw.w2_s;
      a.kv = P.kv + (long)c * P.N * (8 * 33 * 32);
      a.mask = P.mask[call.x_img];
      a.g1 = lw.g1; a.b1 = lw.b1; a.g2 = lw.g2; a.b2 = lw.b2;
      a.v_length = (float)Ts; a.attn_eps = 1e-6f; a.p_out_scale = 1.f / ATTN_P_SCALE; a.ln_eps = 1e-5f;
      a.nseq = P.N; a.T = Tx; a.groups = 0; a.xsplit = 1; a.gpc = 0;
      FoldArgs fa;
      fa.n = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cf = call.fold[i];
        if (cf < 0) continue;
        const PctLayerPtrs& lf = P.layer[P.call[cf].layer];
        KvTile& k = fa.k[fa.n++];
        k.x_sp = nullptr; k.mask = nullptr; k.T = Tx;                   // (fragments and token mask come from the X item)
        k.wkv = lf.wkv; k.wkv_s = lf.wkv_s; k.inv_s = 1.f / (float)Tx;
        k.part = P.part + (((long)cf * P.N + pair) * 8 * P.splits + idx) * (33 * 32); k.head_stride = (long)P.splits * (33 * 32);
      }
      encoder_x_body<true>(a, pair, idx, lds, &fa);
    } else if (type == PCT_K) {
      KvTile k;
      k.x_sp = P.sp[call.s_img] + (long)pair * Ts * 256;
      k.mask = P.mask[call.s_img] ? P.mask[call.s_img] + (long)pair * Ts : nullptr;
      k.T = Ts; k.wkv = lw.wkv; k.wkv_s = lw.wkv_s; k.inv_s = 1.f / (float)Ts;
      k.part = P.part + (cp * 8 * P.splits + idx) * (33 * 32); k.head_stride = (long)P.splits * (33 * 32);
      kv_tile_body(k, idx, lds);
    } else {
      kv_final_body(P.part + (cp * 8 + idx) * P.splits * (33 * 32), ceil_div(Ts, W * PT), P.kv + (cp * 8 + idx) * (33 * 32), lw.merge_f32, idx,
                    P.pm + cp * 65536, lds);
    }
    // publish: every wave's stores drained, then ONE lane releases at agent scope (L2 write-back) and bumps the item's counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the compiler may drop the wait behind buffer_wbl2: guide, pitfall 12)
      if (lane0) {
        __hip_atomic_fetch_add(P.cnt + signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sig2a != PCT_NODEP) __hip_atomic_fetch_add(P.cnt + sig2a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sig2b != PCT_NODEP) __hip_atomic_fetch_add(P.cnt + sig2b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (P.trace) {
          unsigned long long* tr = P.trace + (long)it * 4;
          tr[0] = t_pop; tr[1] = t_rdy; tr[2] = wall_clock64(); tr[3] = blockIdx.x;
        }
      }
    }
  }
}
#undef EFX_RPANEL
#undef EFX_RPANEL_G
#undef EFX_MF_N
#undef EFX_MF_T
#undef EFX_KPANEL
#undef EFX_RD
#undef EFX_USE
#undef EFX_ISSUE
#undef EFX_ISSUE_SRC
#undef EFX_BEGIN
#undef EFX_BEGIN_G
}  // namespace efx
}  // namespace

// the job's kernel arguments and its number of workgroups (0: not a shape the fused kernel takes)
static int make_job(const EncoderXArgs& p, efx::Args& a) {
  if (p.C != 256 || p.nseq <= 0 || p.T <= 0 || !p.wq_s || !p.w0_s || !p.w2_s || !p.kv) return 0;
  a = efx::Args{};
  a.x_sp = p.x_sp; a.x_f32 = p.x_f32; a.out_f32 = p.out_f32; a.out_sp = p.out_sp;
  a.wq = p.wq; a.pm = p.pm; a.pm_seq_stride = p.pm_seq_stride; a.w0 = p.w0; a.w2 = p.w2;
  a.wq_s = p.wq_s; a.w0_s = p.w0_s; a.w2_s = p.w2_s; a.kv = p.kv; a.mask = p.mask;
  a.g1 = p.g1; a.b1 = p.b1; a.g2 = p.g2; a.b2 = p.b2;
  a.v_length = p.v_length; a.attn_eps = p.attn_eps; a.p_out_scale = p.p_out_scale; a.ln_eps = p.ln_eps;
  a.skip_padded = p.skip_padded;
  a.nseq = p.nseq; a.T = p.T; a.groups = ceil_div(ceil_div(p.T, efx::PT), efx::W);
  a.xsplit = p.nseq < NUM_XCD ? NUM_XCD / p.nseq : 1;
  if (a.xsplit > a.groups) a.xsplit = a.groups;
  a.gpc = ceil_div(a.groups, a.xsplit);
  return NUM_XCD * ceil_div(p.nseq * a.xsplit, NUM_XCD) * a.gpc;
}

int encoder_x_workgroups(const EncoderXArgs& p) {
  efx::Args a;
  return make_job(p, a);
}

// workgroups [off0, off0 + n0) of job p0 followed by [off1, off1 + n1) of job p1 in ONE launch (encoder_x2_kernel); n1 == 0: p0 only
int launch_encoder_x2(const EncoderXArgs& p0, int off0, int n0, const EncoderXArgs& p1, int off1, int n1, hipStream_t st) {
  efx::Args2 m;
  const int g0 = make_job(p0, m.j[0]);
  const int g1 = n1 > 0 ? make_job(p1, m.j[1]) : 0;
  if (g0 == 0 || (n1 > 0 && g1 == 0)) return LOFTR_ERR_UNSUPPORTED;
  if (off0 < 0 || n0 <= 0 || off0 + n0 > g0 || off0 % NUM_XCD || n0 % NUM_XCD || n1 < 0) return LOFTR_ERR_BAD_ARG;
  if (n1 > 0 && (off1 < 0 || off1 + n1 > g1 || off1 % NUM_XCD)) return LOFTR_ERR_BAD_ARG;
  if (n1 == 0) m.j[1] = m.j[0];
  m.n0 = n0; m.off0 = off0; m.off1 = off1;
  TimedLaunch tl(LOFTR_T_ENCODER_X, st);
  hipLaunchKernelGGL(efx::encoder_x2_kernel, dim3(n0 + n1), dim3(efx::W * 64), 0, st, m);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

int launch_encoder_x(const EncoderXArgs& p, hipStream_t st) {
  efx::Args a;
  const int grid = make_job(p, a);
  if (grid == 0) return LOFTR_ERR_UNSUPPORTED;
  TimedLaunch tl(LOFTR_T_ENCODER_X, st);
  hipLaunchKernelGGL(efx::encoder_x_kernel, dim3(grid), dim3(efx::W * 64), 0, st, a);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ---- the persistent coarse transformer: ONE launch per LocalFeatureTransformer.forward (see coarse_persistent_kernel) ----------
int launch_coarse_persistent(const PctLaunch& p, hipStream_t st) {
  const PctShape& s = p.shape;
  if (!s.ok() || !p.plan) return LOFTR_ERR_UNSUPPORTED;
  if (p.plan_bytes < pct_plan_bytes(s) || p.ws_bytes < pct_ws_bytes(s)) return LOFTR_ERR_WORKSPACE;
  const size_t cp = (size_t)s.n_calls() * s.N;
  WsAlloc wa(p.ws, p.ws_bytes);
  efx::PctArgs a{};
  a.cnt = wa.take<unsigned>(s.n_counters());
  a.part = wa.take<float>(cp * 8 * s.gmax() * 33 * 32);
  a.kv = wa.take<float>(cp * 8 * 33 * 32);
  a.pm = wa.take<sp_t>(cp * 65536);
  if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
  for (int i = 0; i < 2; ++i) { a.f32[i] = p.f32[i]; a.sp[i] = p.sp[i]; a.mask[i] = p.mask[i]; a.T[i] = s.T[i]; }
  a.N = s.N; a.n_calls = s.n_calls(); a.n_items = (int)s.n_items(); a.splits = s.gmax();
  a.items = reinterpret_cast<const PctItem*>(p.plan) + 1;
  a.trace = p.trace; a.status = p.status; a.signature = p.plan_signature;
  for (int l = 0; l < s.n_layers; ++l) a.layer[l] = p.layer[l];
  for (int c = 0; c < s.n_calls(); ++c) { s.call(c, a.call[c].layer, a.call[c].x_img, a.call[c].s_img); s.folds(c, a.call[c].fold); }
  if (hipMemsetAsync(a.cnt, 0, s.n_counters() * 4, st) != hipSuccess) return LOFTR_ERR_LAUNCH;
  TimedLaunch tl(LOFTR_T_ENCODER_X, st);
  a.skip = loftr_debug_value(LOFTR_DBG_PCT_SKIP);
  const int cap = loftr_debug_value(LOFTR_DBG_PCT_GRID) > 0 ? loftr_debug_value(LOFTR_DBG_PCT_GRID) : 256;
  int grid = a.n_items < cap ? a.n_items : cap;            // one workgroup per CU; fewer resident ones only slow the queue down (in-order pops: no deadlock)
  // "pct_quota" n > 0: workgroups leave after n items and the grid holds enough of them for the whole queue (an item popped is held by a
  // RESIDENT workgroup, so the in-order argument against deadlock is unchanged); between two workgroups a CU is up for grabs
  a.quota = loftr_debug_value(LOFTR_DBG_PCT_QUOTA);
  if (a.quota > 0) grid = ceil_div(a.n_items, a.quota) + 8;
  hipLaunchKernelGGL(efx::coarse_persistent_kernel, dim3(grid), dim3(efx::W * 64), 0, st, a);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
