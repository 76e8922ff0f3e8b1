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
#ifndef EFX_PIPE
#define EFX_PIPE 0                // 1, 2: inline-asm fragment reads with counted lgkmcnt waits, that many units ahead (measured: 2.88 / 2.97 / 3.27 ms per transformer call for 0 / 1 / 2 -- the third buffer spills inside the loops)
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

// One workgroup of the layer: `id` is its index inside the job `a` (= blockIdx.x when the launch holds one job)
__device__ __forceinline__ void encoder_x_body(const Args& a, const int id, char* const lds) {
  // ---- workgroup -> (sequence, group of token blocks): a sequence's groups run back to back on one XCD (weights, P in its L2)
  // (with fewer sequences than XCDs -- 1 / 2 / 4 at the outdoor configuration's batch sizes -- a sequence is cut into xsplit chunks
  //  of groups that take one XCD each: pinned one sequence per XCD, 2 sequences used 64 of the 256 CUs)
  const int xcd = id % NUM_XCD, slot = id / NUM_XCD;
  const int vs = (slot / a.gpc) * NUM_XCD + xcd;        // virtual sequence = (sequence, chunk)
  const int seq = vs / a.xsplit, grp = (vs % a.xsplit) * a.gpc + slot % a.gpc;
  if (seq >= a.nseq || grp >= a.groups) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  const int T = a.T;
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
  int dro[4];                                           // row-octet part of the source offset (rows; dwords of the chunk)
  int dch[4];
#pragma unroll
  for (int oct = 0; oct < 4; ++oct) {
    dro[oct] = oct * 8 + (lane >> 3);
    dch[oct] = ((lane & 7) ^ ((oct * 4 + (lane >> 4)) & 7)) << 2;
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
    char* st__ = lds + (p__ % NST) * STAGE;                                                                \
    _Pragma("unroll") for (int bi__ = 0; bi__ < 2; ++bi__) {                                               \
      const int b__ = wave + 4 * bi__;                                                                     \
      const sp_t* bb__ = kt__ ? base__ + (long)b__ * 32 * pitch__ : base__ + b__ * 32;                     \
      _Pragma("unroll") for (int oct__ = 0; oct__ < 4; ++oct__)                                            \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bb__ + dro[oct__] * pitch__ + dch[oct__]),            \
                                         (lds_ptr_t)(st__ + b__ * BLK + oct__ * 1024), 16, 0, 0);          \
    }                                                                                                      \
  }
  // panel p has landed once at most the DMAs of the NST - 2 newer panels are outstanding (VMEM operations retire in order); the
  // barrier makes every wave's share visible and proves every wave is past panel p - 1, whose stage panel p + NST - 1 then overwrites
#define EFX_BEGIN(p_)                                                                                      \
  {                                                                                                        \
    if (!EFX_PROBE_DMA) LOFTR_WAITCNT_VM(0);                                                               \
    else if (NST >= 4 && (p_) + 2 < NPANEL) LOFTR_WAITCNT_VM(2 * DMA_PER_WAVE);                            \
    else if ((p_) + 1 < NPANEL) LOFTR_WAITCNT_VM(DMA_PER_WAVE);                                            \
    else LOFTR_WAITCNT_VM(0);                                                                              \
    if (EFX_PROBE_BARRIER) __builtin_amdgcn_s_barrier();                                                   \
    if (EFX_PROBE_DMA && (p_) + NST - 1 < NPANEL) EFX_ISSUE((p_) + NST - 1);                               \
  }
  const int a_off = lds_chunk_off(li, g);               // hi chunk of the even k-step; odd k-step: ^ 32, lo: ^ 64
  // LDS fragment reads run one UNIT (four 16-B fragments, six MFMAs = 192 matrix-pipe cycles) ahead of the MFMAs that
  // consume them -- one wave per SIMD: nobody else hides the ds_read latency.  With LDS-DMA in flight hipcc turns every
  // LDS wait into lgkmcnt(0); EFX_USE (an empty asm that names the current fragments) pins that wait BEFORE the next
  // unit's reads are issued, so it only ever waits for reads issued a whole unit earlier.
#define EFX_RD(st_, blk_, odd_, lo_) (*reinterpret_cast<const h16x8*>((st_) + (blk_) * BLK + (a_off ^ (((odd_) ? 32 : 0) | ((lo_) ? 64 : 0)))))
#define EFX_USE(a_, b_, c_, d_) asm volatile("" :: "v"(a_), "v"(b_), "v"(c_), "v"(d_))
#if EFX_PIPE
  // ---- fragment reads as inline asm with COUNTED lgkmcnt waits, EFX_PIPE units ahead (the idiom of score_sweep.h, SWEEP_PIPE).
  // With compiler-visible reads every wait is lgkmcnt(0) (previous note), i.e. a unit can only be ONE unit ahead, and a read
  // issued behind the first MFMAs of unit u has ~5 MFMAs (160 cycles) to land: the measured LDS latency under four waves' fragment
  // traffic plus the DMA fills is longer, and with one wave per SIMD every late fragment is a matrix-pipe bubble (bare loop: 42
  // cycles per MFMA against the 32.7 of tools/micro/mfma_chain.hip).  LDS reads return in order: "unit u has landed" is
  // lgkmcnt(4 x younger units in flight); the wait statement names the fragments it releases ("+v"), which keeps their MFMAs below it.
  h16x8 fr[EFX_PIPE + 1][4];
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
#define EFX_LOADU(buf_, A0_, A1_, A2_, A3_, O01_, O23_)                                                    \
    asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\t"                      \
                 "ds_read_b128 %2, %6 offset:%9\n\tds_read_b128 %3, %7 offset:%9"                           \
                 : "=&v"(fr[buf_][0]), "=&v"(fr[buf_][1]), "=&v"(fr[buf_][2]), "=&v"(fr[buf_][3])          \
                 : "v"(A0_), "v"(A1_), "v"(A2_), "v"(A3_), "i"(O01_), "i"(O23_));
#define EFX_WAITU(buf_, n_)                                                                                \
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fr[buf_][0]), "+v"(fr[buf_][1]), "+v"(fr[buf_][2]), "+v"(fr[buf_][3]) : "i"(n_));
#define EFX_YOUNGER(u_) (4 * ((u_) + EFX_PIPE < 8 ? EFX_PIPE : 7 - (u_)))
  // R panel (32 rows x 256 k): unit u = k-group u (block u): even k-step hi / lo, odd k-step hi / lo
#define EFX_RLOAD(u_) EFX_LOADU((u_) % (EFX_PIPE + 1), ad0__, ad1__, ad2__, ad3__, (u_) * BLK, (u_) * BLK)
#define EFX_RUNIT(u_, bh_, bl_)                                                                            \
    if ((u_) + EFX_PIPE < 8) EFX_RLOAD((u_) + EFX_PIPE)                                                    \
    EFX_WAITU((u_) % (EFX_PIPE + 1), EFX_YOUNGER(u_))                                                      \
    {                                                                                                      \
      const h16x8 eh__ = fr[(u_) % (EFX_PIPE + 1)][0], el__ = fr[(u_) % (EFX_PIPE + 1)][1];                \
      const h16x8 oh__ = fr[(u_) % (EFX_PIPE + 1)][2], ol__ = fr[(u_) % (EFX_PIPE + 1)][3];                \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh__, bl_[2 * (u_)], acc0, 0, 0, 0);                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(el__, bh_[2 * (u_)], acc0, 0, 0, 0);                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh__, bh_[2 * (u_)], acc0, 0, 0, 0);                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh__, bl_[2 * (u_) + 1], acc0, 0, 0, 0);               \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ol__, bh_[2 * (u_) + 1], acc0, 0, 0, 0);               \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh__, bh_[2 * (u_) + 1], acc0, 0, 0, 0);               \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);
#define EFX_RPANEL(st_, bh_, bl_)                                                                          \
  {                                                                                                        \
    const unsigned stb__ = (unsigned)(size_t)(lds_ptr_t)(st_);                                             \
    const unsigned ad0__ = stb__ + a_off, ad1__ = stb__ + (a_off ^ 64), ad2__ = stb__ + (a_off ^ 32), ad3__ = stb__ + (a_off ^ 96); \
    EFX_RLOAD(0) if (EFX_PIPE > 1) EFX_RLOAD(1)                                                            \
    EFX_RUNIT(0, bh_, bl_) EFX_RUNIT(1, bh_, bl_) EFX_RUNIT(2, bh_, bl_) EFX_RUNIT(3, bh_, bl_)            \
    EFX_RUNIT(4, bh_, bl_) EFX_RUNIT(5, bh_, bl_) EFX_RUNIT(6, bh_, bl_) EFX_RUNIT(7, bh_, bl_)            \
  }
  // K panel (256 rows x 32 k): unit u = (output panels 2 (u >> 1), 2 (u >> 1) + 1; k-step u & 1): blocks 2 (u >> 1) and + 1
#define EFX_KLOAD(u_)                                                                                      \
    if ((u_) & 1) EFX_LOADU((u_) % (EFX_PIPE + 1), ad2__, ad3__, ad2__, ad3__, 2 * ((u_) >> 1) * BLK, (2 * ((u_) >> 1) + 1) * BLK) \
    else EFX_LOADU((u_) % (EFX_PIPE + 1), ad0__, ad1__, ad0__, ad1__, 2 * ((u_) >> 1) * BLK, (2 * ((u_) >> 1) + 1) * BLK)
#define EFX_KUNIT(u_, fh_, fl_, out_)                                                                      \
    if ((u_) + EFX_PIPE < 8) { EFX_KLOAD((u_) + EFX_PIPE) }                                                \
    EFX_WAITU((u_) % (EFX_PIPE + 1), EFX_YOUNGER(u_))                                                      \
    {                                                                                                      \
      const h16x8 ah__ = fr[(u_) % (EFX_PIPE + 1)][0], al__ = fr[(u_) % (EFX_PIPE + 1)][1];                \
      const h16x8 bh__ = fr[(u_) % (EFX_PIPE + 1)][2], bl__ = fr[(u_) % (EFX_PIPE + 1)][3];                \
      constexpr int ja__ = 2 * ((u_) >> 1), jb__ = 2 * ((u_) >> 1) + 1, s__ = (u_) & 1;                    \
      out_[ja__] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah__, fl_[s__], out_[ja__], 0, 0, 0);            \
      out_[jb__] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh__, fl_[s__], out_[jb__], 0, 0, 0);            \
      out_[ja__] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al__, fh_[s__], out_[ja__], 0, 0, 0);            \
      out_[jb__] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl__, fh_[s__], out_[jb__], 0, 0, 0);            \
      out_[ja__] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah__, fh_[s__], out_[ja__], 0, 0, 0);            \
      out_[jb__] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh__, fh_[s__], out_[jb__], 0, 0, 0);            \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);
#define EFX_KPANEL(st_, fh_, fl_, out_)                                                                    \
  {                                                                                                        \
    const unsigned stb__ = (unsigned)(size_t)(lds_ptr_t)(st_);                                             \
    const unsigned ad0__ = stb__ + a_off, ad1__ = stb__ + (a_off ^ 64), ad2__ = stb__ + (a_off ^ 32), ad3__ = stb__ + (a_off ^ 96); \
    { EFX_KLOAD(0) } if (EFX_PIPE > 1) { EFX_KLOAD(1) }                                                    \
    EFX_KUNIT(0, fh_, fl_, out_) EFX_KUNIT(1, fh_, fl_, out_) EFX_KUNIT(2, fh_, fl_, out_) EFX_KUNIT(3, fh_, fl_, out_) \
    EFX_KUNIT(4, fh_, fl_, out_) EFX_KUNIT(5, fh_, fl_, out_) EFX_KUNIT(6, fh_, fl_, out_) EFX_KUNIT(7, fh_, fl_, out_) \
  }
#else
  // R panel: acc0 += W[32 rows][256 k] . B fragments bh / bl; unit u = k-group u = k-steps 2u, 2u + 1.  ONE accumulator chain:
  // a dependent v_mfma_f32_32x32x16_f16 issues at the full rate (tools/micro/mfma_chain.hip: 32.7 cycles per MFMA with 1, 2, 3
  // or 4 chains).  The next unit's four fragment reads are issued in pairs BEHIND the first two MFMAs of this unit -- each pair
  // in the shadow of a 32-cycle MFMA instead of as a burst in front of the unit -- and still >= 4 MFMAs ahead of their wait.
#define EFX_RPANEL(st_, bh_, bl_)                                                                          \
  {                                                                                                        \
    h16x8 eh__ = EFX_RD(st_, 0, 0, 0), el__ = EFX_RD(st_, 0, 0, 1), oh__ = EFX_RD(st_, 0, 1, 0), ol__ = EFX_RD(st_, 0, 1, 1); \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                        \
      EFX_USE(eh__, el__, oh__, ol__);                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      h16x8 neh__ = eh__, nel__ = el__, noh__ = oh__, nol__ = ol__;                                        \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh__, bl_[2 * u], acc0, 0, 0, 0);                      \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (u + 1 < 8) { neh__ = EFX_RD(st_, u + 1, 0, 0); nel__ = EFX_RD(st_, u + 1, 0, 1); }               \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(el__, bh_[2 * u], acc0, 0, 0, 0);                      \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (u + 1 < 8) { noh__ = EFX_RD(st_, u + 1, 1, 0); nol__ = EFX_RD(st_, u + 1, 1, 1); }               \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh__, bh_[2 * u], acc0, 0, 0, 0);                      \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh__, bl_[2 * u + 1], acc0, 0, 0, 0);                  \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ol__, bh_[2 * u + 1], acc0, 0, 0, 0);                  \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh__, bh_[2 * u + 1], acc0, 0, 0, 0);                  \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      eh__ = neh__; el__ = nel__; oh__ = noh__; ol__ = nol__;                                              \
    }                                                                                                      \
  }
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

#endif
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
#undef EFX_RPANEL
#undef EFX_KPANEL
#undef EFX_RD
#undef EFX_USE
#if EFX_PIPE
#undef EFX_LOADU
#undef EFX_WAITU
#undef EFX_YOUNGER
#undef EFX_RLOAD
#undef EFX_RUNIT
#undef EFX_KLOAD
#undef EFX_KUNIT
#endif

  // ================= out = x + LayerNorm2(mlp.2 output), fp32 and SP                                          transformer.py:55-58
  if (!live) return;
  {
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
    if (tok < T) {
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
          *reinterpret_cast<f32x4*>(of + f) = o;
        }
        if (!a.out_sp) continue;                          // (loftr_encoder_layer_fwd: fp32 result only)
        h16x8 fh[2], fl[2];
        pack_panel(y, fh, fl);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          *reinterpret_cast<u32x4*>(os + j * 32 + (2 * s2 + g) * 4) = __builtin_bit_cast(u32x4, fh[s2]);
          *reinterpret_cast<u32x4*>(os + j * 32 + 16 + (2 * s2 + g) * 4) = __builtin_bit_cast(u32x4, fl[s2]);
        }
      }
    }
  }
}

__global__ __launch_bounds__(W * 64, 1) void encoder_x_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  encoder_x_body(a, blockIdx.x, lds);
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
  encoder_x_body(*a, second ? (int)blockIdx.x - m.n0 + m.off1 : (int)blockIdx.x + m.off0, lds);
}

// ------------------------------------------------------------------------------------------------------------------
// The TAIL of a launch.  encoder_x_kernel's unit of work is one wave x 32 tokens for the whole layer (~110 us): a call with
// 1200 blocks on the 1024 SIMDs costs two rounds although it holds 1.17 rounds of work.  The blocks beyond the last full
// round of workgroups go through this kernel instead: ONE 32-token block per workgroup, its four waves sharing every panel:
//   * an R panel (32 output features x 256 k) is split along K -- wave w owns k-steps 4 w .. 4 w + 3, i.e. features 64 w .. + 63 of
//     the input, and holds only that slice of x and of the message (32 registers each); the four partial 32 x 32 tiles are
//     all-reduced through a 16 KB LDS buffer (fixed order: every wave gets bitwise the same sum) and every wave runs the
//     (cheap) epilogue redundantly, so all of them hold the resulting Q' / hidden fragments;
//   * a K panel (256 output features x 32 k) is split along N -- wave w owns output panels 2 w, 2 w + 1: no exchange, and what it
//     accumulates (features 64 w .. + 63 of the message, then of the output) is exactly the K slice it needs next: the N split
//     of one stage IS the K split of the following one.  LayerNorm combines four per-wave (mean, M2) pairs (exact parallel form).
// Per block: 864 MFMAs per wave instead of 3072, 24 exchanges; the weights are streamed per 32 tokens here (4x the main kernel's
// L2 traffic per token), which is why this is the tail path and not the kernel.
constexpr int OFF_XB = OFF_TAB + T_N * 4, OFF_ST = OFF_XB + 4 * 4096, LDS_BYTES_C = OFF_ST + 4 * 32 * 8;
static_assert(LDS_BYTES_C <= 160 * 1024, "one workgroup per CU");

__global__ __launch_bounds__(W * 64, 1) void encoder_xc_kernel(Args a, int id0) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES_C];
  const int id = id0 + (int)(blockIdx.x >> 2), xcd = id % NUM_XCD, slot = id / NUM_XCD;
  const int seq = (slot / a.groups) * NUM_XCD + xcd, grp = slot % a.groups;
  const int blk = grp * W + (int)(blockIdx.x & 3);
  const int T = a.T;
  if (seq >= a.nseq || blk * PT >= T) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  const int tok = blk * PT + li;
  const long row = (long)seq * T + min(tok, T - 1);
  float* tab = reinterpret_cast<float*>(lds + OFF_TAB);
  for (int f = threadIdx.x; f < 256; f += W * 64) {
    tab[T_WQS + f] = a.wq_s[f];
    tab[T_KSUM + f] = a.kv[((long)seq * 8 + (f >> 5)) * (33 * 32) + 32 * 32 + (f & 31)];
    tab[T_G1 + f] = a.g1[f]; tab[T_B1 + f] = a.b1[f];
    tab[T_W0S + f] = a.w0_s[f]; tab[T_W0S + 256 + f] = a.w0_s[256 + f];
    tab[T_W2S + f] = a.w2_s[f];
    tab[T_G2 + f] = a.g2[f]; tab[T_B2 + f] = a.b2[f];
  }
  const float mk = (a.mask && !a.mask[row]) ? 0.f : 1.f;
  // this wave's K slice of x: k-steps 4 wave + i (features 64 wave + 16 i + 8 g + e)
  h16x8 xh[4], xl[4];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.x_sp + row * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ks = 4 * wave + i, c = (ks >> 1) * 8 + 2 * (ks & 1) + g;
      xh[i] = __builtin_bit_cast(h16x8, src[c]);
      xl[i] = __builtin_bit_cast(h16x8, src[c + 4]);
    }
  }
  const sp_t* pm = a.pm + (long)seq * a.pm_seq_stride;
  int dro[4], dch[4];
#pragma unroll
  for (int oct = 0; oct < 4; ++oct) {
    dro[oct] = oct * 8 + (lane >> 3);
    dch[oct] = ((lane & 7) ^ ((oct * 4 + (lane >> 4)) & 7)) << 2;
  }
  const int a_off = lds_chunk_off(li, g);
  const bool live = true; (void)live;
  // exchange buffer: partial tile of wave w, register quad q, lane l at ((w * 4 + q) * 64 + l) * 16
  const unsigned xb_w = (unsigned)(size_t)(lds_ptr_t)(lds + OFF_XB) + (unsigned)((wave * 4 * 64 + lane) * 16);
  const char* xb_r = lds + OFF_XB + lane * 16;
  const unsigned st_w = (unsigned)(size_t)(lds_ptr_t)(lds + OFF_ST) + (unsigned)((wave * 32 + li) * 8);
  const float2* st_r = reinterpret_cast<const float2*>(lds + OFF_ST) + li;
#define EFX_CRD(st_, blk_, odd_, lo_) (*reinterpret_cast<const h16x8*>((st_) + (blk_) * BLK + (a_off ^ (((odd_) ? 32 : 0) | ((lo_) ? 64 : 0)))))
  // this wave's share of an R panel: k-groups 2 wave, 2 wave + 1 against its K slice bh / bl (four k-steps)
#define EFX_CR(st_, bh_, bl_, acc_)                                                                        \
  {                                                                                                        \
    const char* s0__ = (st_) + (2 * wave) * BLK;                                                           \
    const h16x8 e0h = EFX_CRD(s0__, 0, 0, 0), e0l = EFX_CRD(s0__, 0, 0, 1), o0h = EFX_CRD(s0__, 0, 1, 0), o0l = EFX_CRD(s0__, 0, 1, 1); \
    const h16x8 e1h = EFX_CRD(s0__, 1, 0, 0), e1l = EFX_CRD(s0__, 1, 0, 1), o1h = EFX_CRD(s0__, 1, 1, 0), o1l = EFX_CRD(s0__, 1, 1, 1); \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0h, bl_[0], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0l, bh_[0], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0h, bh_[0], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(o0h, bl_[1], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(o0l, bh_[1], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(o0h, bh_[1], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1h, bl_[2], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1l, bh_[2], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1h, bh_[2], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(o1h, bl_[3], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(o1l, bh_[3], acc_, 0, 0, 0);                             \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(o1h, bh_[3], acc_, 0, 0, 0);                             \
  }
  // this wave's share of a K panel: output panels 2 wave, 2 wave + 1 against the two k-step fragments fh / fl
#define EFX_CK(st_, fh_, fl_, o0_, o1_)                                                                    \
  {                                                                                                        \
    const char* s0__ = (st_) + (2 * wave) * BLK;                                                           \
    const h16x8 a0h = EFX_CRD(s0__, 0, 0, 0), a0l = EFX_CRD(s0__, 0, 0, 1), a1h = EFX_CRD(s0__, 0, 1, 0), a1l = EFX_CRD(s0__, 0, 1, 1); \
    const h16x8 b0h = EFX_CRD(s0__, 1, 0, 0), b0l = EFX_CRD(s0__, 1, 0, 1), b1h = EFX_CRD(s0__, 1, 1, 0), b1l = EFX_CRD(s0__, 1, 1, 1); \
    o0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, fl_[0], o0_, 0, 0, 0);                               \
    o1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, fl_[0], o1_, 0, 0, 0);                               \
    o0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, fh_[0], o0_, 0, 0, 0);                               \
    o1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0l, fh_[0], o1_, 0, 0, 0);                               \
    o0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, fh_[0], o0_, 0, 0, 0);                               \
    o1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0h, fh_[0], o1_, 0, 0, 0);                               \
    o0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, fl_[1], o0_, 0, 0, 0);                               \
    o1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1h, fl_[1], o1_, 0, 0, 0);                               \
    o0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, fh_[1], o0_, 0, 0, 0);                               \
    o1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1l, fh_[1], o1_, 0, 0, 0);                               \
    o0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, fh_[1], o0_, 0, 0, 0);                               \
    o1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1h, fh_[1], o1_, 0, 0, 0);                               \
  }
  // all-reduce of the four waves' partial tiles (asm LDS stores: a compiler-visible one would drain the weight DMA); the
  // buffer is reused by the next exchange only after at least one panel barrier, which every wave reaches after its reads
#define EFX_XCHG(acc_, full_)                                                                              \
  {                                                                                                        \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                        \
      const f32x4 v4__ = {acc_[4 * q], acc_[4 * q + 1], acc_[4 * q + 2], acc_[4 * q + 3]};                 \
      asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(xb_w), "v"(v4__), "i"(q * 1024) : "memory");    \
    }                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                        \
      f32x4 s4__ = *reinterpret_cast<const f32x4*>(xb_r + (0 * 4 + q) * 1024);                             \
      s4__ += *reinterpret_cast<const f32x4*>(xb_r + (1 * 4 + q) * 1024);                                  \
      s4__ += *reinterpret_cast<const f32x4*>(xb_r + (2 * 4 + q) * 1024);                                  \
      s4__ += *reinterpret_cast<const f32x4*>(xb_r + (3 * 4 + q) * 1024);                                  \
      full_[4 * q] = s4__[0]; full_[4 * q + 1] = s4__[1]; full_[4 * q + 2] = s4__[2]; full_[4 * q + 3] = s4__[3]; \
    }                                                                                                      \
  }
  // LayerNorm statistics over the 256 features of a token: this wave holds 64 of them (v_[2][16] + the partner lane's)
#define EFX_CSTATS(v_, mean_, rstd_, eps_)                                                                 \
  {                                                                                                        \
    float s__ = 0.f;                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int r = 0; r < 16; ++r) s__ += v_[j][r]; \
    s__ += swap32(s__);                                                                                    \
    const float mw__ = s__ * (1.f / 64.f);                                                                 \
    float m2__ = 0.f;                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int r = 0; r < 16; ++r) { const float d = v_[j][r] - mw__; m2__ = fmaf(d, d, m2__); } \
    m2__ += swap32(m2__);                                                                                  \
    if (g == 0) { const f32x2 pr__ = {mw__, m2__}; asm volatile("ds_write_b64 %0, %1" :: "v"(st_w), "v"(pr__) : "memory"); } \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    const float2 e0 = st_r[0], e1 = st_r[32], e2 = st_r[64], e3 = st_r[96];                                \
    mean_ = (e0.x + e1.x + e2.x + e3.x) * 0.25f;                                                           \
    const float d0 = e0.x - mean_, d1 = e1.x - mean_, d2 = e2.x - mean_, d3 = e3.x - mean_;                \
    const float M2__ = (e0.y + e1.y + e2.y + e3.y) + 64.f * (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);       \
    rstd_ = rsqrtf(M2__ * (1.f / 256.f) + (eps_));                                                         \
  }

  LOFTR_WAITCNT_VM(0);
  __syncthreads();
  EFX_ISSUE(0);
  EFX_ISSUE(1);
  if (NST >= 4) EFX_ISSUE(2);
  const int fq = 4 * g;
  f32x16 acc, big0, big1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { big0[r] = 0.f; big1[r] = 0.f; }
  // ================= pass 1
#pragma unroll 1
  for (int h = 0; h < 8; ++h) {
    const int p = 2 * h;
    EFX_BEGIN(p);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    EFX_CR(lds + (p % NST) * STAGE, xh, xl, acc);
    float full[16];
    EFX_XCHG(acc, full);
    float v[16], den = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 ws = *reinterpret_cast<const f32x4*>(tab + T_WQS + 32 * h + fq + 8 * q);
      const f32x4 ks4 = *reinterpret_cast<const f32x4*>(tab + T_KSUM + 32 * h + fq + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = full[4 * q + e] * ws[e];
        x = x > 0.f ? x + 1.f : __expf(x);
        x *= mk;
        v[4 * q + e] = x;
        den = fmaf(x, ks4[e], den);
      }
    }
    den += swap32(den);
    const float z = a.v_length * __builtin_amdgcn_rcpf(den + a.attn_eps);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] *= z;
    h16x8 qh[2], ql[2];
    pack_panel(v, qh, ql);
    EFX_BEGIN(p + 1);
    EFX_CK(lds + ((p + 1) % NST) * STAGE, qh, ql, big0, big1);
  }
  // ---- message slice = LayerNorm1 over all 256 features, this wave's 64 -> its K slice of the mlp.0 input
  h16x8 mh[4], ml[4];
  {
    float vv[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { vv[0][r] = big0[r] * a.p_out_scale; vv[1][r] = big1[r] * a.p_out_scale; }
    float mean, rstd;
    EFX_CSTATS(vv, mean, rstd, a.ln_eps);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float y[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = 32 * (2 * wave + j) + fq + 8 * q;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(tab + T_G1 + f);
        const f32x4 be = *reinterpret_cast<const f32x4*>(tab + T_B1 + f);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[4 * q + e] = (vv[j][4 * q + e] - mean) * rstd * ga[e] + be[e];
      }
      h16x8 fh[2], fl[2];
      pack_panel(y, fh, fl);
      mh[2 * j] = fh[0]; mh[2 * j + 1] = fh[1]; ml[2 * j] = fl[0]; ml[2 * j + 1] = fl[1];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { big0[r] = 0.f; big1[r] = 0.f; }
  // ================= pass 2
#pragma unroll 1
  for (int hp = 0; hp < 16; ++hp) {
    const int p = 16 + 3 * hp;
    EFX_BEGIN(p);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    EFX_CR(lds + (p % NST) * STAGE, xh, xl, acc);
    EFX_BEGIN(p + 1);
    EFX_CR(lds + ((p + 1) % NST) * STAGE, mh, ml, acc);
    float full[16];
    EFX_XCHG(acc, full);
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 ws = *reinterpret_cast<const f32x4*>(tab + T_W0S + 32 * hp + fq + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * q + e] = fmaxf(full[4 * q + e] * ws[e], 0.f);
    }
    h16x8 hh[2], hl[2];
    pack_panel(v, hh, hl);
    EFX_BEGIN(p + 2);
    EFX_CK(lds + ((p + 2) % NST) * STAGE, hh, hl, big0, big1);
  }
  // ================= out slice = x + LayerNorm2(mlp.2 output), features 64 wave .. + 63
  {
    float vv[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ws = *reinterpret_cast<const f32x4*>(tab + T_W2S + 32 * (2 * wave + j) + fq + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[j][4 * q + e] = (j ? big1[4 * q + e] : big0[4 * q + e]) * ws[e];
      }
    float mean, rstd;
    EFX_CSTATS(vv, mean, rstd, a.ln_eps);
    if (tok < T) {
      const float* xr = a.x_f32 + row * 256;
      float* of = a.out_f32 + row * 256;
      sp_t* os = a.out_sp + row * 256;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int jp = 2 * wave + j;
        float y[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = 32 * jp + fq + 8 * q;
          const f32x4 ga = *reinterpret_cast<const f32x4*>(tab + T_G2 + f);
          const f32x4 be = *reinterpret_cast<const f32x4*>(tab + T_B2 + f);
          const f32x4 xr4 = *reinterpret_cast<const f32x4*>(xr + f);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = xr4[e] + ((vv[j][4 * q + e] - mean) * rstd * ga[e] + be[e]);
            y[4 * q + e] = o[e];
          }
          *reinterpret_cast<f32x4*>(of + f) = o;
        }
        if (!a.out_sp) continue;
        h16x8 fh[2], fl[2];
        pack_panel(y, fh, fl);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          *reinterpret_cast<u32x4*>(os + jp * 32 + (2 * s2 + g) * 4) = __builtin_bit_cast(u32x4, fh[s2]);
          *reinterpret_cast<u32x4*>(os + jp * 32 + 16 + (2 * s2 + g) * 4) = __builtin_bit_cast(u32x4, fl[s2]);
        }
      }
    }
  }
#undef EFX_CRD
#undef EFX_CR
#undef EFX_CK
#undef EFX_XCHG
#undef EFX_CSTATS
}
#undef EFX_ISSUE
#undef EFX_BEGIN
}  // namespace efx
}  // namespace

// LOFTR_FUSED_ENCODER=0 keeps the four-kernel form (A/B)
static bool fused_enabled() {
  static const bool on = []() { const char* e = getenv("LOFTR_FUSED_ENCODER"); return !(e && atoi(e) == 0); }();
  return on;
}

// the job's kernel arguments and its number of workgroups (0: not a shape the fused kernel takes)
static int make_job(const EncoderXArgs& p, efx::Args& a) {
  if (!fused_enabled() || p.C != 256 || p.nseq <= 0 || p.T <= 0 || !p.wq_s || !p.w0_s || !p.w2_s || !p.kv) return 0;
  a = efx::Args{};
  a.x_sp = p.x_sp; a.x_f32 = p.x_f32; a.out_f32 = p.out_f32; a.out_sp = p.out_sp;
  a.wq = p.wq; a.pm = p.pm; a.pm_seq_stride = p.pm_seq_stride; a.w0 = p.w0; a.w2 = p.w2;
  a.wq_s = p.wq_s; a.w0_s = p.w0_s; a.w2_s = p.w2_s; a.kv = p.kv; a.mask = p.mask;
  a.g1 = p.g1; a.b1 = p.b1; a.g2 = p.g2; a.b2 = p.b2;
  a.v_length = p.v_length; a.attn_eps = p.attn_eps; a.p_out_scale = p.p_out_scale; a.ln_eps = p.ln_eps;
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
  // Split the launch: the largest whole number of 256-workgroup rounds goes to the main kernel (a wave x 32 tokens for the whole
  // layer), the remainder to the cooperative tail kernel (a workgroup x 32 tokens) when that is the cheaper way to finish:
  // OFF by default (LOFTR_ENCODER_TAIL=1 enables it).  Measured at the bench size (tools/gpu/r3_tail.sh): a cooperative round costs
  // 65 us, not a third of the main kernel's 110 -- 64 barriers, 24 exchanges and a 2 MB weight stream per 32 tokens -- so only a
  // single tail round pays (cross calls: 221 -> 175 us; 2.91 -> 2.79 ms per transformer call, -4 %), while a pair's result would
  // then depend, in the last bits, on the batch it was part of (test_batch_consistency_full_size) and the two-stream bench fills
  // the idle SIMDs of a partial round with convolution workgroups anyway.
  static const bool tail_on = []() { const char* e = getenv("LOFTR_ENCODER_TAIL"); return e && atoi(e) == 1; }();
  auto live = [&](int id) { const int xcd = id % NUM_XCD, slot = id / NUM_XCD; return (slot / a.groups) * NUM_XCD + xcd < a.nseq; };
  int total = 0;
  for (int id = 0; id < grid; ++id) total += live(id);
  const int full = total / 256 * 256;
  int g_main = 0;
  for (int cnt = 0; g_main < grid && cnt < full; ++g_main) cnt += live(g_main);
  int tail = 0;
  for (int id = g_main; id < grid; ++id) tail += live(id);
  const bool use_tail = tail_on && a.xsplit == 1 && tail > 0 && ceil_div(tail * efx::W, 256) <= 1;      // (the tail kernel keeps the one-sequence-per-XCD mapping)
  if (!use_tail) g_main = grid;
  TimedLaunch tl(LOFTR_T_ENCODER_X, st);
  if (g_main > 0) hipLaunchKernelGGL(efx::encoder_x_kernel, dim3(g_main), dim3(efx::W * 64), 0, st, a);
  if (g_main < grid) hipLaunchKernelGGL(efx::encoder_xc_kernel, dim3((grid - g_main) * efx::W), dim3(efx::W * 64), 0, st, a, g_main);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
