// ResNet-FPN building blocks on the split-fp16 GEMM core (the backbone is
// ~80 % of LoFTR.forward once the matching path is fast).
//   reference: src/loftr/backbone/resnet_fpn.py:5-118 (conv3x3 / conv1x1 / BasicBlock / FPN head)
//
// A convolution is an implicit GEMM: rows = output pixels, k = (filter tap, input channel), columns =
// output channels; the A operand is gathered on the fly from the NHWC activation tensor (gemm.h, CONV
// mode).  Activations live in HBM in the SP format [B, H, W, Cp] (channels padded to a multiple of 32,
// pad channels hold zeros).  Eval-mode BatchNorm is folded into the weights (scale) and a per-channel
// bias while they are re-laid out [Cout, Cin, KH, KW] fp32 -> [Cout, KH*KW*Cp] SP; the epilogue adds
// the bias and the residual branch and applies ReLU / LeakyReLU.
#include "gemm.h"
#include <stdlib.h>

namespace {

using CfgD = GemmCfg<256, 128, 4, 2, 3>;     // 8 waves, 3-stage global_load_lds ring, 1 workgroup per CU

struct ConvArgs {
  ASrc a;                       // asrc_conv(x, geometry)
  const sp_t* w; int K;         // [Cout, K] SP, K = KH*KW*Cp
  const float* bias;            // [Cout] (folded BN shift) or null
  const float* wscale;          // [Cout] inverse power-of-two scale of filter row co (conv_prep_kernel; gemm.h)
  const float* x_inv;           // device scalar: inverse scale of the input activation tensor, or null (= 1)
  const sp_t* residual;         // [M, Coutp] SP or null (added before the activation)
  sp_t* y_sp;                   // [M, Coutp] SP or null
  float* y_f32;                 // [M, Cout] fp32 (NHWC) or null
  int M, Cout, Coutp;
  int act;                      // 0 none, 1 ReLU, 2 LeakyReLU(0.01)
  const sp_t* up;               // FPN top-down input [B, Hl, Wl, Coutp] SP or null: y += bilinear_x2(up), align_corners
  int Hl, Wl;                   // (output pixels are then [B, 2Hl, 2Wl])
  float sy, sx;                 // (Hl-1)/(2Hl-1), (Wl-1)/(2Wl-1)
};

template <typename Cfg, bool FULL>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0) {
  const EpiLane<Cfg> e;
  const sp_t* rs = p.residual ? p.residual + (long)m0 * p.Coutp + n0 : nullptr;
  sp_t* os = p.y_sp ? p.y_sp + (long)m0 * p.Coutp + n0 : nullptr;
  float* of = p.y_f32 ? p.y_f32 + (long)m0 * p.Cout + n0 : nullptr;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + e.lcol + j * 32;
    const bool creal = col < p.Cout;              // a real output channel
    const bool cpad = col < p.Coutp;              // inside the padded SP row (pad channels are written as 0)
    const float b = (p.bias && creal) ? p.bias[col] : 0.f;
    const float wsc = (creal ? p.wscale[col] : 1.f) * (p.x_inv ? *p.x_inv : 1.f);      // undo the operands' power-of-two scales
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      f32x16 v = acc[i][j] * wsc;
      if (rs) {
        uint32_t w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int trow = e.lrow + e.rr(i, r);
          const bool ok = (FULL || m0 + trow < p.M) && cpad;
          w[r] = ok ? rs[(unsigned)(trow * p.Coutp + e.spcol + j * 32)] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += sp_value(w[r], e.odd);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = v[r] + b;
        if (p.act == 1) x = fmaxf(x, 0.f);
        if (p.act == 2) x = x > 0.f ? x : 0.01f * x;
        v[r] = creal ? x : 0.f;
      }
      if (of) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int trow = e.lrow + e.rr(i, r);
          if ((FULL || m0 + trow < p.M) && creal) of[(unsigned)(trow * p.Cout + e.lcol + j * 32)] = v[r];
        }
      }
      if (os) {
        uint32_t w[16];
        sp_words16(v, e.odd, w);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int trow = e.lrow + e.rr(i, r);
          if ((FULL || m0 + trow < p.M) && cpad) os[(unsigned)(trow * p.Coutp + e.spcol + j * 32)] = w[r];
        }
      }
    }
  }
}

// FPN top-down epilogue (resnet_fpn.py:110-112 / :115-117): y = acc + bilinear_x2(up), SP out.  The accumulators
// are parked in LDS (the operand ring is dead by now) so that the gather of the half-resolution map and the
// stores run in (pixel, channel-octet) form -- 16-B accesses, one geometry computation per eight channels --
// instead of four scalar gathers per accumulator register.
template <typename Cfg>
__device__ __forceinline__ void conv_epilogue_up(const ConvArgs& p, f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0,
                                                 float* lds) {
  constexpr int LD = Cfg::BN + 4;
  static_assert(Cfg::BM * LD <= Cfg::LDS_FLOATS, "accumulator tile does not fit the operand ring");
  const EpiLane<Cfg> e;
  __syncthreads();                                  // every wave is done reading the last k-tile
  const float xinv = p.x_inv ? *p.x_inv : 1.f;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + e.lcol + j * 32;
    const float wsc = (col < p.Cout ? p.wscale[col] : 1.f) * xinv;        // undo the operands' power-of-two scales
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) lds[(e.lrow + e.rr(i, r)) * LD + e.lcol + j * 32] = acc[i][j][r] * wsc;
  }
  __syncthreads();
  const int Wo = 2 * p.Wl, Ho = 2 * p.Hl;
  constexpr int OCTS = Cfg::BN / 8;
  for (int it = threadIdx.x; it < Cfg::BM * OCTS; it += Cfg::THREADS) {
    const int row = it / OCTS, oct = it - row * OCTS;
    const int pix = m0 + row, col0 = n0 + oct * 8;
    if (pix >= p.M || col0 >= p.Coutp) continue;
    const int t = pix / Wo, x = pix - t * Wo;
    const int bb = t / Ho, y = t - bb * Ho;
    // torch upsample_bilinear2d, align_corners=True: src = dst * (in - 1) / (out - 1)
    const float fy = p.sy * (float)y, fx = p.sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int dy = y0 < p.Hl - 1 ? p.Wl * p.Coutp : 0, dx = x0 < p.Wl - 1 ? p.Coutp : 0;
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const int off = sp_octet_off(col0 >> 3);
    const sp_t* s00 = p.up + (unsigned)(((bb * p.Hl + y0) * p.Wl + x0) * p.Coutp + off);
    const sp_t* src[4] = {s00, s00 + dx, s00 + dy, s00 + dy + dx};
    u32x4 hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = *reinterpret_cast<const u32x4*>(src[k]);
      lo[k] = *reinterpret_cast<const u32x4*>(src[k] + 16);
    }
    float v[4][8], o[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) sp_unpack8(hi[k], lo[k], v[k]);
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(&lds[row * LD + oct * 8]);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(&lds[row * LD + oct * 8 + 4]);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      o[c] = col0 + c < p.Cout      // pad channels of the SP row are written as 0 (their weight rows are clamped reads)
                 ? (c < 4 ? a0[c] : a1[c - 4]) + (hy * (hx * v[0][c] + lx * v[1][c]) + ly * (hx * v[2][c] + lx * v[3][c]))
                 : 0.f;
    u32x4 oh, ol;
    sp_pack8(o, oh, ol);
    sp_t* dst = p.y_sp + (unsigned)(pix * p.Coutp + off);
    *reinterpret_cast<u32x4*>(dst) = oh;
    *reinterpret_cast<u32x4*>(dst + 16) = ol;
  }
}

template <typename Cfg, bool UP>
__global__ __launch_bounds__(Cfg::THREADS, 2) void conv_kernel(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  int tm, tn;
  if (!xcd_tile(ceil_div(p.M, Cfg::BM), ceil_div(p.Coutp, Cfg::BN), tm, tn)) return;
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
  f32x16 acc[Cfg::TM][Cfg::TN];
  gemm_mainloop<Cfg, true>(p.a, p.w, p.K, p.M, p.Cout, p.K, m0, n0, lds, acc, p.Coutp);
  if constexpr (UP) conv_epilogue_up<Cfg>(p, acc, m0, n0, lds);
  else if (m0 + Cfg::BM <= p.M) conv_epilogue<Cfg, true>(p, acc, m0, n0);
  else conv_epilogue<Cfg, false>(p, acc, m0, n0);
}

// ------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions (12 of the 21 backbone convolutions, ~85 % of their time): the nine
// filter taps of one 32-channel group read nine shifted views of the SAME pixels, so instead of DMA-ing a
// 256-row A tile per (tap, channel group) k-tile the workgroup keeps the (8+2) x (32+2) input patch of its
// 8 x 32 output tile in LDS -- one patch load per channel group serves nine k-tiles (A traffic / 9, and with
// it the DMA issue + LDS fill work the isolation runs in DESIGN.md §5 identified as the main-loop overhead).
//   tile   : 8 x 32 output pixels of one image = 256 GEMM rows, x 128 output channels; 8 waves (4 x 2): wave
//            (wm, wn) owns output rows 2wm, 2wm+1 (two 32-pixel MFMA row tiles) x 64 channels.
//   LDS    : patch ring 2 x 344 rows x 128 B (zero rows outside the image come from a zero page), weight ring
//            3 x 128 rows x 128 B, same 16-B-chunk XOR swizzle as gemm.h (applied to the global address).
//   k order: channel group outermost, then the tap column kx, tap row ky innermost: k-tile t = (cg, kx, ky) reads
//            weight tap ky * 3 + kx; consecutive ky share one of their two patch rows, kept in registers.
//   DMA    : per wave 6 patch instructions per channel group (issued at the group's first k-tile for the NEXT
//            group) and 2 weight instructions per k-tile (two tiles ahead); loads retire in order, so "weight
//            tile t landed" is vmcnt(2 [+6 if a patch was issued in one of the last two iterations]).
namespace c3 {
constexpr int TY = 8, TX = 32, PW = TX + 2, PH = TY + 2, PROWS = PW * PH;            // 340 patch pixels
constexpr int PSLOTS = 43, PQ = 6;                                                     // ceil(340 / 8) DMA slots, 6 per wave
constexpr int PATCH_BYTES = PSLOTS * 1024, BTILE_BYTES = 128 * 128;
constexpr int NB = 4;                                                                  // weight ring stages
constexpr int LDS_BYTES = 2 * PATCH_BYTES + NB * BTILE_BYTES + 1024;                    // + 1 KB scratch for the unused slots
using Cfg = GemmCfg<256, 128, 4, 2, 3>;                                                 // wave layout / epilogue helpers only
}  // namespace c3

struct Conv3Args {
  const sp_t* x; int B, H, W, Cp, Cin;
  const sp_t* w; int K;
  const float* bias; const float* wscale; const float* x_inv; const sp_t* residual; sp_t* y_sp; float* y_f32;
  int Cout, Coutp, act;
  const sp_t* zeros;
  int tiles_x, tiles_y;
  float* pbuf;                  // conv3x3_duo.h, Cfg<.., REM>: scratch [pixels][9 R] of the tap-decomposed remainder channels
};

__global__ __launch_bounds__(512, 2) void conv3x3_kernel(Conv3Args p) {
  using namespace c3;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  char* const patch_base = lds;
  char* const bring_base = lds + 2 * PATCH_BYTES;
  char* const scratch = lds + 2 * PATCH_BYTES + NB * BTILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int rsub = lane >> 3, slot = lane & 7;

  // ---- persistent workgroup: one per CU, walking the XCD-chunked tile order of common.h with a grid stride
  // (a multiple of 8, so a workgroup stays on its XCD).  The DMA prologue of the NEXT tile is issued before the
  // epilogue of the current one: the ~2 us patch / weight fetch latency at the start of a tile and the store
  // drain at its end no longer leave the matrix pipe idle (1 workgroup per CU: nothing else would cover them).
  const int tiles_m = p.B * p.tiles_y * p.tiles_x, tiles_n = ceil_div(p.Coutp, 128);
  const int chunk_m = ceil_div(tiles_m, NUM_XCD), nvirt = NUM_XCD * chunk_m * tiles_n;
  int vid = blockIdx.x;
  int nb = 0, ny0 = 0, nx0 = 0, nn0 = 0;       // next tile: image, first output row / column, first output channel
  bool have = false;
#define C3_NEXT_TILE()                                                                                      \
  {                                                                                                         \
    have = false;                                                                                           \
    for (; vid < nvirt; vid += gridDim.x) {                                                                 \
      const int xcd__ = vid % NUM_XCD, slot__ = vid / NUM_XCD;                                              \
      const int local__ = slot__ / tiles_n, tm__ = xcd__ * chunk_m + local__;                               \
      if (local__ < chunk_m && tm__ < tiles_m) {                                                            \
        nb = tm__ / (p.tiles_y * p.tiles_x);                                                                \
        const int trem__ = tm__ - nb * (p.tiles_y * p.tiles_x);                                             \
        ny0 = (trem__ / p.tiles_x) * TY; nx0 = (trem__ % p.tiles_x) * TX; nn0 = (slot__ % tiles_n) * 128;   \
        have = true;                                                                                        \
        break;                                                                                              \
      }                                                                                                     \
    }                                                                                                       \
  }
  C3_NEXT_TILE()
  if (!have) return;

  // ---- DMA source offsets (dwords) of the next tile -------------------------------------------
  int poff[PQ];                       // patch rows: -1 = outside the image / unused slot -> zero page
  int boff[2];
#define C3_DMA_OFFSETS()                                                                                    \
  {                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < PQ; ++q) {                                                        \
      const int s = q * 8 + wave, r = s * 8 + rsub;                                                         \
      const int py = r / PW, px = r - py * PW;                                                              \
      const int gy = ny0 - 1 + py, gx = nx0 - 1 + px;                                                       \
      const bool in = s < PSLOTS && r < PROWS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W; \
      poff[q] = in ? ((nb * p.H + gy) * p.W + gx) * p.Cp + ((slot ^ ((r >> 1) & 7)) << 2) : -1;             \
    }                                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                         \
      const int r = (q * 8 + wave) * 8 + rsub;                                                              \
      boff[q] = min(nn0 + r, p.Cout - 1) * p.K + ((slot ^ ((r >> 1) & 7)) << 2);                            \
    }                                                                                                       \
  }
  C3_DMA_OFFSETS()
  const int gpt = p.Cp >> 5, nk = 9 * gpt;

#define C3_ISSUE_PATCH(cg_, stage_)                                                                         \
  {                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < PQ; ++q) {                                                        \
      const sp_t* g__ = poff[q] >= 0 ? p.x + (poff[q] + (cg_) * 32) : p.zeros;                              \
      char* d__ = (q * 8 + wave < PSLOTS) ? patch_base + (stage_) * PATCH_BYTES + (q * 8 + wave) * 1024 : scratch; \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)g__, (lds_ptr_t)d__, 16, 0, 0);                           \
    }                                                                                                       \
  }
#define C3_ISSUE_B(cg_, tap_, stage_)                                                                       \
  {                                                                                                         \
    const int k0__ = (tap_) * p.Cp + (cg_) * 32;                                                            \
    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                           \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.w + (boff[q] + k0__)),                                 \
                                       (lds_ptr_t)(bring_base + (stage_) * BTILE_BYTES + (q * 8 + wave) * 1024), 16, 0, 0); \
  }

  f32x16 acc[2][2];
  const int g = lane >> 5, tx = lane & 31;
  int bbase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int br = wn * 64 + j * 32 + tx;
    bbase[j] = br * 128 + ((g ^ ((br >> 1) & 7)) << 4);
  }

  // prologue of the first tile: patch 0, weight tiles 0, 1, 2 (sequence positions (kx 0, ky 0..2) = taps 0, 3, 6; nk >= 9)
  C3_ISSUE_PATCH(0, 0);
  C3_ISSUE_B(0, 0, 0);
  C3_ISSUE_B(0, 3, 1);
  C3_ISSUE_B(0, 6, 2);
  for (;;) {                                             // ---- tiles of this workgroup ----
  const int b = nb, y0 = ny0, x0 = nx0, n0 = nn0;
  const int nact = min(2, (p.Coutp - (n0 + wn * 64) + 31) / 32);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int bstage = 0;                                        // ring stage of k-tile t (tile t+3 is issued into stage t-1)
  int cg = 0, kx = 0;                                    // this trip runs k-tiles (cg, kx, ky = 0, 1, 2)
  int cg3 = 0, q3 = 3;                                   // sequence position (kx * 3 + ky) of the k-tile issued next, three ahead
  bool patch_m1 = false, patch_m2 = false;               // a patch was issued in iteration t-1 / t-2
  // A fragments of the wave's four patch rows 2wm .. 2wm+3 at column offset kx, [row][16-wide k-step]: output rows
  // (2wm, 2wm+1) read patch rows (ky, ky+1), so walking ky innermost each k-tile after the first needs ONE new row
  // (4 instead of 6 row reads per kx: a third of the A-side LDS traffic stays in registers).
  h16x8 fh[4][2], fl[4][2];
  h16x8 bh[2][2], bl[2][2];                              // B fragments [k-step][column tile]

  // Software pipeline across the k-tile barrier: the barrier of k-tile t guarantees that tiles <= t+1 have landed
  // (4-stage weight ring, three tiles in flight), so the k-step-0 fragments of tile t+1 are read during the k-step-1
  // MFMAs of tile t and the first MFMAs of a k-tile issue right after its barrier instead of behind an LDS round
  // trip.  The trip body is instantiated per (active column tiles NJ_, dead second k-step DEAD_): no
  // data-dependent branches inside the k-loop.
#define C3_LOAD_ROW(jr_, ks_, sP_, kx_)                                                                     \
  {                                                                                                         \
    const int pr__ = (wm * 2 + (jr_)) * PW + (kx_) + tx;           /* patch pixel of this lane's output pixel */ \
    const int ab__ = pr__ * 128 + ((g ^ ((pr__ >> 1) & 7)) << 4);                                            \
    /* chunk c = g | ks << 1 | lo << 2 (disjoint bits): the four chunks of a row are one address XOR {0,32,64,96} */ \
    fh[jr_][ks_] = *reinterpret_cast<const h16x8*>((sP_) + (ab__ ^ ((ks_) << 5)));                          \
    fl[jr_][ks_] = *reinterpret_cast<const h16x8*>((sP_) + (ab__ ^ (((ks_) << 5) | 64)));                   \
  }
#define C3_LOAD_B(ks_, stage_, NJ_)                                                                         \
  {                                                                                                         \
    const char* sB__ = bring_base + (stage_) * BTILE_BYTES;                                                 \
    _Pragma("unroll") for (int j = 0; j < (NJ_); ++j) {                                                     \
      bh[ks_][j] = *reinterpret_cast<const h16x8*>(sB__ + (bbase[j] ^ ((ks_) << 5)));                       \
      bl[ks_][j] = *reinterpret_cast<const h16x8*>(sB__ + (bbase[j] ^ (((ks_) << 5) | 64)));                \
    }                                                                                                       \
  }
#define C3_MFMAS(ks_, R0_, NJ_)                                                                             \
  {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
      _Pragma("unroll") for (int j = 0; j < (NJ_); ++j)                                                     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[(R0_) + i][ks_], bh[ks_][j], acc[i][j], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
      _Pragma("unroll") for (int j = 0; j < (NJ_); ++j)                                                     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[(R0_) + i][ks_], bl[ks_][j], acc[i][j], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
      _Pragma("unroll") for (int j = 0; j < (NJ_); ++j)                                                     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[(R0_) + i][ks_], bh[ks_][j], acc[i][j], 0, 0, 0); \
  }
#define C3_PIN_PAIRS(n_)                                                                                    \
    _Pragma("unroll") for (int u__ = 0; u__ < (n_); ++u__) {                                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                    \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                    \
    }
  // one k-tile: (cg, tap = KY * 3 + kx); its k-step-0 fragments are already in registers
#define C3_ITER(KY, NJ_, DEAD_)                                                                             \
  {                                                                                                         \
    const int tt__ = t + (KY);                                                                              \
    /* weight tile tt+1 (and, in order before it, every patch it can need) landed; younger loads stay in flight */ \
    const int newer__ = (tt__ + 2 < nk ? 2 : 0) + ((patch_m1 || patch_m2) ? PQ : 0);                        \
    if (newer__ >= 2 + PQ) LOFTR_WAITCNT_VM(2 + PQ);                                                        \
    else if (newer__ >= PQ) LOFTR_WAITCNT_VM(PQ);                                                           \
    else if (newer__ >= 2) LOFTR_WAITCNT_VM(2);                                                             \
    else LOFTR_WAITCNT_VM(0);                                                                               \
    __builtin_amdgcn_s_barrier();                                                                           \
    if (tt__ + 3 < nk) C3_ISSUE_B(cg3, (q3 % 3) * 3 + q3 / 3, (bstage + 3) & 3);                            \
    if (++q3 == 9) { q3 = 0; ++cg3; }                                                                       \
    patch_m2 = patch_m1;                                                                                    \
    patch_m1 = false;                                                                                       \
    if ((KY) == 0 && kx == 0 && cg + 1 < gpt) { C3_ISSUE_PATCH(cg + 1, (cg + 1) & 1); patch_m1 = true; }    \
    if ((NJ_) > 0) {                                                                                        \
      constexpr int n1__ = ((KY) == 0 ? 4 : 2) + 2 * (NJ_);        /* k-step-1 reads of this tile */         \
      constexpr int n0__ = ((KY) == 2 ? 4 : 2) + 2 * (NJ_);        /* k-step-0 reads of the next tile */     \
      if (!(DEAD_)) {                                                                                       \
        if ((KY) == 0) { C3_LOAD_ROW(0, 1, sP, kx); C3_LOAD_ROW(1, 1, sP, kx); }                            \
        else C3_LOAD_ROW((KY) + 1, 1, sP, kx);                                                              \
        C3_LOAD_B(1, bstage, NJ_);                                                                          \
      }                                                                                                     \
      C3_MFMAS(0, KY, NJ_);                                                                                 \
      /* next k-tile (after the last one: harmless reads of stale LDS) */                                   \
      if ((KY) == 2) { C3_LOAD_ROW(0, 0, sPn, kxn); C3_LOAD_ROW(1, 0, sPn, kxn); }                          \
      else C3_LOAD_ROW((KY) + 2, 0, sP, kx);                                                                \
      C3_LOAD_B(0, (bstage + 1) & 3, NJ_);                                                                  \
      if (!(DEAD_)) C3_MFMAS(1, KY, NJ_);                                                                   \
      /* pinned order: one fragment read behind each of the first MFMAs of a k-step */                      \
      if (!(DEAD_)) {                                                                                       \
        C3_PIN_PAIRS(n1__) __builtin_amdgcn_sched_group_barrier(0x008, 6 * (NJ_) - n1__, 0);                \
      }                                                                                                     \
      C3_PIN_PAIRS(n0__) __builtin_amdgcn_sched_group_barrier(0x008, 6 * (NJ_) - n0__, 0);                  \
    }                                                                                                       \
    bstage = (bstage + 1) & 3;                                                                              \
  }
  // three k-tiles (ky = 0, 1, 2) at (cg, kx)
#define C3_TRIP(NJ_, DEAD_)                                                                                 \
  {                                                                                                         \
    const char* sP = patch_base + (cg & 1) * PATCH_BYTES;                                                   \
    const int kxn = kx == 2 ? 0 : kx + 1;                          /* tap column / patch of the next trip */ \
    const char* sPn = patch_base + ((cg + (kx == 2 ? 1 : 0)) & 1) * PATCH_BYTES;                            \
    C3_ITER(0, NJ_, DEAD_)                                                                                  \
    C3_ITER(1, NJ_, DEAD_)                                                                                  \
    C3_ITER(2, NJ_, DEAD_)                                                                                  \
    if (++kx == 3) { kx = 0; ++cg; }                                                                        \
  }
  // channels >= Cin of the last group are zero padding (activations AND folded weights): when they fill the whole
  // second 16-wide k-step (e.g. Cin = 196 -> 192..207 | 208..223) its MFMAs are skipped -- exact, 1/14 of the work
  const bool dead_last = p.Cin <= (gpt - 1) * 32 + 16;
#define C3_LOOP(NJ_)                                                                                        \
  {                                                                                                         \
    /* pipeline fill: k-step-0 fragments of k-tile 0 (patch 0 and weight tile 0 landed: tiles 1, 2 may be in flight) */ \
    LOFTR_WAITCNT_VM(4);                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                           \
    if ((NJ_) > 0) {                                                                                        \
      C3_LOAD_ROW(0, 0, patch_base, 0);                                                                     \
      C3_LOAD_ROW(1, 0, patch_base, 0);                                                                     \
      C3_LOAD_B(0, 0, NJ_);                                                                                 \
    }                                                                                                       \
    int t = 0;                                                                                              \
    for (; t < nk - 9; t += 3) C3_TRIP(NJ_, 0)                                                              \
    if (dead_last) { for (; t < nk; t += 3) C3_TRIP(NJ_, 1) }                                               \
    else { for (; t < nk; t += 3) C3_TRIP(NJ_, 0) }                                                         \
  }
  if (nact == 2) C3_LOOP(2)
  else if (nact == 1) C3_LOOP(1)
  else C3_LOOP(0)
#undef C3_LOOP
#undef C3_TRIP
#undef C3_ITER
#undef C3_PIN_PAIRS
#undef C3_MFMAS
#undef C3_LOAD_B
#undef C3_LOAD_ROW

  // ---- next tile: its DMA prologue goes out before this tile's epilogue -----------------------------
  vid += gridDim.x;
  C3_NEXT_TILE()
  __builtin_amdgcn_s_barrier();                          // every wave is done reading the patch / weight rings
  if (have) {
    C3_DMA_OFFSETS()
    C3_ISSUE_PATCH(0, 0);
    C3_ISSUE_B(0, 0, 0);
    C3_ISSUE_B(0, 3, 1);
    C3_ISSUE_B(0, 6, 2);
  }

  // ---- epilogue: bias (folded BN shift), residual, activation, SP / fp32 stores ---------------------
  // (its loads and stores are younger than the prologue DMAs above: the vmcnt waits of the next tile's first
  //  k-tiles, which count only DMA instructions, are then stricter than needed, never weaker)
  const bool odd = lane & 1;
  const int Ho = p.H, Wo = p.W;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + tx;
    const bool creal = col < p.Cout, cpad = col < p.Coutp;
    const float bia = (p.bias && creal) ? p.bias[col] : 0.f;
    const float wsc = (creal ? p.wscale[col] : 1.f) * (p.x_inv ? *p.x_inv : 1.f);      // undo the operands' power-of-two scales
    const int spc = (col & ~31) + (odd ? 16 : 0) + ((col & 31) >> 1);      // dword of this lane inside the SP row
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = y0 + wm * 2 + i;
      f32x16 v = acc[i][j] * wsc;
      uint32_t rw[16];
      if (p.residual) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * g;
          const bool ok = y < Ho && x < Wo && cpad;
          rw[r] = ok ? p.residual[(unsigned)(((b * Ho + y) * Wo + x) * p.Coutp + spc)] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += sp_value(rw[r], odd);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float xv = v[r] + bia;
        if (p.act == 1) xv = fmaxf(xv, 0.f);
        if (p.act == 2) xv = xv > 0.f ? xv : 0.01f * xv;
        v[r] = creal ? xv : 0.f;
      }
      if (p.y_f32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (y < Ho && x < Wo && creal) p.y_f32[(unsigned)(((b * Ho + y) * Wo + x) * p.Cout + col)] = v[r];
        }
      }
      if (p.y_sp) {
        uint32_t w16[16];
        sp_words16(v, odd, w16);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (y < Ho && x < Wo && cpad) p.y_sp[(unsigned)(((b * Ho + y) * Wo + x) * p.Coutp + spc)] = w16[r];
        }
      }
    }
  }
  if (!have) break;
  }                                                      // ---- tiles of this workgroup ----
#undef C3_ISSUE_PATCH
#undef C3_ISSUE_B
#undef C3_DMA_OFFSETS
#undef C3_NEXT_TILE
}

#include "conv3x3_duo.h"


// Weight preparation: fold eval-mode BN, transpose to tap-major, pad channels, encode as SP.
//   w [Cout, Cin, KH, KW] -> wsp [Cout, KH*KW*Cp];  bias[co] = beta - mean * scale,  scale = gamma / sqrt(var + eps)
//   grid (ceil(groups_per_row / 8), Cout), 256 threads: one half-wave per 32-column SP group.
// Per output channel: BN scale / shift and the power-of-two that lifts the largest |folded weight| of the filter row
// to [2^13, 2^14) (gemm.h: sp_row_scale).   grid (Cout), 256 threads.
__global__ __launch_bounds__(256) void conv_rowscale_kernel(const float* __restrict__ w, const float* __restrict__ bn_w,
                                                            const float* __restrict__ bn_b, const float* __restrict__ bn_m,
                                                            const float* __restrict__ bn_v, float eps, int Cin, int KH, int KW,
                                                            long s_co, long s_ci, long s_ky, long s_kx,
                                                            float* __restrict__ bias, float* __restrict__ wscale) {
  const int co = blockIdx.x;
  float scale = 1.f, shift = 0.f;
  if (bn_w) {
    scale = bn_w[co] / sqrtf(bn_v[co] + eps);
    shift = bn_b[co] - bn_m[co] * scale;
  }
  float m = 0.f;
  for (int t = threadIdx.x; t < Cin * KH * KW; t += 256) {
    const int c = t % Cin, tap = t / Cin;
    m = fmaxf(m, fabsf(w[co * s_co + c * s_ci + (tap / KW) * s_ky + (tap % KW) * s_kx] * scale));
  }
  __shared__ float red[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float inv;
    (void)sp_row_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), &inv);
    bias[co] = shift;
    wscale[co] = inv;
  }
}

__global__ __launch_bounds__(256) void conv_prep_kernel(const float* __restrict__ w, const float* __restrict__ bn_w,
                                                        const float* __restrict__ bn_b, const float* __restrict__ bn_m,
                                                        const float* __restrict__ bn_v, float eps, int Cin, int Cp,
                                                        int KH, int KW, long s_co, long s_ci, long s_ky, long s_kx,
                                                        sp_t* __restrict__ wsp, const float* __restrict__ wscale) {
  const int co = blockIdx.y;
  const int K = KH * KW * Cp;
  float scale = 1.f;
  if (bn_w) scale = bn_w[co] / sqrtf(bn_v[co] + eps);
  const float up = 1.f / wscale[co];              // exact power of two (conv_rowscale_kernel ran before)
  const int grp = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (grp * 32 >= K) return;
  const int col = grp * 32 + (threadIdx.x & 31);
  const int tap = col / Cp, c = col - tap * Cp;
  float v = 0.f;
  if (c < Cin) v = (w[co * s_co + c * s_ci + (tap / KW) * s_ky + (tap % KW) * s_kx] * scale) * up;
  sp_store(wsp + (long)co * K, col, v, true);
}

// out = lateral + bilinear_x2(low), align_corners=True       (resnet_fpn.py:111-116: F.interpolate + add)
//   low [B, Hl, Wl, Cp], lateral / out [B, 2Hl, 2Wl, Cp], all SP.   One thread per (pixel, channel octet):
//   five pairs of 16-B loads (4 taps + lateral, hi and lo chunk each), 8 lerps, one pair of 16-B stores.
__global__ __launch_bounds__(256) void upsample_add_kernel(const sp_t* __restrict__ low, const sp_t* __restrict__ lat,
                                                           sp_t* __restrict__ out, int Hl, int Wl, int Cp, long nitems) {
  const long item = (long)blockIdx.x * 256 + threadIdx.x;
  if (item >= nitems) return;
  const int octs = Cp >> 3;
  const long pix = item / octs;
  const int oct = (int)(item - pix * octs);
  const int Ho = 2 * Hl, Wo = 2 * Wl;
  const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho);
  const long b = pix / ((long)Wo * Ho);
  // torch upsample_bilinear2d, align_corners=True: src = dst * (in - 1) / (out - 1)
  const float sy = Ho > 1 ? (float)(Hl - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wl - 1) / (float)(Wo - 1) : 0.f;
  const float fy = sy * (float)y, fx = sx * (float)x;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < Hl - 1 ? 1 : 0), x1 = x0 + (x0 < Wl - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const int off = sp_octet_off(oct);
  const sp_t* src[5] = {low + ((b * Hl + y0) * Wl + x0) * Cp + off, low + ((b * Hl + y0) * Wl + x1) * Cp + off,
                        low + ((b * Hl + y1) * Wl + x0) * Cp + off, low + ((b * Hl + y1) * Wl + x1) * Cp + off,
                        lat + pix * Cp + off};
  u32x4 hi[5], lo[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    hi[t] = *reinterpret_cast<const u32x4*>(src[t]);
    lo[t] = *reinterpret_cast<const u32x4*>(src[t] + 16);
  }
  float v[5][8];
#pragma unroll
  for (int t = 0; t < 5; ++t) sp_unpack8(hi[t], lo[t], v[t]);
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e)
    r[e] = v[4][e] + (hy * (hx * v[0][e] + lx * v[1][e]) + ly * (hx * v[2][e] + lx * v[3][e]));
  u32x4 oh, ol;
  sp_pack8(r, oh, ol);
  sp_t* o = out + pix * Cp + off;
  *reinterpret_cast<u32x4*>(o) = oh;
  *reinterpret_cast<u32x4*>(o + 16) = ol;
}

// Stem: nn.Conv2d(1, C0, 7, stride 2, pad 3, bias=False) + eval BatchNorm + ReLU (resnet_fpn.py:52-54,101)
// as a direct convolution on the vector units: K = 49 is far too short for the matrix pipeline and the
// single input channel would waste 31/32 of an SP group.  One workgroup computes a TY x TX patch of
// output pixels for all C0 channels: the (2TY+5) x (2TX+5) input window lives in LDS and is read as
// wave-wide broadcasts, each lane keeps the 49 folded weights of its channel in registers, and the
// result is written straight in the SP operand format (32 consecutive channels = one 128-B group).
//   grid (ceil(Wo/TX), ceil(Ho/TY), B), 256 threads = 2 pixel slots x (C0 = 128 channels)
constexpr int STEM_TY = 8, STEM_TX = 32;
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, long sxb, long sxh, long sxw,
                                                        int H, int W, const float* __restrict__ w, long s_co, long s_ky,
                                                        long s_kx, const float* __restrict__ bn_w,
                                                        const float* __restrict__ bn_b, const float* __restrict__ bn_m,
                                                        const float* __restrict__ bn_v, float eps, int C0, int Ho, int Wo,
                                                        sp_t* __restrict__ y) {
  constexpr int IH = 2 * STEM_TY + 5, IW = 2 * STEM_TX + 5;
  __shared__ float tile[IH][IW + 3];
  const int b = blockIdx.z, oy0 = blockIdx.y * STEM_TY, ox0 = blockIdx.x * STEM_TX;
  const float* xb = x + (long)b * sxb;
  for (int e = threadIdx.x; e < IH * IW; e += 256) {
    const int r = e / IW, c = e - r * IW;
    const int iy = 2 * oy0 - 3 + r, ix = 2 * ox0 - 3 + c;
    tile[r][c] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? xb[iy * sxh + ix * sxw] : 0.f;
  }
  const int ch = threadIdx.x & 127, slot = threadIdx.x >> 7;            // channel, pixel slot (0/1)
  float wr[49];
  float scale = 1.f, shift = 0.f;
  if (ch < C0) {
    if (bn_w) { scale = bn_w[ch] / sqrtf(bn_v[ch] + eps); shift = bn_b[ch] - bn_m[ch] * scale; }
#pragma unroll
    for (int t = 0; t < 49; ++t) wr[t] = w[ch * s_co + (t / 7) * s_ky + (t % 7) * s_kx] * scale;
  } else {
#pragma unroll
    for (int t = 0; t < 49; ++t) wr[t] = 0.f;
  }
  __syncthreads();
  const int Cp = (C0 + 31) / 32 * 32;
  // four horizontally adjacent output pixels at a time: their 7 x 13 input footprint is read once (91 LDS
  // broadcasts instead of 4 x 49) and feeds four accumulators
  for (int q = slot; q < STEM_TY * STEM_TX / 4; q += 2) {
    const int py = q / (STEM_TX / 4), px = (q - py * (STEM_TX / 4)) * 4;
    float acc[4] = {shift, shift, shift, shift};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      float row[13];
#pragma unroll
      for (int c = 0; c < 13; ++c) row[c] = tile[2 * py + ky][2 * px + c];
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += row[2 * e + kx] * wr[ky * 7 + kx];
    }
    const int oy = oy0 + py;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ox = ox0 + px + e;
      const float v = fmaxf(acc[e], 0.f);
      const bool ok = oy < Ho && ox < Wo && ch < Cp;
      const uint32_t word = sp_word(ch < C0 ? v : 0.f, ch & 1);
      if (ok) y[(((long)b * Ho + oy) * Wo + ox) * Cp + sp_index(ch)] = word;
    }
  }
}

// SP [rows, Cp] -> fp32 [rows, C]   (debug / hand-over helper)
__global__ void sp_to_f32_kernel(const sp_t* __restrict__ src, float* __restrict__ dst, long rows, int C, int Cp) {
  const long row = blockIdx.x;
  for (int c = threadIdx.x; c < Cp; c += blockDim.x) {
    const float v = sp_value(src[row * Cp + sp_index(c)], c & 1);
    if (c < C) dst[row * C + c] = v;
  }
}

}  // namespace

extern "C" size_t loftr_conv_workspace_bytes(int Cin, int Cout, int KH, int KW) {
  if (Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return 0;
  return align_up((size_t)Cout * KH * KW * ceil32(Cin) * 4, 256) + 2 * align_up((size_t)Cout * 4, 256) + 2048;
}

// Grid of a persistent kernel that holds one workgroup per CU: min(virtual workgroups, CUs), CUs rounded down to a
// multiple of the XCD count so that the grid stride keeps every workgroup on its XCD (debug switch "conv_persist_cap" = n >= 8:
// cap the grid at n workgroups -- tests: many tiles per workgroup).
static unsigned persistent_grid(unsigned nvirt) {
  const int cap = loftr_debug_value(LOFTR_DBG_CONV_PERSIST_CAP);
  const int forced = cap >= NUM_XCD ? cap / NUM_XCD * NUM_XCD : -1;      // -1: ask the device
  int cus = forced;
  if (cus < 0) {                                        // per CURRENT device (a process may drive several GPUs)
    static int per_dev[64];                             // 0 = not asked yet; benign race: every thread computes the same value
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nvirt;
    if (per_dev[dev] == 0) {
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return nvirt;
      per_dev[dev] = n / NUM_XCD * NUM_XCD;
    }
    cus = per_dev[dev];
  }
  return cus > 0 && nvirt > (unsigned)cus ? (unsigned)cus : nvirt;
}

// Second half of the tap-decomposed remainder channels (conv3x3_duo.h, Cfg<.., REM>): out[y][x][c0 + c] = act(sum over the taps (ky, kx) of
// P[(y + ky - 1, x + kx - 1)][ky * 3 + kx][c] + bias + residual), zero padding = taps outside the image skipped; writes the whole SP group of the
// columns c0 .. c0 + 31 (pad channels as zeros).  One thread per pixel: nine R-vectors of P (16-byte loads at R = 4; the neighbours' lines are
// shared through the caches), the residual's two 16-byte pieces, eight 16-byte stores = the pixel's full 128-byte line; taps in a fixed order.
template <int RT>     // RT = R when R == 4 (vector loads), 0: any R <= 7
__global__ __launch_bounds__(256) void conv_rem_gather_kernel(const float* __restrict__ pbuf, long npix, int H, int W, int R, int c0, int Coutp,
                                                              const float* __restrict__ bias, const sp_t* __restrict__ residual, int act,
                                                              sp_t* __restrict__ y_sp) {
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  const int x = (int)(pix % W), y = (int)((pix / W) % H);
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = 0.f;
  const float* pp = pbuf + pix * (9 * R);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xx = x + kx - 1;
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
        const float* q = pp + ((long)(ky - 1) * W + (kx - 1)) * (9 * R) + (ky * 3 + kx) * R;
        if (RT == 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(q);
          v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        } else {
#pragma unroll
          for (int c = 0; c < 7; ++c) if (c < R) v[c] += q[c];
        }
      }
    }
  sp_t* row = y_sp + pix * Coutp + c0;                 // the group: dwords 0 .. 15 = hi halves of the column pairs, 16 .. 31 = lo halves
  float rv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) rv[c] = 0.f;
  if (residual) {
    const sp_t* rr = residual + pix * Coutp + c0;
    const u32x4 hi = *reinterpret_cast<const u32x4*>(rr), lo = *reinterpret_cast<const u32x4*>(rr + 16);
    sp_unpack8(hi, lo, rv);
  }
  const float slope = act == 1 ? 0.f : act == 2 ? 0.01f : 1.f;
  float o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float t = v[c] + ((bias && c < R) ? bias[c0 + c] : 0.f) + rv[c];
    o[c] = c < R ? fmaxf(t, slope * t) : 0.f;
  }
  u32x4 oh, ol;
  sp_pack8(o, oh, ol);
  const u32x4 z = {0u, 0u, 0u, 0u};
  u32x4* dst = reinterpret_cast<u32x4*>(row);
  dst[0] = oh; dst[1] = z; dst[2] = z; dst[3] = z; dst[4] = ol; dst[5] = z; dst[6] = z; dst[7] = z;
}

// Folded weights live in a caller-owned buffer laid out [SP weights Cout x K][bias Cout][inverse row scales Cout][zero page 256 B]
// (loftr_conv_workspace_bytes): conv_prepare fills it, conv_run consumes it.
struct ConvPrepared { sp_t* wsp; float* bias; float* wscale; sp_t* zeros; };
static bool conv_prepared_layout(void* buf, size_t bytes, int Cin, int Cout, int KH, int KW, ConvPrepared& o) {
  WsAlloc wa(buf, bytes);
  o.wsp = wa.take<sp_t>((size_t)Cout * KH * KW * ceil32(Cin));
  o.bias = wa.take<float>(Cout);
  o.wscale = wa.take<float>(Cout);
  o.zeros = wa.take<sp_t>(64);
  return wa.ok();
}

static int conv_prepare(const float* weight, const long* weight_strides, int Cin, int Cout, int KH, int KW,
                        const float* bn_weight, const float* bn_bias, const float* bn_mean, const float* bn_var, float bn_eps,
                        void* buf, size_t bytes, hipStream_t st) {
  LOFTR_CHECK_ARG(weight && weight_strides && buf && Cin > 0 && Cout > 0 && KH > 0 && KW > 0);
  LOFTR_CHECK_ARG((bn_weight == nullptr) == (bn_bias == nullptr) && (bn_weight == nullptr) == (bn_mean == nullptr) &&
                  (bn_weight == nullptr) == (bn_var == nullptr));
  ConvPrepared pr;
  if (!conv_prepared_layout(buf, bytes, Cin, Cout, KH, KW, pr)) return LOFTR_ERR_WORKSPACE;
  const int Cp = ceil32(Cin), K = KH * KW * Cp;
  (void)hipMemsetAsync(pr.zeros, 0, 256, st);
  hipLaunchKernelGGL(conv_rowscale_kernel, dim3(Cout), dim3(256), 0, st, weight, bn_weight, bn_bias, bn_mean, bn_var, bn_eps,
                     Cin, KH, KW, weight_strides[0], weight_strides[1], weight_strides[2], weight_strides[3], pr.bias, pr.wscale);
  hipLaunchKernelGGL(conv_prep_kernel, dim3(ceil_div(K / 32, 8), Cout), dim3(256), 0, st, weight, bn_weight, bn_bias,
                     bn_mean, bn_var, bn_eps, Cin, Cp, KH, KW, weight_strides[0], weight_strides[1], weight_strides[2],
                     weight_strides[3], pr.wsp, pr.wscale);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// Output channels beyond 192 that the remainder form of the 3x3 kernel takes (0: not that shape): 224 padded columns, 1 .. 7 real ones in the
// last group (9 taps x R <= 64 columns of P)
static int conv_rem_channels(int Cout, int KH, int KW, int stride) {
  return (KH == 3 && KW == 3 && stride == 1 && ceil32(Cout) == 224 && Cout - 192 >= 1 && Cout - 192 <= 7) ? Cout - 192 : 0;
}
extern "C" size_t loftr_conv_scratch_bytes(int B, int H, int W, int Cout, int KH, int KW, int stride) {
  const int R = conv_rem_channels(Cout, KH, KW, stride);
  return (R > 0 && B > 0 && H > 0 && W > 0) ? (size_t)B * H * W * 9 * R * 4 : 0;
}

static int conv_run(const uint32_t* x_sp, int B, int H, int W, int Cin, const void* prepared, size_t prepared_bytes, int Cout,
                    int KH, int KW, int stride, int pad, int act, const uint32_t* residual_sp, const uint32_t* up_sp,
                    uint32_t* y_sp, float* y_f32, void* stream, const float* x_inv = nullptr, void* scratch = nullptr, size_t scratch_bytes = 0) {
  LOFTR_CHECK_ARG(x_sp && prepared && (y_sp || y_f32) && B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  const bool shared_gpu = (act & LOFTR_CONV_SHARED_GPU) != 0;      // see loftr_hip.h: no persistent workgroups
  act &= ~LOFTR_CONV_SHARED_GPU;
  LOFTR_CHECK_ARG(KH > 0 && KW > 0 && stride > 0 && pad >= 0 && act >= 0 && act <= 2);
  if (B == 0) return LOFTR_OK;
  hipStream_t st = (hipStream_t)stream;
  ConvGeom g;
  g.H = H; g.W = W; g.Cp = ceil32(Cin); g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
  g.Ho = (H + 2 * pad - KH) / stride + 1;
  g.Wo = (W + 2 * pad - KW) / stride + 1;
  if (g.Ho <= 0 || g.Wo <= 0 || H >= 32768 || W >= 32768) return LOFTR_ERR_UNSUPPORTED;
  if (up_sp && ((g.Ho & 1) || (g.Wo & 1) || KH != 1 || KW != 1 || stride != 1 || pad != 0 || !y_sp || y_f32 || residual_sp || act != 0))
    return LOFTR_ERR_UNSUPPORTED;                 // top-down step: 1x1 lateral conv, low map exactly [B, H/2, W/2, ceil32(Cout)]
  const long M = (long)B * g.Ho * g.Wo;
  if (M * (long)ceil32(Cout) >= (1L << 31) || (long)B * H * W * g.Cp >= (1L << 31)) return LOFTR_ERR_UNSUPPORTED;
  const int K = KH * KW * g.Cp;
  ConvPrepared pr;
  if (!conv_prepared_layout(const_cast<void*>(prepared), prepared_bytes, Cin, Cout, KH, KW, pr)) return LOFTR_ERR_WORKSPACE;
  sp_t* wsp = pr.wsp;
  float* bias = pr.bias;
  sp_t* zeros = pr.zeros;
  g.zeros = zeros;
  ConvArgs p;
  p.a = asrc_conv(x_sp, g);
  p.w = wsp; p.K = K; p.bias = bias; p.wscale = pr.wscale; p.x_inv = x_inv; p.residual = residual_sp; p.y_sp = y_sp; p.y_f32 = y_f32;
  p.M = (int)M; p.Cout = Cout; p.Coutp = ceil32(Cout); p.act = act;
  p.up = up_sp; p.Hl = g.Ho / 2; p.Wl = g.Wo / 2;
  p.sy = g.Ho > 1 ? (float)(p.Hl - 1) / (float)(g.Ho - 1) : 0.f;
  p.sx = g.Wo > 1 ? (float)(p.Wl - 1) / (float)(g.Wo - 1) : 0.f;
  if (loftr_debug_value(LOFTR_DBG_CONV_PATCH) && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !up_sp) {
    Conv3Args c;
    c.x = x_sp; c.B = B; c.H = H; c.W = W; c.Cp = g.Cp; c.Cin = Cin; c.w = wsp; c.K = K; c.bias = bias; c.wscale = pr.wscale; c.x_inv = x_inv; c.residual = residual_sp;
    c.y_sp = y_sp; c.y_f32 = y_f32; c.Cout = Cout; c.Coutp = ceil32(Cout); c.act = act; c.zeros = zeros;
    c.tiles_x = ceil_div(W, c3::TX); c.tiles_y = ceil_div(H, c3::TY); c.pbuf = nullptr;
    const bool duo = loftr_debug_value(LOFTR_DBG_CONV_DUO) != 0;
    const bool wide = duo && c.Coutp == 32 * 7;
    TimedLaunch tl(wide ? LOFTR_T_CONV3W : LOFTR_T_CONV3, st);
    // round 4 (conv3x3_duo.h): 128-column tiles on two 4-wave workgroups per CU; 192 / 224 columns on one 8-wave workgroup whose wave
    // pairs split the column tiles (tools/gpu/r4_octo.sh)
    if (duo && c.Coutp % 128 == 0) {
      using CF = c3d::Cfg<4, 2, 4>;
      c.tiles_y = ceil_div(H, CF::TY);
      hipLaunchKernelGGL((conv3x3_duo_kernel<CF>), dim3(xcd_grid(B * c.tiles_x * c.tiles_y, c.Coutp / 128)), dim3(256), 0, st, c);
    } else if (duo && c.Coutp == 192) {        // six column tiles (3 + 3 per wave pair)
      using CF = c3d::Cfg<6, 2, 4, 8, 2>;
      c.tiles_y = ceil_div(H, CF::TY);
      hipLaunchKernelGGL((conv3x3_duo_kernel<CF>), dim3(xcd_grid(B * c.tiles_x * c.tiles_y, 1)), dim3(512), 0, st, c);
    } else if (wide && conv_rem_channels(Cout, KH, KW, stride) > 0 && !y_f32 && loftr_debug_value(LOFTR_DBG_CONV_REM) &&
               scratch && scratch_bytes >= (size_t)B * H * W * 9 * conv_rem_channels(Cout, KH, KW, stride) * 4) {
      // round 6: 192 columns + the channels beyond them as a tap-decomposed product at the centre-tap steps (conv3x3_duo.h), then the gather
      using CF = c3d::Cfg<6, 2, 4, 8, 2, true>;
      c.tiles_y = ceil_div(H, CF::TY);
      c.pbuf = reinterpret_cast<float*>(scratch);
      hipLaunchKernelGGL((conv3x3_duo_kernel<CF>), dim3(xcd_grid(B * c.tiles_x * c.tiles_y, 1)), dim3(512), 0, st, c);
      LOFTR_CHECK_LAUNCH();
      const long npix = (long)B * H * W;
      if (Cout - 192 == 4)
        hipLaunchKernelGGL(conv_rem_gather_kernel<4>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, c.pbuf, npix, H, W, 4, 192, c.Coutp,
                           bias, residual_sp, act, y_sp);
      else
        hipLaunchKernelGGL(conv_rem_gather_kernel<0>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, c.pbuf, npix, H, W, Cout - 192, 192, c.Coutp,
                           bias, residual_sp, act, y_sp);
    } else if (wide) {
      using CF = c3d::Cfg<7, 2, 4, 8, 2>;
      c.tiles_y = ceil_div(H, CF::TY);
      hipLaunchKernelGGL((conv3x3_duo_kernel<CF>), dim3(xcd_grid(B * c.tiles_x * c.tiles_y, 1)), dim3(512), 0, st, c);
    } else                        // any other Cout (not a multiple of 128, not 192 / 224): the generic patch kernel, column tiles of 128
      hipLaunchKernelGGL(conv3x3_kernel, dim3(shared_gpu ? xcd_grid(B * c.tiles_x * c.tiles_y, ceil_div(c.Coutp, 128))
                                                         : persistent_grid(xcd_grid(B * c.tiles_x * c.tiles_y, ceil_div(c.Coutp, 128)))),
                         dim3(512), 0, st, c);
    LOFTR_CHECK_LAUNCH();
    return LOFTR_OK;
  }
  {
    TimedLaunch tl(LOFTR_T_CONV, st);
    if (up_sp)
      hipLaunchKernelGGL((conv_kernel<CfgD, true>), dim3(xcd_grid(ceil_div(p.M, CfgD::BM), ceil_div(p.Coutp, CfgD::BN))),
                         dim3(CfgD::THREADS), 0, st, p);
    else
      hipLaunchKernelGGL((conv_kernel<CfgD, false>), dim3(xcd_grid(ceil_div(p.M, CfgD::BM), ceil_div(p.Coutp, CfgD::BN))),
                         dim3(CfgD::THREADS), 0, st, p);
  }
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

#ifdef LOFTR_CONV_PROBE
// probe builds only (python -m loftr_amd.build --variant probe -DLOFTR_CONV_PROBE): where conv3x3_duo_kernel drops its time stamps
extern "C" int loftr_conv_probe_buffer(void* buf) {
  long long* b = (long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_conv_probe), &b, sizeof(b)) == hipSuccess ? LOFTR_OK : LOFTR_ERR_LAUNCH;
}
#endif

extern "C" int loftr_conv_prepare(const float* weight, const long* weight_strides, int Cin, int Cout, int KH, int KW,
                                  const float* bn_weight, const float* bn_bias, const float* bn_mean, const float* bn_var,
                                  float bn_eps, void* prepared, size_t prepared_bytes, void* stream) {
  return conv_prepare(weight, weight_strides, Cin, Cout, KH, KW, bn_weight, bn_bias, bn_mean, bn_var, bn_eps, prepared,
                      prepared_bytes, (hipStream_t)stream);
}

extern "C" int loftr_conv_bn_act_prepared(const uint32_t* x_sp, int B, int H, int W, int Cin, const void* prepared,
                                          size_t prepared_bytes, int Cout, int KH, int KW, int stride, int pad, int act,
                                          const uint32_t* residual_sp, const uint32_t* low_sp, uint32_t* y_sp, float* y_f32,
                                          const float* x_inv_scale, void* stream) {
  return conv_run(x_sp, B, H, W, Cin, prepared, prepared_bytes, Cout, KH, KW, stride, pad, act, residual_sp, low_sp, y_sp, y_f32,
                  stream, x_inv_scale);
}

extern "C" int loftr_conv_bn_act_prepared_scratch(const uint32_t* x_sp, int B, int H, int W, int Cin, const void* prepared,
                                                  size_t prepared_bytes, int Cout, int KH, int KW, int stride, int pad, int act,
                                                  const uint32_t* residual_sp, const uint32_t* low_sp, uint32_t* y_sp, float* y_f32,
                                                  const float* x_inv_scale, void* scratch, size_t scratch_bytes, void* stream) {
  return conv_run(x_sp, B, H, W, Cin, prepared, prepared_bytes, Cout, KH, KW, stride, pad, act, residual_sp, low_sp, y_sp, y_f32,
                  stream, x_inv_scale, scratch, scratch_bytes);
}

extern "C" int loftr_conv_bn_act(const uint32_t* x_sp, int B, int H, int W, int Cin, const float* weight,
                                 const long* weight_strides, int Cout, int KH, int KW, int stride, int pad, const float* bn_weight, const float* bn_bias,
                                 const float* bn_mean, const float* bn_var, float bn_eps, int act,
                                 const uint32_t* residual_sp, uint32_t* y_sp, float* y_f32, void* ws, size_t ws_bytes,
                                 const float* x_inv_scale, void* stream) {
  LOFTR_CHECK_ARG(x_sp && (y_sp || y_f32) && B >= 0 && H > 0 && W > 0);
  if (B == 0) return LOFTR_OK;
  const int rc = conv_prepare(weight, weight_strides, Cin, Cout, KH, KW, bn_weight, bn_bias, bn_mean, bn_var, bn_eps, ws, ws_bytes,
                              (hipStream_t)stream);
  if (rc != LOFTR_OK) return rc;
  return conv_run(x_sp, B, H, W, Cin, ws, ws_bytes, Cout, KH, KW, stride, pad, act, residual_sp, nullptr, y_sp, y_f32, stream, x_inv_scale);
}

extern "C" int loftr_conv1x1_upsample_add(const uint32_t* x_sp, int B, int H, int W, int Cin, const float* weight,
                                          const long* weight_strides, int Cout, const uint32_t* low_sp, uint32_t* y_sp,
                                          void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(x_sp && low_sp && y_sp && B >= 0 && H > 0 && W > 0);
  if (B == 0) return LOFTR_OK;
  const int rc = conv_prepare(weight, weight_strides, Cin, Cout, 1, 1, nullptr, nullptr, nullptr, nullptr, 0.f, ws, ws_bytes,
                              (hipStream_t)stream);
  if (rc != LOFTR_OK) return rc;
  return conv_run(x_sp, B, H, W, Cin, ws, ws_bytes, Cout, 1, 1, 1, 0, 0, nullptr, low_sp, y_sp, nullptr, stream);
}

extern "C" int loftr_stem_conv_bn_relu(const float* x, const long* x_strides, int B, int H, int W, const float* weight,
                                       const long* weight_strides, int C0, const float* bn_weight, const float* bn_bias,
                                       const float* bn_mean, const float* bn_var, float bn_eps, uint32_t* y_sp,
                                       void* stream) {
  LOFTR_CHECK_ARG(x && x_strides && weight && weight_strides && y_sp && B >= 0 && H > 0 && W > 0 && C0 > 0);
  LOFTR_CHECK_ARG((bn_weight == nullptr) == (bn_bias == nullptr) && (bn_weight == nullptr) == (bn_mean == nullptr) &&
                  (bn_weight == nullptr) == (bn_var == nullptr));
  if (C0 > 128) return LOFTR_ERR_UNSUPPORTED;                 // one lane per channel, 128 channel lanes per workgroup
  if (B == 0) return LOFTR_OK;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  hipLaunchKernelGGL(stem_conv_kernel, dim3(ceil_div(Wo, STEM_TX), ceil_div(Ho, STEM_TY), B), dim3(256), 0,
                     (hipStream_t)stream, x, x_strides[0], x_strides[2], x_strides[3], H, W, weight, weight_strides[0],
                     weight_strides[2], weight_strides[3], bn_weight, bn_bias, bn_mean, bn_var, bn_eps, C0, Ho, Wo, y_sp);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

extern "C" int loftr_upsample2x_add(const uint32_t* low_sp, const uint32_t* lateral_sp, uint32_t* out_sp, int B, int Hl,
                                    int Wl, int C, void* stream) {
  LOFTR_CHECK_ARG(low_sp && lateral_sp && out_sp && B >= 0 && Hl > 0 && Wl > 0 && C > 0);
  if (B == 0) return LOFTR_OK;
  const int Cp = ceil32(C);
  const long pixels = (long)B * 4 * Hl * Wl;
  if (pixels >= (1L << 31)) return LOFTR_ERR_UNSUPPORTED;
  const long nitems = pixels * (Cp / 8);
  if ((nitems + 255) / 256 >= (1L << 31)) return LOFTR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(upsample_add_kernel, dim3((unsigned)((nitems + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     low_sp, lateral_sp, out_sp, Hl, Wl, Cp, nitems);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

extern "C" int loftr_sp_from_f32(const float* src, uint32_t* dst_sp, long rows, int C, void* stream) {
  LOFTR_CHECK_ARG(src && dst_sp && rows >= 0 && C > 0);
  return launch_sp_convert1(src, C, dst_sp, rows, C, (hipStream_t)stream);
}

extern "C" int loftr_sp_from_f32_scaled(const float* src, uint32_t* dst_sp, long rows, int C, float* inv_scale_out, void* stream) {
  LOFTR_CHECK_ARG(src && dst_sp && inv_scale_out && rows >= 0 && C > 0);
  return launch_sp_convert1(src, C, dst_sp, rows, C, (hipStream_t)stream, inv_scale_out);
}

extern "C" int loftr_sp_to_f32(const uint32_t* src_sp, float* dst, long rows, int C, void* stream) {
  LOFTR_CHECK_ARG(src_sp && dst && rows >= 0 && C > 0);
  if (rows == 0) return LOFTR_OK;
  const int Cp = ceil32(C);
  hipLaunchKernelGGL(sp_to_f32_kernel, dim3((unsigned)rows), dim3(Cp > 256 ? 256 : Cp), 0, (hipStream_t)stream, src_sp,
                     dst, rows, C, Cp);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
