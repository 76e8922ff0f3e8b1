// 3x3 / stride 1 / pad 1 convolutions, round 4: TWO workgroups per CU.
//   reference: src/loftr/backbone/resnet_fpn.py:5-40 (conv3x3 + BN + ReLU / residual of a BasicBlock), :64-83 (FPN heads)
//
// What round 3's profile said (tools/micro/conv_layers.py, per-tile time = a + b * k-tiles fitted over layers that run the same
// kernel): the k-loop of conv3x3_kernel runs the matrix pipe at ~90 % (b = 1.03 us per k-tile against 0.93 at 100 %), but every
// 8 x 32-pixel tile pays a = 20 us (30 us with a residual branch) of epilogue -- residual loads, ~25 VALU instructions per value
// for scale / bias / activation / the fp32 -> (hi, lo) split, 64 scalar-width stores per wave and their drain -- and of pipeline
// fill, during which the CU's matrix pipe idles: one 150 KB workgroup per CU means nobody else is there to use it.  That is 47 %
// of the time of the four layer1 convolutions (36 k-tiles) and 30 % of the 72-k-tile layers.
//
// Here a workgroup is FOUR waves (one per SIMD) and needs < 80 KB of LDS, so two workgroups share a CU and run out of phase: one's
// epilogue / prologue overlaps the other's k-loop.  To fit:
//   * the input patch is double-buffered per HALF channel group (one 16-wide k-step: 64 B per pixel = hi chunks {2h, 2h+1}, lo
//     chunks {4+2h, 5+2h} of the pixel's 128-B SP group) instead of per group: 2 x 22 KB for the (8+2) x (32+2) patch;
//   * a weight stage is one tap x one k-step: NT*32 rows x 64 B (8 KB at 128 columns), ring of NB stages, NB-1 steps ahead;
//   * k order: channel group, half, tap column kx, tap row ky innermost (consecutive ky share patch rows, kept in registers).
//   * a wave owns RW output rows x NT column tiles (2 x 4 at 128 columns: 128 accumulator registers, 8 B-fragment + 2.7 A-fragment
//     reads per 24 MFMAs -- 0.44 reads per MFMA against 0.56 in conv3x3_kernel); one barrier per step (3 * RW * NT MFMAs).
// LDS rows are 64 B (4 chunks of 16 B: hi k 0-7, hi k 8-15, lo k 0-7, lo k 8-15); chunk c of row r sits at slot c ^ ((r >> 2) & 3):
// any 16 consecutive rows read through one chunk index cover the 64 banks exactly once (ds_read_b128 serves 16 lanes per pass).
// The swizzle is applied on the GLOBAL side of the DMA (global_load_lds writes lane l's 16 B at base + 16 l).
#pragma once

#ifdef LOFTR_CONV_PROBE
__device__ long long* g_conv_probe = nullptr;   // set by loftr_conv_probe_buffer (probe builds only)
#endif
// timing probes (wrong results): 0 = no per-step barrier / no DMA after the prologue / fragment reads only before the loop / no epilogue
#ifndef C3D_PROBE_BARRIER
#define C3D_PROBE_BARRIER 1
#endif
#ifndef C3D_PROBE_DMA
#define C3D_PROBE_DMA 1
#endif
#ifndef C3D_PROBE_READS
#define C3D_PROBE_READS 1
#endif
#ifndef C3D_PROBE_EPI
#define C3D_PROBE_EPI 1
#endif
#ifndef C3D_BDIST
#define C3D_BDIST 1               // column tiles the B-fragment reads run ahead of their MFMAs (2: +8 registers)
#endif

namespace c3d {
constexpr int TX = 32, PW = TX + 2;
// NT column tiles per workgroup tile, RW output rows per wave, NB weight ring stages, WAVES waves of which WN share a row pair and
// split the NT column tiles between them (wave = wn * WM + wr: waves w and w + 4 of an 8-wave workgroup sit on the same SIMD).
//   Cfg<4, 2, 4, 4, 1>  "duo"   256 px x 128 columns, 79 KB, two workgroups per CU
//   Cfg<7, 1, 3, 4, 1>  "duo"   128 px x 224 columns, 71 KB, two workgroups per CU
//   Cfg<4, 2, 4, 8, 1>  "octo"  512 px x 128 columns, 111 KB, one workgroup per CU: HALF the weight DMA per MFMA
//   Cfg<7, 2, 4, 8, 2>  "octo"  256 px x 224 columns (a wave: 2 rows x 4 or 3 tiles), 101 KB: half the weight DMA of the duo form,
//                       whose 15 KB per 21 MFMAs and wave exceed what the global -> LDS path delivers (~14 B / clk / CU)
//   Cfg<6, 2, 4, 8, 2, true>  "octo + remainder" (round 6): 256 px x 192 columns + the 1 .. 7 output channels beyond 192 (Cout = 196) as a
//                       TAP-DECOMPOSED product.  The 7th column tile of Cfg<7, ..> spends 32 columns x K = 9 Cin on 4 real channels; but
//                       conv = sum over the taps of shifted 1x1 convolutions, so P[pixel][tap][c] = sum_cin x[pixel][cin] w[c][tap][cin] on the
//                       UNSHIFTED pixels is ONE product with N = 9 taps x 4 channels = 36 columns and K = Cin: two column tiles (one per wave
//                       pair) at the centre-tap steps only, whose A fragments ARE the unshifted pixels -- 56 instead of 63 tile-steps per nine
//                       taps.  P goes to a scratch buffer (fp32, 36 per pixel); conv_rem_gather_kernel adds the nine shifted P's, bias,
//                       residual, activation and writes the SP group of the channels 192 .. 223.  Its B rows are a VIEW of the prepared
//                       filter: row (tap, c) = filter row 192 + c, columns tap * Cp .. (no extra weight preparation).
template <int NT_, int RW_, int NB_, int WAVES_ = 4, int WN_ = 1, bool REM_ = false>
struct Cfg {
  static constexpr int NT = NT_, RW = RW_, NB = NB_, LA = NB_ - 1;               // LA: weight stages in flight ahead of the running step
  static constexpr bool REM = REM_;
  static constexpr int WAVES = WAVES_, WN = WN_, WM = WAVES_ / WN_, NJ = (NT_ + WN_ - 1) / WN_;   // NJ: column tiles per wave (the last split may own NJ - 1)
  static constexpr int TY = WM * RW, PH = TY + 2, PROWS = PW * PH;
  static constexpr int PSLOTS = (PROWS + 15) / 16, PQ = (PSLOTS + WAVES - 1) / WAVES;   // 1 KB DMA slots (16 rows) of a patch half; per wave
  static constexpr int PHALF_BYTES = PSLOTS * 1024;
  static constexpr int BROWS = NT * 32 + (REM_ ? 64 : 0), BSLOTS = BROWS / 16, BQ = (BSLOTS + WAVES - 1) / WAVES;   // REM: + 2 column tiles of P rows (centre-tap steps only)
  static constexpr int BSTAGE_BYTES = BSLOTS * 1024;
  static constexpr int LDS_BYTES = 2 * PHALF_BYTES + NB * BSTAGE_BYTES + 1024;    // + 1 KB scratch: destination of the unused DMA slots
  static constexpr int WG_PER_CU = WAVES == 4 ? 2 : 1;
  static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS");
  static_assert(RW <= 2, "patch rows are overwritten in place: row j of the next tap column while rows 2 .. RW+1 are in use");
  static_assert(LA >= 2 && LA <= 3, "vmcnt bookkeeping below");
  static_assert(WN == 1 || NT - (WN - 1) * NJ >= NJ - 1, "the last column split owns NJ or NJ - 1 tiles");
  static_assert(!REM_ || (WN_ == 2 && NT_ % 2 == 0), "remainder form: two wave splits, one P tile each, equal main shares");
};
// Issue order of one step (see C3D_STEP): group J = 3 RW MFMAs with R reads slotted one behind each of its first MFMAs.
template <int R, int M>
__device__ __forceinline__ void pin_group() {
  if constexpr (R > 0 && M > 0) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    pin_group<R - 1, M - 1>();
  } else if constexpr (R > 0) {
    __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
  } else if constexpr (M > 0) {
    __builtin_amdgcn_sched_group_barrier(0x008, M, 0);
  }
}
template <int J, int NT, int RW, int KY>
__device__ __forceinline__ void pin_groups() {
  if constexpr (J < NT) {
    pin_group<2 + (J < (KY < 2 ? 1 : RW) ? 2 : 0), 3 * RW>();
    pin_groups<J + 1, NT, RW, KY>();
  }
}
}  // namespace c3d

template <typename CF>
__global__ __launch_bounds__(CF::WAVES * 64, CF::WG_PER_CU) void conv3x3_duo_kernel(Conv3Args p) {
  using namespace c3d;
  constexpr int NT = CF::NT, RW = CF::RW, NB = CF::NB, LA = CF::LA, PQ = CF::PQ, BQ = CF::BQ;
  constexpr int WAVES = CF::WAVES, WM = CF::WM, NJ = CF::NJ;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  __shared__ __attribute__((aligned(16))) char lds[CF::LDS_BYTES];
  char* const patch_base = lds;
  char* const bring_base = lds + 2 * CF::PHALF_BYTES;
  char* const scratch = lds + 2 * CF::PHALF_BYTES + NB * CF::BSTAGE_BYTES;

#ifdef LOFTR_CONV_PROBE
  const long long pt0__ = wall_clock64();
#endif
  int tm, tn;
  constexpr bool REM = CF::REM;
  constexpr int NJA = NJ + (REM ? 1 : 0);                        // accumulator tiles per wave: its main column tiles + (REM) one tile of P
  const int tiles_m = p.B * p.tiles_y * p.tiles_x, tiles_n = REM ? 1 : ceil_div(p.Coutp, NT * 32);
  if (!xcd_tile(tiles_m, tiles_n, tm, tn)) return;
  const int b = tm / (p.tiles_y * p.tiles_x), trem = tm - b * (p.tiles_y * p.tiles_x);
  const int y0 = (trem / p.tiles_x) * CF::TY, x0 = (trem % p.tiles_x) * TX, n0 = tn * (NT * 32);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, tx = lane & 31;
  const int wr = wave % WM, wn = wave / WM;                     // this wave's row pair and column split
  const int jc0 = wn * NJ;                                      // its first column tile
  const int nj = min(NJ, NT - jc0);                             // ... and how many it owns (wave-uniform)
  const int drow = lane >> 2, dpos = lane & 3;                  // DMA: row inside a 16-row slot, 16-B position inside the 64-B row

  // ---- DMA source offsets (dwords): the chunk a lane fetches is fixed by (row, position); half / group / tap are added at issue
  int poff[PQ];                                                 // -1: outside the image / unused slot -> zero page
#pragma unroll
  for (int q = 0; q < PQ; ++q) {
    const int s_ = q * WAVES + wave, r = s_ * 16 + drow;
    const int py = r / PW, px = r - py * PW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const int c = dpos ^ ((r >> 2) & 3);
    const bool in = s_ < CF::PSLOTS && r < CF::PROWS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    poff[q] = in ? ((b * p.H + gy) * p.W + gx) * p.Cp + (((c & 1) + ((c >> 1) << 2)) << 2) : -1;
  }
  int boff[BQ];
#pragma unroll
  for (int q = 0; q < BQ; ++q) {
    const int r = (q * WAVES + wave) * 16 + drow;
    const int c = dpos ^ ((r >> 2) & 3);
    boff[q] = min(n0 + r, p.Cout - 1) * p.K + (((c & 1) + ((c >> 1) << 2)) << 2);      // rows >= Cout: clamped copies (never stored)
    if (REM && r >= NT * 32) {
      // P row pc = (tap, c): filter row NT * 32 + c, columns tap * Cp + (group, half) -- independent of the step's tap; rows beyond 9 R: zero page
      const int pc = r - NT * 32, R = p.Cout - NT * 32;
      boff[q] = pc < 9 * R ? (NT * 32 + pc % R) * p.K + (pc / R) * p.Cp + (((c & 1) + ((c >> 1) << 2)) << 2) : -1;
    }
  }
  const int gpt = p.Cp >> 5;
  // channels >= Cin of the last group are zero padding (activations AND folded weights): when they fill its whole second k-step
  // (Cin = 196 -> 192..207 | 208..223) that half is skipped -- exact
  const int nhalf = 2 * gpt - (p.Cin <= (gpt - 1) * 32 + 16 ? 1 : 0), ns = nhalf * 9;

  // (base pointers laundered per issue: otherwise the 64-bit DMA addresses are hoisted out of the loop and spilled)
#define C3D_ISSUE_PATCH(q_)                                                                                 \
  {                                                                                                         \
    const sp_t* xb__ = p.x;                                                                                 \
    asm volatile("" : "+s"(xb__));                                                                          \
    const int ko__ = ((q_) >> 1) * 32 + ((q_) & 1) * 8;                                                     \
    char* dst__ = patch_base + ((q_) & 1) * CF::PHALF_BYTES;                                                \
    _Pragma("unroll") for (int q = 0; q < PQ; ++q) {                                                        \
      const sp_t* g__ = poff[q] >= 0 ? xb__ + (poff[q] + ko__) : p.zeros;                                   \
      char* d__ = (q * WAVES + wave < CF::PSLOTS) ? dst__ + (q * WAVES + wave) * 1024 : scratch;            \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)g__, (lds_ptr_t)d__, 16, 0, 0);                           \
    }                                                                                                       \
  }
  // weight stage of step (half q_, index i_ = kx * 3 + ky inside the half): tap ky * 3 + kx
#define C3D_ISSUE_B(q_, i_, stage_)                                                                         \
  {                                                                                                         \
    const sp_t* wb__ = p.w;                                                                                 \
    asm volatile("" : "+s"(wb__));                                                                          \
    const int kx__ = (i_) / 3, ky__ = (i_) - kx__ * 3;                                                      \
    const int ko__ = (ky__ * 3 + kx__) * p.Cp + ((q_) >> 1) * 32 + ((q_) & 1) * 8;                          \
    char* dst__ = bring_base + (stage_) * CF::BSTAGE_BYTES;                                                 \
    const int kr__ = ((q_) >> 1) * 32 + ((q_) & 1) * 8;          /* P rows: the (group, half) offset only */  \
    _Pragma("unroll") for (int q = 0; q < BQ; ++q) {                                                        \
      const bool rem__ = REM && (q * WAVES + wave) * 16 >= NT * 32;        /* wave-uniform */                \
      if (rem__ && (i_) != 4) {      /* P rows are used at the centre tap only: elsewhere a 4-byte dummy keeps the per-stage DMA count (vmcnt bookkeeping) */ \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)p.zeros, (lds_ptr_t)scratch, 4, 0, 0);                  \
      } else {                                                                                              \
        char* d__ = (q * WAVES + wave < CF::BSLOTS) ? dst__ + (q * WAVES + wave) * 1024 : scratch;          \
        const sp_t* g__ = rem__ ? (boff[q] >= 0 ? wb__ + (boff[q] + kr__) : p.zeros) : wb__ + (boff[q] + ko__); \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g__, (lds_ptr_t)d__, 16, 0, 0);                         \
      }                                                                                                     \
    }                                                                                                       \
  }

  // ---- accumulators start at the residual branch (resnet_fpn.py:37: x + y), not at zero: the epilogue computes act(acc * wsc + bias)
  // with wsc an exact power of two per column, so acc0 = residual / wsc puts the residual in exactly.  Its loads are issued HERE, ahead
  // of the DMA prologue (they retire first: no vmcnt bookkeeping) and land during the pipeline fill -- in round 3's epilogues they were
  // 8 dependent load -> use rounds per tile while the matrix pipe waited.
  const bool odd = lane & 1;
  const int Ho = p.H, Wo = p.W;
  const float xinv = p.x_inv ? *p.x_inv : 1.f;
  // wave-uniform: every pixel of the tile inside the image -> no predication of the pixel index in the fast paths (columns beyond Cout
  // are handled by clamped table reads and a select: the padded SP row always exists)
  const bool full = y0 + CF::TY <= Ho && x0 + TX <= Wo;
  const int nc0 = n0 + jc0 * 32;                                // first column of this wave
  const unsigned lane_sp = (unsigned)(4 * g * p.Coutp + (odd ? 16 : 0) + (tx >> 1));     // this lane's dword inside (pixel x0 + 4 g, group of the wave's column tile 0)
  f32x16 acc[RW][NJA];
  if (p.residual) {
    if (full) {
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const sp_t* rp = p.residual + ((size_t)((b * Ho + y0 + wr * RW + i) * Wo + x0) * p.Coutp + nc0);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (j < nj) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[i][j][r] = __uint_as_float((rp + (size_t)(((r & 3) + 8 * (r >> 2)) * p.Coutp + j * 32))[lane_sp]);
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int y = y0 + wr * RW + i;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = nc0 + j * 32 + tx;
          const int spc = (col & ~31) + (odd ? 16 : 0) + ((col & 31) >> 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const bool ok = y < Ho && x < Wo && col < p.Coutp && j < nj;
            acc[i][j][r] = __uint_as_float(ok ? p.residual[(unsigned)(((b * Ho + y) * Wo + x) * p.Coutp + spc)] : 0u);
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  int bbase[NJA];                                              // byte offset of this lane's hi fragment of the wave's column tile j inside a stage
#pragma unroll
  for (int j = 0; j < NJA; ++j) {
    const int br = (REM && j == NJ) ? (NT + wn) * 32 + tx : min(jc0 + j, NT - 1) * 32 + tx;       // (REM: the wave pair's tile of P rows)
    bbase[j] = br * 64 + ((g ^ ((br >> 2) & 3)) << 4);
  }
  if (REM) {
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][NJ][r] = 0.f;
  }
  // patch pixel of this lane's output pixel in the wave's patch row jr at tap column kx: (wr * RW + jr) * PW + kx + tx
#define C3D_LOAD_ROW(jr_, sP_, kx_)                                                                         \
  {                                                                                                         \
    const int pr__ = (wr * RW + (jr_)) * PW + (kx_) + tx;                                                 \
    const int ab__ = pr__ * 64 + ((g ^ ((pr__ >> 2) & 3)) << 4);                                            \
    fh[jr_] = *reinterpret_cast<const h16x8*>((sP_) + ab__);                                                \
    fl[jr_] = *reinterpret_cast<const h16x8*>((sP_) + (ab__ ^ 32));                                         \
  }
#define C3D_LOAD_B(h_, l_, sB_, j_)                                                                         \
  {                                                                                                         \
    h_ = *reinterpret_cast<const h16x8*>((sB_) + bbase[j_]);                                                \
    l_ = *reinterpret_cast<const h16x8*>((sB_) + (bbase[j_] ^ 32));                                         \
  }

  // ---- prologue: patch half 0, weight stages 0 .. LA-1 (ns >= 9 > LA)
  C3D_ISSUE_PATCH(0);
#pragma unroll
  for (int s = 0; s < LA; ++s) C3D_ISSUE_B(0, s, s);
  int q3 = 0, i3 = LA;                                         // (half, index) of the step whose weights are issued next, LA ahead
  int stage = 0;                                               // ring stage of the running step
  bool patch_m1 = false, patch_m2 = false;                     // a patch half was issued one / two steps ago
  h16x8 fh[RW + 2], fl[RW + 2];                                // A fragments of the wave's patch rows at the running tap column
  h16x8 bh0, bl0;                                              // B fragment of column tile 0 of the running step
  [[maybe_unused]] h16x8 bh1 = {}, bl1 = {};                   // C3D_BDIST 2: ... and of column tile 1
  // pipeline fill: patch half 0 and weight stage 0 have landed once only stages 1 .. LA-1 are in flight
  LOFTR_WAITCNT_VM((LA - 1) * BQ);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int jr = 0; jr < RW; ++jr) C3D_LOAD_ROW(jr, patch_base, 0);
  C3D_LOAD_B(bh0, bl0, bring_base, 0);
  if (C3D_BDIST == 2) { C3D_LOAD_B(bh1, bl1, bring_base, (NJ > 1 ? 1 : 0)); }

  // One step = one tap x one 16-wide k-step.  Its A rows and its first B fragment are already in registers (read during the
  // previous step); the barrier of step s guarantees that weight stage s + 1 (and the patch half it may open) has landed.
#define C3D_STEP(KX, KY, NJ_)                                                                                  \
  {                                                                                                         \
    {   /* weight stage s + 1 has landed once only what was issued after it is outstanding: stage s + 2 (LA = 3) and a patch \
           half issued in one of the last LA - 1 steps (loads retire in order) */                           \
      const bool w2__ = LA == 3 && s + 2 < ns;                                                              \
      const bool pp__ = patch_m1 || (LA == 3 && patch_m2);                                                  \
      if (w2__ && pp__) LOFTR_WAITCNT_VM(BQ + PQ);                                                          \
      else if (pp__) LOFTR_WAITCNT_VM(PQ);                                                                  \
      else if (w2__) LOFTR_WAITCNT_VM(BQ);                                                                  \
      else LOFTR_WAITCNT_VM(0);                                                                             \
    }                                                                                                       \
    if (C3D_PROBE_BARRIER) __builtin_amdgcn_s_barrier();                                                    \
    if (C3D_PROBE_DMA && s + LA < ns) C3D_ISSUE_B(q3, i3, stage + LA >= NB ? stage + LA - NB : stage + LA); \
    if (++i3 == 9) { i3 = 0; ++q3; }                                                                        \
    patch_m2 = patch_m1;                                                                                    \
    patch_m1 = false;                                                                                       \
    if (C3D_PROBE_DMA && (KX) == 0 && (KY) == 0 && hq + 1 < nhalf) { C3D_ISSUE_PATCH(hq + 1); patch_m1 = true; } \
    const char* sB__ = bring_base + stage * CF::BSTAGE_BYTES;                                               \
    const int nstage__ = stage + 1 == NB ? 0 : stage + 1;                                                   \
    /* Issue order, pinned below: group j = column tile j's 3 RW MFMAs; behind its first MFMAs, one read each, go the B fragment of \
       tile j + 1 (tile 0 of the next step after the last) and, in groups 0 / 1, the patch rows the NEXT step needs (after the last \
       step: harmless reads of stale LDS).  Rows are overwritten in place: the running step reads rows KY .. KY + RW - 1. */ \
    h16x8 ch__ = bh0, cl__ = bl0;                                                                           \
    [[maybe_unused]] h16x8 dh__ = bh1, dl__ = bl1;             /* C3D_BDIST 2: the fragment after the running one */ \
    _Pragma("unroll") for (int j = 0; j < (NJ_); ++j) {                                                     \
      h16x8 nh__, nl__;                                                                                     \
      if (!C3D_PROBE_READS) { nh__ = ch__; nl__ = cl__; }                                                   \
      else if (j + C3D_BDIST < (NJ_)) { C3D_LOAD_B(nh__, nl__, sB__, j + C3D_BDIST); }                      \
      else { C3D_LOAD_B(nh__, nl__, bring_base + nstage__ * CF::BSTAGE_BYTES, (j + C3D_BDIST - (NJ_)) % (NJ_)); } \
      if (C3D_PROBE_READS && j < ((KY) < 2 ? 1 : RW)) {                                                     \
        if ((KY) < 2) { C3D_LOAD_ROW((KY) + RW, sP, KX); }                                                  \
        else if ((KX) < 2) { C3D_LOAD_ROW(j, sP, (KX) + 1); }                                               \
        else { C3D_LOAD_ROW(j, sPn, 0); }                                                                   \
      }                                                                                                     \
      _Pragma("unroll") for (int i = 0; i < RW; ++i)                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[(KY) + i], ch__, acc[i][j], 0, 0, 0);         \
      _Pragma("unroll") for (int i = 0; i < RW; ++i)                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[(KY) + i], cl__, acc[i][j], 0, 0, 0);         \
      _Pragma("unroll") for (int i = 0; i < RW; ++i)                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[(KY) + i], ch__, acc[i][j], 0, 0, 0);         \
      if (C3D_BDIST == 2) { ch__ = dh__; cl__ = dl__; dh__ = nh__; dl__ = nl__; }                           \
      else { ch__ = nh__; cl__ = nl__; }                                                                    \
    }                                                                                                       \
    if (C3D_PROBE_READS) c3d::pin_groups<0, (NJ_), RW, (KY)>();                                             \
    bh0 = ch__; bl0 = cl__;                                                                                 \
    if (C3D_BDIST == 2) { bh1 = dh__; bl1 = dl__; }                                                         \
    stage = nstage__;                                                                                       \
    ++s;                                                                                                    \
  }
  if (p.residual) {                                             // raw SP words -> residual / wsc (both lanes of a pair exchange halves)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = nc0 + j * 32 + tx;
      const float wsc = (col < p.Cout ? p.wscale[col] : 1.f) * xinv;
      const float winv = __uint_as_float(0x7F000000u - __float_as_uint(wsc));      // 1 / wsc, exact: wsc is a power of two
#pragma unroll
      for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = sp_value(__float_as_uint(acc[i][j][r]), odd) * winv;
    }
  }
#ifdef LOFTR_CONV_PROBE
  const long long pt1__ = wall_clock64();
#endif
  __builtin_amdgcn_s_setprio(1);                               // the k-loop outranks the partner workgroup's prologue / epilogue at issue (+1.5 %)
  int s = 0;
  // the k-loop, instantiated per number of column tiles the wave owns (no data-dependent branch inside a step)
#define C3D_LOOP(NJ_)                                                                                       \
  for (int hq = 0; hq < nhalf; ++hq) {                                                                      \
    const char* sP = patch_base + (hq & 1) * CF::PHALF_BYTES;                                               \
    const char* sPn = patch_base + ((hq + 1) & 1) * CF::PHALF_BYTES;                                        \
    C3D_STEP(0, 0, NJ_) C3D_STEP(0, 1, NJ_) C3D_STEP(0, 2, NJ_)                                             \
    C3D_STEP(1, 0, NJ_) C3D_STEP(1, 1, (NJ_) + (REM ? 1 : 0)) C3D_STEP(1, 2, NJ_)     /* REM: + the P tile at the centre tap */ \
    C3D_STEP(2, 0, NJ_) C3D_STEP(2, 1, NJ_) C3D_STEP(2, 2, NJ_)                                             \
  }
  if (CF::WN == 1 || nj == NJ) { C3D_LOOP(NJ) }
  else { C3D_LOOP((NJ > 1 ? NJ - 1 : 1)) }
#undef C3D_LOOP
#undef C3D_STEP
#undef C3D_LOAD_B
#undef C3D_LOAD_ROW
#undef C3D_ISSUE_B
#undef C3D_ISSUE_PATCH

#ifdef LOFTR_CONV_PROBE
  const long long pt2__ = wall_clock64();
#endif
  __builtin_amdgcn_s_setprio(0);
  // ---- epilogue: bias (folded BN shift), activation, SP / fp32 stores (the other workgroup of the CU computes meanwhile).
  // act(x) = max(x, slope * x): slope 1 = none, 0 = ReLU, 0.01 = LeakyReLU -- no branch on the activation inside the loops.
  const float slope = p.act == 1 ? 0.f : p.act == 2 ? 0.01f : 1.f;
  if (!C3D_PROBE_EPI) {                                          // probe: keep the accumulators alive, store nothing
    float z__ = 0.f;
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) z__ += acc[i][j][0] + acc[i][j][15];
    if (z__ == 12345.678f && p.y_f32) p.y_f32[0] = z__;
  } else if (full) {
    // fast path: uniform (SGPR) row bases + one per-lane offset, stores not predicated on the pixel
    const unsigned lane_f32 = (unsigned)(4 * g * p.Cout + tx);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j >= nj) continue;
      const int col = nc0 + j * 32 + tx, colc = min(col, p.Cout - 1);
      const bool creal = col < p.Cout;                                        // (the SP row's pad channels are written as zeros)
      const float bia = p.bias ? p.bias[colc] : 0.f;
      const float wsc = p.wscale[colc] * xinv;                                // undo the operands' power-of-two scales
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const size_t pix0 = (size_t)((b * Ho + y0 + wr * RW + i) * Wo + x0);
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float xv = fmaf(acc[i][j][r], wsc, bia);
          v[r] = creal ? fmaxf(xv, slope * xv) : 0.f;
        }
        if (p.y_f32 && creal) {
          float* op = p.y_f32 + (pix0 * p.Cout + nc0 + j * 32);
#pragma unroll
          for (int r = 0; r < 16; ++r) (op + (size_t)(((r & 3) + 8 * (r >> 2)) * p.Cout))[lane_f32] = v[r];
        }
        if (p.y_sp) {
          uint32_t w16[16];
          sp_words16(v, odd, w16);
          sp_t* op = p.y_sp + (pix0 * p.Coutp + nc0 + j * 32);
#pragma unroll
          for (int r = 0; r < 16; ++r) (op + (size_t)(((r & 3) + 8 * (r >> 2)) * p.Coutp))[lane_sp] = w16[r];
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j >= nj) continue;
      const int col = nc0 + j * 32 + tx;
      const bool creal = col < p.Cout, cpad = col < p.Coutp;
      const float bia = (p.bias && creal) ? p.bias[col] : 0.f;
      const float wsc = (creal ? p.wscale[col] : 1.f) * xinv;
      const int spc = (col & ~31) + (odd ? 16 : 0) + ((col & 31) >> 1);       // dword of this lane inside the SP row
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int y = y0 + wr * RW + i;
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float xv = fmaf(acc[i][j][r], wsc, bia);
          v[r] = creal ? fmaxf(xv, slope * xv) : 0.f;
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
  }
  // ---- REM: this wave pair's tile of P = x W_rem (unshifted pixels, columns (tap, c)) -> scratch, times the filter rows' scales; the nine
  // shifted terms are added by conv_rem_gather_kernel
  if constexpr (REM) {
    if (C3D_PROBE_EPI) {
      const int R = p.Cout - NT * 32, pc = wn * 32 + tx;             // P column of this lane
      const bool creal = pc < 9 * R;
      const float wsc = p.wscale[NT * 32 + (creal ? pc % R : 0)] * xinv;
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int y = y0 + wr * RW + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (creal && y < Ho && x < Wo) p.pbuf[(size_t)((b * Ho + y) * Wo + x) * (9 * R) + pc] = acc[i][NJ][r] * wsc;
        }
      }
    }
  }
#ifdef LOFTR_CONV_PROBE
  if (g_conv_probe && tid == 0) {               // timing probe build only (tools/micro/conv_probe.py): 100 MHz wall clock stamps per workgroup
    LOFTR_WAITCNT_VM(0);
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long* o = g_conv_probe + (long)blockIdx.x * 6;
    o[0] = pt0__; o[1] = pt1__; o[2] = pt2__; o[3] = wall_clock64(); o[4] = hw; o[5] = xcc;
  }
#endif
}
