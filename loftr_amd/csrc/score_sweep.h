// Included by coarse_match.hip INSIDE its anonymous namespace (one translation unit: the kernels below use its Geometry,
// constants and reduction helpers).  Split out for readability only.
#pragma once

// ==========================================================================================
// Score-volume kernels, second generation (C == 256): a STATIONARY-OPERAND SWEEP instead of a tiled GEMM.
//
// A wave keeps the fp16 (hi, lo) MFMA fragments of 32 descriptors of image 0 for the whole K = 256 in
// registers (128 VGPRs) and sweeps them over 32-column panels of image-1 descriptors that the workgroup's
// EIGHT waves (256 rows, one workgroup per CU) share through LDS: 32 KB per panel, a four-stage ring filled by
// global_load_lds two panels ahead.  Per panel and wave: 48 MFMAs from 32 ds_read_b128 -- no B-operand staging,
// ONE s_barrier per panel (per 48 MFMAs; the tiled loop needs one per 24) and 85 B of DMA per MFMA (341 B
// there): the global -> LDS path (~6.5 TB/s chip-wide) was what bounded a 4-wave version of this kernel.
// The MFMA is issued with the image-1 panel as the A (row) operand and the image-0 fragments as the B (column)
// operand, so in the accumulator layout a LANE owns one row i of the score matrix (lane & 31) and its 16
// registers are 16 columns j = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of the panel:
//   * row statistics (pass A: online max / sum exp; pass B: running max + first argmax of conf) are lane-private
//     running values over the whole sweep -- no cross-lane reduction per tile, one half-wave exchange at the end;
//   * each lane holds 4 consecutive columns per register quad -> conf_matrix leaves as 16-byte stores;
//   * column statistics are a 32-lane DPP reduction per panel and wave, written as per-wave partials.
// The two waves that share a SIMD (w and w + 4) run HALF A PERIOD APART: waves 0-3 do {MFMAs of panel p, epilogue
// of panel p} between two barriers, waves 4-7 do {epilogue of panel p-1, MFMAs of panel p}, so one wave's VALU-only
// epilogue (exp, DPP reductions, stores) always runs under its partner's MFMAs instead of next to its epilogue.
// No LDS store and no VGPR-destination global load is issued inside the panel loop: either makes hipcc wait
// vmcnt(0) and would drain the DMA ring every iteration.
// Work unit = (pair, 256-row block, chunk of panels); the chunking depends on S only, so a pair's results do not
// depend on the batch it is in.  Row partials: one per (row, chunk); column partials: one per (column, 32-row wave).
#ifndef SWEEP_PROBE_EPI
#define SWEEP_PROBE_EPI 1        // 0: skip the epilogues (timing probe only; wrong results)
#endif
#ifndef SWEEP_PROBE_LSE
#define SWEEP_PROBE_LSE 1        // 0: pass B on (max, 1/sum) statistics instead of log-sum-exp biases
#endif
#ifndef SWEEP_PROBE_FAST
#define SWEEP_PROBE_FAST 1       // 0: pass A exact variant only
#endif
#ifndef SWEEP_PROBE_ACC1
#define SWEEP_PROBE_ACC1 0       // 1: ONE accumulator chain (no acc0 + acc1 in the epilogue): pass A -2.5 %, pass B +1.5 %, net 0.
                                 // MUST be the same in both passes: conf = exp2(2 v - LSE) near 1 relies on pass B reproducing
                                 // pass A's v bit for bit (a different summation order costs 1e-4 at logits of a few hundred)
#endif
#ifndef SWEEP_PROBE_PRIO
#define SWEEP_PROBE_PRIO 0       // 1: s_setprio 1 around the MFMA block (probe)
#endif
#ifndef SWEEP_PIPE
#define SWEEP_PIPE 0             // 0: compiler-visible ds_reads in the panel loop, order and waits left to hipcc (A/B)
#endif
#ifndef PIPE_DEP
#define PIPE_DEP 1               // phases of fragment prefetch
#endif
#define PIPE_NBUF (PIPE_DEP + 1)
#ifndef SWEEP_PROBE_NOSTORE
#define SWEEP_PROBE_NOSTORE 0    // 1: pass B without the conf_matrix stores (timing probe)
#endif
#ifndef SWEEP_PROBE_COAL
#define SWEEP_PROBE_COAL 0       // 1: pass B stores in the pattern of a lane = column layout: 16 x 4 B, 128 B contiguous per half-wave
#endif
#ifndef SWEEP_PROBE_NODMA
#define SWEEP_PROBE_NODMA 0      // 1: no LDS-DMA (timing probes only; wrong results)
#endif
#ifndef SWEEP_PROBE_NOLDS
#define SWEEP_PROBE_NOLDS 0      // 1: no panel fragment reads
#endif
#ifndef SWEEP_PROBE_NOBAR
#define SWEEP_PROBE_NOBAR 0      // 1: no per-panel barrier
#endif
#ifndef SWEEP_PROBE_SKEW
#define SWEEP_PROBE_SKEW 1       // 0: all eight waves in phase
#endif
namespace sweep {
// A 16-byte vector at 4-byte alignment: global_load / store_dwordx4 take any dword-aligned address (the runtime runs the GPU in
// unaligned access mode); rows of an [L, S] volume with S % 4 != 0 start at odd dword offsets.
struct __attribute__((packed, aligned(4))) F4U { f32x4 v; };
constexpr int W = 8, BR = 32 * W, PC = 32, KS = 16, STAGE = PC * 1024, NST = 4, MAXP = 32;
constexpr int OFF_CSTAT = NST * STAGE;                    // float2 [MAXP * PC] column (max, 1/sum) of the chunk (pass B)
constexpr int OFF_MASK = OFF_CSTAT + MAXP * PC * 8;       // uint8  [MAXP * PC] mask1 of the chunk
constexpr int LDS_BYTES = OFF_MASK + MAXP * PC;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
constexpr int DMA_PER_WAVE = PC * 8 / 8 / W;              // global_load_lds instructions per wave per panel (4)
constexpr int TIE_BIT = 1 << 30;                          // set in the argmax word of a row partial: the maximum is attained twice

struct Args {
  const sp_t* f0; const sp_t* f1;
  int N, L, S;
  int RB, NCH, PPC, NP;             // row blocks, column chunks, panels per chunk, panels in total
  float scale;
  const uint8_t* mask0; const uint8_t* mask1;
  float2* rowpart; float2* colpart;                 // pass A out: [N][NCH][L], [N][RB * W][S]
  const float2* rowstat; const float2* colstat;     // pass B in
  float* conf;                                      // pass B out or null
  float2* rowmax_part; float* colmax_part;          // pass B out: [N][NCH][L] (max, argmax | TIE_BIT), [N][RB * W][S]
  int* exact_flags;                                 // pass A: [N * NCH * RB] units the exact variant has to (re)do, or null = all
};

__device__ __forceinline__ int jr(int r, int g) { return 8 * (r >> 2) + 4 * g + (r & 3); }
// the value of register (lane & 15) of a 16-register vector: lane l of a half-wave then owns column jr(l & 15, g)
__device__ __forceinline__ float pick16(const f32x16& v, int sel) {
  float x = v[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) x = sel == r ? v[r] : x;
  return x;
}

// ---- transposed reductions over the 32 lanes of a half-wave --------------------------------------------------
// A plain reduction of 16 registers across 32 lanes costs 16 x 5 cross-lane steps and leaves every lane with all 16
// results.  Here every step HALVES the registers a lane carries (it keeps the half selected by one of its lane bits and
// hands the other half to its partner, who keeps exactly that one): 8 + 4 + 2 + 1 + 1 steps, and lane l ends up with
// the full reduction of register tr_reg(l) only -- which is all the column partials need (one lane stores one column).
//   step partners: l ^ 7 (row_half_mirror), l ^ 1, l ^ 2 (quad_perm), l ^ 8 (row_ror:8), l ^ 16 (v_permlane16_swap);
//   register kept by lane l:  8 * bit2(l) + 4 * bit0(l) + 2 * bit1(l) + bit3(l).
__device__ __forceinline__ int tr_reg(int l) { return 8 * ((l >> 2) & 1) + 4 * (l & 1) + 2 * ((l >> 1) & 1) + ((l >> 3) & 1); }
#define SWEEP_DPPF(v_, c_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), c_, 0xF, 0xF, true))
#define SWEEP_DPPI(v_, c_) __builtin_amdgcn_update_dpp(0, v_, c_, 0xF, 0xF, true)
// lane-dependent choice between two registers as ONE v_bfi_b32 on a precomputed all-ones / all-zeros lane mask
// (a bool select costs a v_cmp + hazard nops + v_cndmask each time: hipcc re-materialises the comparison)
__device__ __forceinline__ int bsel(int m, int a1, int a0) { return (a1 & m) | (a0 & ~m); }      // m ? a1 : a0
__device__ __forceinline__ float bself(int m, float a1, float a0) { return __int_as_float(bsel(m, __float_as_int(a1), __float_as_int(a0))); }
struct TrMasks { int m0, m1, m2, m3; };            // lane bit k set -> all ones
__device__ __forceinline__ TrMasks tr_masks(int lane) { return TrMasks{-(lane & 1), -((lane >> 1) & 1), -((lane >> 2) & 1), -((lane >> 3) & 1)}; }
__device__ __forceinline__ float treduce_add16(const f32x16& v, const TrMasks& t) {
  float x8[8], x4[4], x2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) x8[k] = bself(t.m2, v[8 + k], v[k]) + SWEEP_DPPF(bself(t.m2, v[k], v[8 + k]), 0x141);
#pragma unroll
  for (int k = 0; k < 4; ++k) x4[k] = bself(t.m0, x8[4 + k], x8[k]) + SWEEP_DPPF(bself(t.m0, x8[k], x8[4 + k]), 0xB1);
#pragma unroll
  for (int k = 0; k < 2; ++k) x2[k] = bself(t.m1, x4[2 + k], x4[k]) + SWEEP_DPPF(bself(t.m1, x4[k], x4[2 + k]), 0x4E);
  const float x1 = bself(t.m3, x2[1], x2[0]) + SWEEP_DPPF(bself(t.m3, x2[0], x2[1]), 0x128);
  const auto q = __builtin_amdgcn_permlane16_swap(__float_as_int(x1), __float_as_int(x1), false, false);
  return __int_as_float(q[0]) + __int_as_float(q[1]);
}
// the same with a signed-integer maximum: on the bit patterns of NON-NEGATIVE floats it is the float maximum (and any
// negative float, the "invalid" marker -1, loses) -- without the canonicalisation fmaxf costs on cross-lane values
__device__ __forceinline__ int treduce_imax16(const int (&v)[16], const TrMasks& t) {
  int x8[8], x4[4], x2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) x8[k] = max(bsel(t.m2, v[8 + k], v[k]), SWEEP_DPPI(bsel(t.m2, v[k], v[8 + k]), 0x141));
#pragma unroll
  for (int k = 0; k < 4; ++k) x4[k] = max(bsel(t.m0, x8[4 + k], x8[k]), SWEEP_DPPI(bsel(t.m0, x8[k], x8[4 + k]), 0xB1));
#pragma unroll
  for (int k = 0; k < 2; ++k) x2[k] = max(bsel(t.m1, x4[2 + k], x4[k]), SWEEP_DPPI(bsel(t.m1, x4[k], x4[2 + k]), 0x4E));
  const int x1 = max(bsel(t.m3, x2[1], x2[0]), SWEEP_DPPI(bsel(t.m3, x2[0], x2[1]), 0x128));
  const auto q = __builtin_amdgcn_permlane16_swap(x1, x1, false, false);
  return max((int)q[0], (int)q[1]);
}
constexpr float LOG2E = 1.4426950408889634f;
constexpr float FAST_SPREAD = 64.f;    // max - min of a wave's 32 x 32 score tile up to which one shared exp reference is exact enough

// FASTA (pass A only): the lean epilogue with ONE shared exp reference per 32 x 32 wave tile.  It is exact only while
// the tile's values span less than FAST_SPREAD and the tile is full, so this variant gives up on a unit the moment a
// tile fails the test (or at once for units with a partial last panel): it raises the unit's flag and the
// exact variant -- launched right behind it on the same grid, returning immediately for unflagged units -- redoes
// that unit with per-row / per-column references.  Two kernels instead of one two-path kernel: together the paths
// exceed the 256-VGPR budget of a 512-thread workgroup (30 spills measured).
// TRACKJ (pass B): per-ELEMENT tracking of the first row arg-max (5 VALU per element).  Without it the sweep only tracks,
// per lane, the maximum and the PANEL it first occurred in (11 VALU per panel) and select_kernel finds the column --
// and any second occurrence -- by reading those 32 entries of conf_matrix back; that needs the materialised matrix.
// The sweep is issue-bound (about five non-MFMA instructions fit under one 32-cycle MFMA), so this matters.
template <int PASS, bool HAS_MASK, bool FASTA = false, bool TRACKJ = true>
__global__ __launch_bounds__(512, 2) void score_sweep_kernel(Args a) {
  static_assert(!(FASTA && (PASS != 0 || HAS_MASK)), "the shared-reference path is pass A without masks");
  static_assert(TRACKJ || (PASS == 1 && !HAS_MASK), "panel-level tracking is the unmasked pass B");
  constexpr bool LSE = SWEEP_PROBE_LSE && PASS == 1 && !HAS_MASK;      // pass B on log-sum-exp biases (one exp2(fma) per element)
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  // ---- unit: groups (chunk, pair) are dealt to the XCDs; the row blocks of a group run back to back on it
  const int id = blockIdx.x, xcd = id % NUM_XCD, slot = id / NUM_XCD;
  const int grp = (slot / a.RB) * NUM_XCD + xcd, rb = slot % a.RB;
  if (grp >= a.N * a.NCH) return;
  const int n = grp % a.N, cc = grp / a.N;
  const int p0 = cc * a.PPC, np = min(a.PPC, a.NP - p0);      // panels of this chunk (>= 1 by construction)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  const bool late = SWEEP_PROBE_SKEW && wave >= W / 2;         // this wave runs its epilogue half a period later
  const int L = a.L, S = a.S;
  const int row = rb * BR + wave * 32 + li;
  const bool row_ok = row < L;
  const bool rows_full = FASTA || rb * BR + BR <= L;           // block-uniform (the fast variant only keeps full units)
  int* const unit_flag = a.exact_flags ? a.exact_flags + (grp * a.RB + rb) : nullptr;
  const bool rows_part = rb * BR + BR > L;                     // block-uniform: the last row block of the pair is partial
  if (PASS == 0 && FASTA) {                                    // partial last panel: the exact variant's job
    if ((p0 + np) * PC > S) { if (threadIdx.x == 0) *unit_flag = 1; return; }
  }
  if (PASS == 0 && !FASTA && unit_flag && *unit_flag == 0) return;   // the fast variant has done this unit
  const sp_t* f0n = a.f0 + (long)n * L * 256;
  const sp_t* f1n = a.f1 + (long)n * S * 256;

  // ---- stationary operand: this lane's 16-byte MFMA fragments of its row, all 16 k-steps, hi and lo
  h16x8 bh[KS], bl[KS];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(f0n + (long)min(row, L - 1) * 256);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = (ks >> 1) * 8 + 2 * (ks & 1) + g;
      bh[ks] = __builtin_bit_cast(h16x8, src[c]);
      bl[ks] = __builtin_bit_cast(h16x8, src[c + 4]);
    }
  }
  // ---- chunk tables -> LDS (ordinary loads and LDS stores happen only here, before any DMA is in flight)
  float2* cstat_s = reinterpret_cast<float2*>(lds + OFF_CSTAT);
  uint8_t* mask_s = reinterpret_cast<uint8_t*>(lds + OFF_MASK);
  bool cols_dead = HAS_MASK;                       // every column of this chunk that this thread looked at is padding
  for (int t = threadIdx.x; t < np * PC; t += 512) {
    const int col = min(p0 * PC + t, S - 1);
    if (PASS == 1) {
      const float2 cs = a.colstat[(long)n * S + col];
      // LSE form: -log2 sum_i exp(v_ij) = -(max * log2 e) + log2(1 / sum): independent of which reference "max" was
      if (LSE) reinterpret_cast<float*>(cstat_s)[t] = fmaf(-cs.x, LOG2E, __builtin_amdgcn_logf(cs.y));
      else cstat_s[t] = cs;
    }
    if (HAS_MASK) { const uint8_t m1 = a.mask1[(long)n * S + col]; mask_s[t] = m1; cols_dead = cols_dead && m1 == 0; }
  }
  float rm = 0.f, rs = 0.f;                        // pass B: row (max, 1/sum); LSE form: rm = -log2 sum_j exp(v_ij)
  if (PASS == 1) {
    const float2 t = a.rowstat[(long)n * L + min(row, L - 1)];
    rm = LSE ? fmaf(-t.x, LOG2E, __builtin_amdgcn_logf(t.y)) : t.x; rs = t.y;
  }
  const float k2 = 2.f * a.scale * LOG2E;          // LSE form: conf = exp2(k2 * dot + rm + cb_j)
  const bool mrow = HAS_MASK ? (a.mask0[(long)n * L + min(row, L - 1)] != 0) : true;
  // Padding masks (MegaDepth batches): a unit all of whose rows or all of whose columns are padding holds nothing but the fill value
  // -1e9 (coarse_matching.py:115-118) whatever the descriptors are -- its DMA, barriers and 48 MFMAs per panel are skipped and the
  // epilogues run on zero accumulators, which they overwrite with the fill value exactly as they would the real dot products
  // (bit-identical results; at 840 x 840 padded from 840 x 560 that is 56 % of the units).  Block-uniform.
  const bool dead = HAS_MASK && (__syncthreads_and(cols_dead) || __syncthreads_and(!mrow || !row_ok));

  // ---- DMA of one panel: 32 rows x 8 k-groups x 128 B = 32 instructions, 4 per wave (k-group = wave)
  // dword offset of this lane's 16 B inside a panel row set, per row octet: recomputed per issue (a few VALU per panel)
  // rather than held in registers across the loop
#define SWEEP_DOFF(oct_) ((oct_) * 8 * 256 + (lane >> 3) * 256 + (((lane & 7) ^ (((oct_) * 4 + (lane >> 4)) & 7)) << 2) + wave * 32)
#define SWEEP_ISSUE(p_)                                                                                  \
  {                                                                                                      \
    const int col0__ = (p0 + (p_)) * PC;                                                                 \
    char* st__ = lds + ((p_) & (NST - 1)) * STAGE + wave * 4096;                                         \
    if (col0__ + PC <= S) {                                                                              \
      const sp_t* base__ = f1n + (long)col0__ * 256;                                                     \
      _Pragma("unroll") for (int oct__ = 0; oct__ < 4; ++oct__)                                          \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base__ + SWEEP_DOFF(oct__)), (lds_ptr_t)(st__ + oct__ * 1024), 16, 0, 0); \
    } else {                  /* last panel of the matrix: rows beyond S re-read row S-1 (masked later) */ \
      _Pragma("unroll") for (int oct__ = 0; oct__ < 4; ++oct__) {                                        \
        const int r__ = oct__ * 8 + (lane >> 3);                                                         \
        const int gc__ = min(col0__ + r__, S - 1);                                                       \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(f1n + (long)gc__ * 256 + (SWEEP_DOFF(oct__) - r__ * 256)), \
                                         (lds_ptr_t)(st__ + oct__ * 1024), 16, 0, 0);                    \
      }                                                                                                  \
    }                                                                                                    \
  }

  // ---- running row state (lane private)
  // pass A: sum of exp(v - ref_run) over the columns seen so far; ref_run >= every v seen is a running REFERENCE, not
  // necessarily the maximum -- the merge kernels and pass B only ever use max-reference + log(sum) combinations
  float ref_run = SENTINEL, s_run = 0.f;
  float best = -1.f; int bestj = 0; bool tie = false;   // pass B
  const TrMasks trm = tr_masks(lane);
  const int trcol = jr(tr_reg(lane & 15), g);       // panel column whose transposed reduction ends in this lane
  [[maybe_unused]] const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;     // SWEEP_PIPE: LDS address of the panel ring
  const int a_off = lds_chunk_off(li, g);          // hi chunk of the even k-step; odd k-step: ^ 32, lo: ^ 64 (chunk + 2 / + 4)
  const int sel = lane & 15;
  const long part_row = ((long)n * a.RB * W + rb * W + wave) * S;          // this wave's row of the column partials

  // Epilogue of panel p_ on the accumulators: statistics / conf_matrix.  A macro, not a lambda (captures of the register
  // arrays by reference end up in scratch); expanded twice (early and late waves).
#define SWEEP_ACC(r_) (SWEEP_PROBE_ACC1 ? acc0[r_] : acc0[r_] + acc1[r_])
#define SWEEP_EPILOGUE(p_)                                                                               \
  {                                                                                                      \
    const int col0 = (p0 + (p_)) * PC;                                                                   \
    const bool fullp = FASTA || col0 + PC <= S;    /* panel-uniform */                                   \
    f32x16 v;                                                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) v[r] = SWEEP_ACC(r) * a.scale;                        \
    if (HAS_MASK) {                                                                                      \
      _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                     \
        if (!(mrow && mask_s[(p_) * PC + jr(r, g)])) v[r] = LOFTR_NEG_INF;   /* masked_fill_(~(m0 x m1), -INF)  :115-118 */ \
    }                                                                                                    \
    if (!fullp) {                                                                                        \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) if (col0 + jr(r, g) >= S) v[r] = SENTINEL;          \
    }                                                                                                    \
    const int mycol = col0 + jr(sel, g);           /* the column this lane stores a partial for */       \
    if (PASS == 2) {                               /* Sinkhorn: the scaled, mask-filled score itself (coarse_matching.py:123-126) */ \
      if (rows_full || row_ok) {                                                                         \
        float* co = a.conf + ((long)n * L + row) * S + col0 + 4 * g;                                     \
        if (fullp) {   /* 16-byte stores; rows of an S % 4 != 0 volume (outdoor 105 x 105 grids) are only 4-byte aligned: F4U */ \
          _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
            reinterpret_cast<F4U*>(co + 8 * q)->v = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; \
        } else {                                                                                         \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) if (col0 + jr(r, g) < S) co[8 * (r >> 2) + (r & 3)] = v[r]; \
        }                                                                                                \
      }                                                                                                  \
    } else if (PASS == 0) {                                                                                     \
      float tm = FASTA ? SWEEP_ACC(0) : v[0], tn = tm;      /* FASTA: extrema of the RAW dot products (scale > 0) */ \
      _Pragma("unroll") for (int r = 1; r < 16; r += 2) {                                                \
        const float x0 = FASTA ? SWEEP_ACC(r) : v[r], x1 = r + 1 < 16 ? (FASTA ? SWEEP_ACC(r + 1) : v[r + 1]) : x0; \
        tm = fmaxf(fmaxf(tm, x0), x1); tn = fminf(fminf(tn, x0), x1);            /* v_max3 / v_min3 */  \
      }                                                                                                  \
      if (FASTA) {                                                                                       \
        float R = half_max(tm); R = fmaxf(R, swap32(R));           /* maximum / minimum of the wave's 32 x 32 tile */ \
        float mnw = -half_max(-tn); mnw = fminf(mnw, swap32(mnw));                                       \
        R *= a.scale; mnw *= a.scale;                                                                    \
        if (!(R - mnw <= FAST_SPREAD) && lane == 0) *unit_flag = 1;      /* (also catches NaN) -> redone exactly */ \
        /* ONE exponential per element, relative to the tile maximum R, serves the row AND the column sums: every   \
           element is within FAST_SPREAD of R, so nothing that matters to any row or column underflows.  Packed     \
           fp32 arithmetic (v_pk_fma_f32 / v_pk_add_f32: two elements per instruction) on the raw accumulators,     \
           exponent = dot * (scale log2 e) - R log2 e with the same scale log2 e = k2 / 2 pass B uses */            \
        const f32x2 sl2 = {0.5f * k2, 0.5f * k2}, nrk2 = {-R * LOG2E, -R * LOG2E};                       \
        f32x16 e;                                                                                        \
        f32x2 ss2 = {0.f, 0.f};                                                                          \
        _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                              \
          const f32x2 x2 = __builtin_elementwise_fma(f32x2{SWEEP_ACC(r), SWEEP_ACC(r + 1)}, sl2, nrk2);  \
          e[r] = __builtin_amdgcn_exp2f(x2.x); e[r + 1] = __builtin_amdgcn_exp2f(x2.y);                  \
          ss2 += f32x2{e[r], e[r + 1]};                                                                  \
        }                                                                                                \
        const float ssum = ss2.x + ss2.y;                                                                \
        const float Rn = fmaxf(ref_run, R);                                                              \
        s_run = s_run * fexp(ref_run - Rn) + ssum * fexp(R - Rn);                                        \
        ref_run = Rn;                                                                                    \
        /* columns: sum of e over the wave's 32 rows, one column per lane, as (reference, sum); rows beyond L are      \
           clamped copies of row L-1 (harmless for the tile extrema) and drop out here */                \
        if (rows_part) { _Pragma("unroll") for (int r = 0; r < 16; ++r) e[r] = row_ok ? e[r] : 0.f; }    \
        const float csum = treduce_add16(e, trm);                                                        \
        if (li < 16) a.colpart[part_row + col0 + trcol] = make_float2(R, csum);                          \
      } else {                                                                                           \
        /* exact path: per-row reference = the running row maximum, per-column reference = the column maximum */    \
        const float mn = fmaxf(ref_run, tm);                                                             \
        float ssum = 0.f;                                                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) ssum += fexp(v[r] - mn);    /* exp(SENTINEL - x) == 0 */ \
        s_run = s_run * fexp(ref_run - mn) + ssum;                                                       \
        ref_run = mn;                                                                                    \
        f32x16 cm;                                                                                       \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) cm[r] = (rows_full || row_ok) ? v[r] : SENTINEL;  \
        f32x16 e = cm;                                                                                   \
        half_max16(cm);                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) e[r] = fexp(e[r] - cm[r]);                        \
        half_sum16(e);                                                                                   \
        const float2 mine = make_float2(pick16(cm, sel), pick16(e, sel));                                \
        if (li < 16 && (fullp || mycol < S)) a.colpart[part_row + mycol] = mine;                         \
      }                                                                                                  \
    } else {                                                                                             \
      /* conf = softmax(sim, dim=1) * softmax(sim, dim=2) = exp((v - rowmax) + (v - colmax)) / (rowsum * colsum)   :119 */ \
      f32x16 c;                                                                                          \
      if (LSE) {                                                                                         \
        /* = exp2(2 v log2e - LSE_row - LSE_col): the (acc0 + acc1) * scale above folds into the fma */  \
        const f32x2 k22 = {k2, k2}, rm2 = {rm, rm};                                                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                  \
          const f32x4 cb = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(cstat_s) + (p_) * PC + 8 * q + 4 * g); \
          _Pragma("unroll") for (int e = 0; e < 4; e += 2) {            /* two elements per v_pk_add / v_pk_fma */ \
            const f32x2 x2 = __builtin_elementwise_fma(f32x2{SWEEP_ACC(4 * q + e), SWEEP_ACC(4 * q + e + 1)}, k22, \
                                                       rm2 + f32x2{cb[e], cb[e + 1]});                   \
            c[4 * q + e] = __builtin_amdgcn_exp2f(x2.x); c[4 * q + e + 1] = __builtin_amdgcn_exp2f(x2.y); \
          }                                                                                              \
        }                                                                                                \
      } else {                                                                                           \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                    \
        const f32x4* cs4 = reinterpret_cast<const f32x4*>(cstat_s + (p_) * PC + 8 * q + 4 * g);   /* (max, 1/sum) x 4 columns */ \
        const f32x4 c01 = cs4[0], c23 = cs4[1];                                                          \
        const float cmx[4] = {c01.x, c01.z, c23.x, c23.z}, cis[4] = {c01.y, c01.w, c23.y, c23.w};        \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
          const float x = v[4 * q + e];                                                                  \
          c[4 * q + e] = fexp((x - rm) + (x - cmx[e])) * (rs * cis[e]);                                  \
        }                                                                                                \
      }                                                                                                  \
      }                                                                                                  \
      if (!fullp) {                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) if (col0 + jr(r, g) >= S) c[r] = -1.f;            \
      }                                                                                                  \
      if (a.conf && (rows_full || row_ok)) {                                                             \
        float* co = a.conf + ((long)n * L + row) * S + col0 + 4 * g;                                     \
        if (SWEEP_PROBE_NOSTORE) {                                                                       \
        } else if (SWEEP_PROBE_COAL) {      /* timing probe: the store pattern of a lane = column layout (wrong data) */ \
          float* cq = a.conf + ((long)n * L + rb * BR + wave * 32) * S + col0 + li;                      \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) cq[(long)jr(r, g) * S] = c[r];                  \
        } else if (fullp) {                                                                              \
          _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
            reinterpret_cast<F4U*>(co + 8 * q)->v = f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]}; \
        } else {                                                                                         \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) if (col0 + jr(r, g) < S) co[8 * (r >> 2) + (r & 3)] = c[r]; \
        }                                                                                                \
      }                                                                                                  \
      if (TRACKJ) {                                                                                      \
        /* row: running (max, FIRST argmax, attained-twice flag); registers ascend in column order for this half */ \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
          const bool gt = c[r] > best;                                                                   \
          tie = gt ? false : (tie || c[r] == best);                                                      \
          bestj = gt ? col0 + jr(r, g) : bestj;                                                          \
          best = gt ? c[r] : best;                                                                       \
        }                                                                                                \
      } else {                                                                                           \
        /* row: running maximum and the FIRST panel that attains it (select_kernel finds the column) */  \
        float pm = c[0];                                                                                 \
        _Pragma("unroll") for (int r = 1; r < 16; r += 2) pm = fmaxf(fmaxf(pm, c[r]), r + 1 < 16 ? c[r + 1] : c[r]);   /* v_max3 */ \
        const bool gt = pm > best;                                                                       \
        tie = gt ? false : (tie || pm == best);                                                          \
        bestj = gt ? (p0 + (p_)) : bestj;                                                                \
        best = gt ? pm : best;                                                                           \
      }                                                                                                  \
      /* columns: max over the wave's rows, one column per lane (conf >= 0: integer maximum of the bit patterns) */ \
      int cb[16];                                                                                        \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) cb[r] = __float_as_int((rows_full || row_ok) ? c[r] : -1.f); \
      const float cmine = __int_as_float(treduce_imax16(cb, trm));                            \
      if (li < 16 && (fullp || col0 + trcol < S)) a.colmax_part[part_row + col0 + trcol] = cmine;        \
    }                                                                                                    \
  }

  // Every ordinary load above must be COMPLETE before the first DMA is issued: hipcc waits vmcnt(0) at the first use
  // of a VGPR-destination load, and a first use inside the panel loop would drain the in-flight DMA every iteration.
  LOFTR_WAITCNT_VM(0);
  __syncthreads();                                 // tables visible; no DMA in flight yet
  if (!dead) {
    SWEEP_ISSUE(0);
    if (np > 1) SWEEP_ISSUE(1);
  }
  // Stores a wave issues between two DMA issues (they sit between the DMA of panel p+1 and the barrier of panel p+1
  // in the in-order VMEM queue): pass A one partial store; pass B four conf stores + one partial store.  Panels that
  // take the scalar-store tail path are followed by a full drain instead.
  constexpr int ST = PASS == 0 ? 1 : PASS == 2 ? 4 : (SWEEP_PROBE_NOSTORE ? 1 : SWEEP_PROBE_COAL ? 17 : 5);
  f32x16 acc0, acc1;
  bool drain = false;                              // block-uniform: the previous period issued an unknown number of stores
  for (int p = 0; p < np; ++p) {
    // panel p has landed once at most {DMA of panel p+1, the epilogue stores issued after it} are outstanding (VMEM
    // operations retire in order).  The late waves have not stored anything before period 2, so the count only
    // includes the stores from there on (conservative for the early waves at p = 1).
    if (!dead) {
    if (drain || p + 1 >= np) LOFTR_WAITCNT_VM(0);
    else if (p < 2) LOFTR_WAITCNT_VM(DMA_PER_WAVE);
    else LOFTR_WAITCNT_VM(DMA_PER_WAVE + ST);
    if (!SWEEP_PROBE_NOBAR) __builtin_amdgcn_s_barrier();   // ... for every wave; and every wave is past the MFMAs of panel p-2
    if (p + 2 < np && !SWEEP_PROBE_NODMA) SWEEP_ISSUE(p + 2);
    }
    drain = !((p0 + p) * PC + PC <= S) && PASS >= 1;      // only the ragged last panel of the matrix takes the scalar-store path
    const char* st = lds + (p & (NST - 1)) * STAGE;
#if SWEEP_PIPE
    // ---- 48 MFMAs in eight phases of two k-steps (one 4 KB k-group of the panel: hi / lo fragments of an even and an
    // odd k-step = four ds_read_b128), the fragments of phase ph + PIPE_DEP in flight while phase ph multiplies.
    // The reads are inline asm with COUNTED lgkmcnt waits: while LDS-DMA is in flight hipcc's wait insertion degrades
    // every LDS dependency to lgkmcnt(0) (the DMA counts as a pending flat access), so with compiler-visible reads a
    // wave exposes the full LDS latency once per ds_read group and cannot keep the matrix pipe busy on its own --
    // which is what the half-period skew of the two waves of a SIMD relies on.  LDS reads return in order, so
    // "phase ph has landed" is lgkmcnt(4 PIPE_DEP) right after the reads of phase ph + PIPE_DEP were issued; the wait
    // statement names the fragments it releases ("+v"), which is what keeps their MFMAs below it.
    h16x8 fr[PIPE_NBUF][4];
    const unsigned stb = lds_base + (p & (NST - 1)) * STAGE;
    const unsigned ad0 = stb + a_off, ad1 = stb + (a_off ^ 64), ad2 = stb + (a_off ^ 32), ad3 = stb + (a_off ^ 96);
#define SWEEP_LOADPH(ph_)                                                                                \
    asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\t"                  \
                 "ds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8"                       \
                 : "=&v"(fr[(ph_) % PIPE_NBUF][0]), "=&v"(fr[(ph_) % PIPE_NBUF][1]), "=&v"(fr[(ph_) % PIPE_NBUF][2]),  \
                   "=&v"(fr[(ph_) % PIPE_NBUF][3])                                                       \
                 : "v"(ad0), "v"(ad1), "v"(ad2), "v"(ad3), "i"((ph_) * 4096));
#define SWEEP_WAITPH(ph_, n_)                                                                            \
    asm volatile("s_waitcnt lgkmcnt(%4)"                                                                 \
                 : "+v"(fr[(ph_) % PIPE_NBUF][0]), "+v"(fr[(ph_) % PIPE_NBUF][1]), "+v"(fr[(ph_) % PIPE_NBUF][2]),    \
                   "+v"(fr[(ph_) % PIPE_NBUF][3])                                                        \
                 : "i"(n_));
#define SWEEP_PHASE(ph_)                                                                                 \
    if ((ph_) + PIPE_DEP < 8) SWEEP_LOADPH((ph_) + PIPE_DEP)                                             \
    SWEEP_WAITPH(ph_, 4 * ((ph_) + PIPE_DEP < 8 ? PIPE_DEP : 7 - (ph_)))                                 \
    {                                                                                                    \
      const h16x8 ah0 = fr[(ph_) % PIPE_NBUF][0], al0 = fr[(ph_) % PIPE_NBUF][1];                        \
      const h16x8 ah1 = fr[(ph_) % PIPE_NBUF][2], al1 = fr[(ph_) % PIPE_NBUF][3];                        \
      if (SWEEP_PROBE_ACC1) {                                                                            \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[2 * (ph_)], acc0, 0, 0, 0);                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[2 * (ph_)], acc0, 0, 0, 0);                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[2 * (ph_)], acc0, 0, 0, 0);                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh[2 * (ph_) + 1], acc0, 0, 0, 0);            \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl[2 * (ph_) + 1], acc0, 0, 0, 0);            \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh[2 * (ph_) + 1], acc0, 0, 0, 0);            \
      } else {                                                                                           \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[2 * (ph_)], acc0, 0, 0, 0);                \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[2 * (ph_)], acc1, 0, 0, 0);                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[2 * (ph_)], acc0, 0, 0, 0);                \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh[2 * (ph_) + 1], acc1, 0, 0, 0);            \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl[2 * (ph_) + 1], acc0, 0, 0, 0);            \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh[2 * (ph_) + 1], acc1, 0, 0, 0);            \
      }                                                                                                  \
    }                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);
#if SWEEP_PROBE_NOLDS
#undef SWEEP_LOADPH
#undef SWEEP_WAITPH
#define SWEEP_LOADPH(ph_)
#define SWEEP_WAITPH(ph_, n_)
#pragma unroll
    for (int b_ = 0; b_ < PIPE_NBUF; ++b_) { fr[b_][0] = bh[b_]; fr[b_][1] = bl[b_]; fr[b_][2] = bh[b_ + 4]; fr[b_][3] = bl[b_ + 4]; }
#endif
    SWEEP_LOADPH(0)
    if (PIPE_DEP > 1) SWEEP_LOADPH(1)
    if (SWEEP_PROBE_EPI && late && p > 0) SWEEP_EPILOGUE(p - 1);      // (its VALU work covers the latency of the first fragments)
    if (SWEEP_PROBE_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    SWEEP_PHASE(0) SWEEP_PHASE(1) SWEEP_PHASE(2) SWEEP_PHASE(3) SWEEP_PHASE(4) SWEEP_PHASE(5) SWEEP_PHASE(6) SWEEP_PHASE(7)
#undef SWEEP_LOADPH
#undef SWEEP_WAITPH
#undef SWEEP_PHASE
#else
    if (SWEEP_PROBE_EPI && late && p > 0) SWEEP_EPILOGUE(p - 1);
    // ---- 48 MFMAs: two accumulators alternate so that no MFMA depends on its predecessor
    if (SWEEP_PROBE_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (!dead)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const char* sk = st + (ks >> 1) * 4096;
      const h16x8 ah = *reinterpret_cast<const h16x8*>(sk + (a_off ^ ((ks & 1) ? 32 : 0)));
      const h16x8 al = *reinterpret_cast<const h16x8*>(sk + (a_off ^ ((ks & 1) ? 96 : 64)));
      if (SWEEP_PROBE_ACC1) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc0, 0, 0, 0);
      } else if (ks & 1) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc0, 0, 0, 0);
      }
    }
#endif
    if (SWEEP_PROBE_PRIO) __builtin_amdgcn_s_setprio(0);
    if (SWEEP_PROBE_EPI && !late) SWEEP_EPILOGUE(p);
    if (!SWEEP_PROBE_EPI) { s_run += acc0[0] + (SWEEP_PROBE_ACC1 ? 0.f : acc1[5]); best += acc0[3] + (SWEEP_PROBE_ACC1 ? 0.f : acc1[7]); }     // keep the MFMAs alive
  }
  if (SWEEP_PROBE_EPI && late) SWEEP_EPILOGUE(np - 1);
#undef SWEEP_ISSUE
#undef SWEEP_DOFF
#undef SWEEP_EPILOGUE
#undef SWEEP_ACC
#undef SWEEP_DPPF
#undef SWEEP_DPPI
  // ---- row partials of this chunk: combine the two half-waves (they hold disjoint columns of the same row)
  if (PASS == 2) return;
  float2* rp = (PASS == 0 ? a.rowpart : a.rowmax_part) + ((long)n * a.NCH + cc) * L;
  if (PASS == 0) {
    const float ro = swap32(ref_run), so = swap32(s_run);
    const float M = fmaxf(ref_run, ro);
    const float Ssum = s_run * fexp(ref_run - M) + so * fexp(ro - M);
    if (g == 0 && row_ok) rp[row] = make_float2(M, Ssum);
  } else {
    const float bo = swap32(best);
    const int jo = __float_as_int(swap32(__int_as_float(bestj)));
    const bool to = swap32(tie ? 1.f : 0.f) != 0.f;
    const bool other = bo > best || (bo == best && jo < bestj);
    const bool t = (bo == best) || (bo > best ? to : (bo < best ? tie : false));
    const float B = other ? bo : best;
    const int J = other ? jo : bestj;
    if (g == 0 && row_ok) rp[row] = make_float2(B, __int_as_float(J | (t ? TIE_BIT : 0)));
  }
}
}  // namespace sweep
