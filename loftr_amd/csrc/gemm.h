// fp32-accurate "NT" GEMM main loop on the CDNA4 matrix cores:  acc[m][n] += sum_k A[m][k] * B[n][k]
//
// Precision design ("f16x3 split").  The dual-softmax confidences must match the reference within
// 1e-4 and the sub-pixel key points within 1e-3 px; bf16/fp16/tf32-rounded operands miss that by
// 10-100x, and gfx950 has no xf32.  The fp32-input MFMA (v_mfma_f32_32x32x2_f32) is
// exact but runs at the fp32 VECTOR rate (157 TFLOP/s), 1/16 of the fp16 matrix rate.  Instead every
// fp32 value x that feeds a GEMM is represented by two fp16 numbers
//        hi = fp16(x),   lo = fp16(x - hi)          (x - hi is exact in fp32)
// and each product is evaluated as three fp16 MFMAs with fp32 accumulation,
//        a*b  ~=  hi_a*hi_b + hi_a*lo_b + lo_a*hi_b          (dropped: lo_a*lo_b ~ 2^-22 |a*b|)
// i.e. ~22 mantissa bits per product at 3/16 of the cost of the fp32 MFMA (peak 2.5 PF / 3).  Every
// fp16 x fp16 product is exact in fp32, so the only extra error over an fp32 fma chain is the
// representation / dropped-term error: measured end-to-end through the 8 coarse layers + dual-softmax it
// moves conf by 1.4e-5 vs an fp64 run (fp32 chain: 1.9e-5), DESIGN.md §5.
// Representation error and SCALING.  |x - hi - lo| <= 2^-22 |x| holds only while lo is a NORMAL fp16 number,
// i.e. for |x| >= 2^-3: below that lo is subnormal and the error is the absolute 2^-25 (the MFMA does not flush
// subnormals, tools/micro/f16_denorm.hip) -- 2e-5 relative at |x| = 1e-3.  What a GEMM needs is error small against
// the ROW it is summed with, so operands are stored pre-multiplied by a power of two that lifts the row maximum to
// [2^13, 2^14) (exact; undone in the consumer's epilogue, also exactly):
//   * weight matrices / convolution filters: one exponent per output row (sp_convert with SpJobs::inv_scale,
//     conv_prep_kernel), multiplied back per output column -- entries down to 2^-17 of the row maximum keep 22 bits;
//   * activations entering through the C-ABI as fp32 (loftr_linear_fwd: per row; loftr_sp_from_f32_scaled: per
//     tensor) likewise; activations produced INSIDE the pipeline are LayerNorm / BatchNorm bounded (O(1)) and are
//     stored unscaled -- an epilogue cannot know its tensor's maximum before it has written it.
// Range: a scaled |x| stays below 2^14, an unscaled one must stay below the fp16 maximum 65504.
//
// The SP ("split pair") tensor format.  Doing the split inside the GEMM costs ~4 VALU instructions
// per operand element per tile -- measured 9 VALU per MFMA, matrix pipe 23 % busy (profiles/,
// r01 v2).  So GEMM operands live in HBM already split: a row-major [rows, K] tensor keeps, for
// every group of 32 consecutive k, 32 hi halfs followed by 32 lo halfs (128 B -- exactly the
// footprint and row pitch of the fp32 tensor it replaces).  Producers write SP from their
// epilogues (sp_store below: one dword per lane, full 128-B lines), weights are converted once
// per call (sp_convert.hip), and the main loop is copy + MFMA only.
//
// Tiling (per workgroup of WM*WN waves of 64 lanes):
//   * block tile BM x BN, k-tile BK = 32 elements = one 128-B SP group per row; each wave owns a
//     (BM/WM) x (BN/WN) sub-tile made of TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each),
//     v_mfma_f32_32x32x16_f16.
//   * staging global -> registers -> LDS as plain 16-B copies, LDS double buffered, A registers
//     double buffered on top (A streams from HBM: loads run two k-tiles ahead; B -- weights or one
//     pair's descriptors -- is L2 resident: one tile ahead).
//   * LDS rows are the 128-B SP groups, unpadded, 16-B chunk c of row r stored at slot
//     c ^ ((r >> 1) & 7): fragment reads (ds_read_b128, lane (i, g) -> row i, chunk 2*kstep + g for
//     hi, 4 + 2*kstep + g for lo) are conflict free in each of the hardware's 16-lane service
//     groups and the 8 lanes that stage one row write one full 128-B line.
//   * MFMA operand mapping: v_mfma_f32_32x32x16_f16 wants A[i = lane&31][k = 8*(lane>>5) .. +7]
//     as 8 halfs per lane = exactly one 16-B chunk.
//   * C/D layout (dtype independent on gfx950): col = lane & 31,
//     row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), reg in [0,16).
#pragma once
#include "common.h"

// hooks for tools/micro/gemm_probe.hip (bottleneck isolation); identity in the product build
#ifndef GEMM_PROBE_K
#define GEMM_PROBE_K(k) (k)
#endif
#ifndef GEMM_PROBE_MFMA
#define GEMM_PROBE_MFMA 1
#endif
#ifndef GEMM_PROBE_BARRIER
#define GEMM_PROBE_BARRIER 1      // 0: drop the per-k-tile barrier (wrong results, timing only)
#endif
#ifndef GEMM_PROBE_DMA
#define GEMM_PROBE_DMA 1          // 0: issue the ring's DMAs only in the prologue
#endif
#ifndef GEMM_PROBE_A_EVERY
#define GEMM_PROBE_A_EVERY 1      // n > 1: issue the A-operand DMAs only for every n-th k-tile (emulates a patch-in-LDS conv)
#endif
#ifndef GEMM_PROBE_DSREAD
#define GEMM_PROBE_DSREAD 1       // 0: read the fragments of the first k-tile only
#endif

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t sp_t;            // one dword of an SP tensor (two halfs); a row of K elements is K dwords

// Where the rows of the A operand come from (all SP tensors).
//   * plain:            row r, k-group kg  ->  p0 + r * ld0 + kg * 32         (dwords)
//   * concatenated K:   k >= ksplit reads p1 (cat([x, msg], dim=2) of transformer.py:55 without
//                       materialising the concatenation); same row pitch, ksplit % 32 == 0
//   * gathered rows:    r -> gather[r]  (fine_preprocess.py:51-52 picks coarse features at
//                       (b_ids, i_ids) -- the index already folds b*L + i)
//   * implicit im2col:  (CONV mode of the main loop) row m = output pixel (b, yo, xo) of a KH x KW
//                       convolution over an NHWC SP activation tensor [B, H, W, Cp]; k runs over
//                       (tap, channel); k-tile kt covers channels 32*(kt % (Cp/32)) .. +31 of tap
//                       kt / (Cp/32) and reads pixel (yo*stride - pad + ky, xo*stride - pad + kx), zeros
//                       outside the image -- no unfolded volume is ever materialised.
struct ConvGeom { int H, W, Cp, KH, KW, stride, pad, Ho, Wo; const sp_t* zeros; };   // zeros: >= 16 B of zeros (DMA source of out-of-image taps)
struct ASrc {
  const sp_t* p0; int ld0;
  const sp_t* p1; int ksplit;
  const int64_t* gather;
  ConvGeom cv;
};
__host__ __device__ static inline ASrc asrc_plain(const sp_t* p, int ld) { return ASrc{p, ld, nullptr, 1 << 30, nullptr, {}}; }
__host__ __device__ static inline ASrc asrc_cat(const sp_t* p0, const sp_t* p1, int ld, int ksplit) {
  return ASrc{p0, ld, p1, ksplit, nullptr, {}};
}
__host__ __device__ static inline ASrc asrc_gather(const sp_t* p, int ld, const int64_t* idx) {
  return ASrc{p, ld, nullptr, 1 << 30, idx, {}};
}
__host__ __device__ static inline ASrc asrc_conv(const sp_t* x, const ConvGeom& g) {
  return ASrc{x, g.Cp, nullptr, 1 << 30, nullptr, g};
}

// NS_ = 0: register-staged double buffering (below); NS_ >= 2: NS_-stage LDS ring filled by
// global_load_lds (direct global -> LDS DMA, no staging registers) -- gemm_mainloop_dma.
template <int BM_, int BN_, int WM_, int WN_, int NS_ = 0>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, BK = 32, WM = WM_, WN = WN_, NS = NS_;
  static constexpr int WAVES = WM * WN;
  static constexpr int THREADS = WM * WN * 64;
  static constexpr int WTM = BM / WM, WTN = BN / WN;           // wave sub-tile
  static constexpr int TM = WTM / 32, TN = WTN / 32;           // 32x32 MFMA tiles per wave
  static constexpr int A_F4 = BM * 8 / THREADS;                // 16-B chunks per thread per tile
  static constexpr int B_F4 = BN * 8 / THREADS;
  static constexpr int TILE_A = BM * 128, TILE_B = BN * 128;   // bytes
  static constexpr int STAGE_BYTES = TILE_A + TILE_B;
  static constexpr size_t LDS_BYTES = (NS ? NS : 2) * (size_t)STAGE_BYTES;
  static constexpr int LDS_FLOATS = (int)(LDS_BYTES / 4);
  static constexpr int A_DMA = BM / (8 * WAVES), B_DMA = BN / (8 * WAVES);   // DMA instructions per wave per tile
  static_assert(NS == 0 || (BM % (8 * WAVES) == 0 && BN % (8 * WAVES) == 0), "DMA mapping: 8 rows per wave instruction");
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be a multiple of 32");
  static_assert((BM * 8) % THREADS == 0 && (BN * 8) % THREADS == 0, "loader mapping");
};

// byte offset of 16-B chunk `c` (0..7) of row `r` inside a staged tile
__device__ __forceinline__ int lds_chunk_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// ---- SP format helpers ---------------------------------------------------------------------
// x -> (hi, lo) packed as hi | lo << 16
__device__ __forceinline__ uint32_t sp_pack(float v) {
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  return (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
}
// Two consecutive values -> the hi dword (hi_a | hi_b << 16) and the lo dword of their SP pair.  Written on 2-vectors so that hipcc
// selects gfx950's packed conversions: v_cvt_pk_f16_f32 (round to nearest even, both values), two v_cvt_f32_f16, one v_pk_add_f32,
// v_cvt_pk_f16_f32 -- 5 instructions per PAIR and the results are already packed (the scalar form costs ~6 per VALUE with the
// and / or / shift packing).  Same values as sp_pack() bit for bit.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sp_pack2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const f32x2 v = {a, b};
  const h16x2 h = __builtin_convertvector(v, h16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const h16x2 l = __builtin_convertvector(r, h16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
// The dword a lane must store so that a pair of lanes (even, odd) owning two consecutive columns
// (c, c+1) of one SP row emits: even lane -> hi dword (hi_c | hi_c+1 << 16) at group*32 + c/2,
// odd lane -> lo dword (lo_c | lo_c+1 << 16) at group*32 + 16 + c/2.  32 lanes = one 128-B group.
// Both lanes of a pair must execute this (DPP exchange with lane ^ 1).
__device__ __forceinline__ uint32_t sp_word(float v, bool odd) {
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  const uint32_t hb = __builtin_bit_cast(uint16_t, h), lb = __builtin_bit_cast(uint16_t, l);
  const uint32_t send = odd ? hb : lb;               // what the partner lane needs
  const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, true);   // lane ^ 1
  return odd ? (recv | (lb << 16)) : (hb | (recv << 16));
}
// Same for the 16 accumulator registers of one MFMA tile, in three phases (split / exchange /
// merge) so that the DPP exchanges of different registers issue back to back instead of each
// waiting out its VALU->DPP hazard.
__device__ __forceinline__ void sp_words16(const f32x16& v, bool odd, uint32_t (&w)[16]) {
  uint32_t hb[16], lb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const _Float16 h = (_Float16)v[r];
    const _Float16 l = (_Float16)(v[r] - (float)h);
    hb[r] = __builtin_bit_cast(uint16_t, h);
    lb[r] = __builtin_bit_cast(uint16_t, l);
    w[r] = odd ? hb[r] : lb[r];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) w[r] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[r], 0xB1, 0xF, 0xF, true);
#pragma unroll
  for (int r = 0; r < 16; ++r) w[r] = odd ? (w[r] | (lb[r] << 16)) : (hb[r] | (w[r] << 16));
}
// Inverse of sp_word: the lane loads the dword at sp_index(col) of its row (even lanes the hi pair,
// odd lanes the lo pair) and the pair of lanes exchanges halves.  Both lanes must execute this.
__device__ __forceinline__ float sp_value(uint32_t w, bool odd) {
  const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, true);   // lane ^ 1
  const uint32_t hb = odd ? (recv >> 16) : (w & 0xffffu);
  const uint32_t lb = odd ? (w >> 16) : (recv & 0xffffu);
  return (float)__builtin_bit_cast(_Float16, (uint16_t)hb) + (float)__builtin_bit_cast(_Float16, (uint16_t)lb);
}
// Eight consecutive channels (8q .. 8q+7 of one 32-channel group) <-> the 16-B hi chunk (dwords 4q .. 4q+3
// of the group) and the 16-B lo chunk (dwords 16+4q ..): the form used by the streaming (non-GEMM) kernels,
// one thread per octet, 16-B accesses, no cross-lane traffic.
__device__ __forceinline__ void sp_pack8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t a = sp_pack(v[2 * e]), b = sp_pack(v[2 * e + 1]);
    hi[e] = (a & 0xffffu) | (b << 16);
    lo[e] = (a >> 16) | (b & 0xffff0000u);
  }
}
__device__ __forceinline__ void sp_unpack8(const u32x4& hi, const u32x4& lo, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = (float)__builtin_bit_cast(_Float16, (uint16_t)(hi[e] & 0xffffu)) +
               (float)__builtin_bit_cast(_Float16, (uint16_t)(lo[e] & 0xffffu));
    v[2 * e + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(hi[e] >> 16)) +
                   (float)__builtin_bit_cast(_Float16, (uint16_t)(lo[e] >> 16));
  }
}
// dword offset of the hi chunk of channel octet `oct` (= channel / 8) inside an SP row; the lo chunk is 16 dwords later
__device__ __forceinline__ int sp_octet_off(int oct) { return (oct >> 2) * 32 + (oct & 3) * 4; }
// dword index inside an SP row of the word sp_word() produces for column `col`
__device__ __forceinline__ int sp_index(int col) { return (col & ~31) + ((col & 1) ? 16 : 0) + ((col & 31) >> 1); }
// convenience: predicated single store (row_ptr = first dword of the row)
__device__ __forceinline__ void sp_store(sp_t* row_ptr, int col, float v, bool pred) {
  const uint32_t w = sp_word(v, col & 1);
  if (pred) row_ptr[sp_index(col)] = w;
}

// Epilogue addressing.  `base[idx]` with a 32-bit index still costs a 64-bit multiply-add per access (the scale by
// sizeof(T) may carry into bit 32): hipcc emits v_mad_u64_u32 / v_lshl_add_u64 per store.  A 32-bit BYTE offset added to
// a block-uniform base pointer maps onto the global_load / global_store "SGPR base + 32-bit VGPR offset" form: one
// v_add_u32 (or none, if the compiler folds constants) per access.  Tiles are < 4 GB, so byte offsets fit 32 bits.
#ifdef LOFTR_EPI_OLD      // A/B switch (tools/gpu/ab_lib.sh): element indexing as before
template <typename T> __device__ __forceinline__ void st_off(T* base, unsigned byte_off, T v) { base[byte_off / (unsigned)sizeof(T)] = v; }
template <typename T> __device__ __forceinline__ T ld_off(const T* base, unsigned byte_off) { return base[byte_off / (unsigned)sizeof(T)]; }
#else
template <typename T> __device__ __forceinline__ void st_off(T* base, unsigned byte_off, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
template <typename T> __device__ __forceinline__ T ld_off(const T* base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
#endif

// Wave w of a workgroup owns sub-tile (wm, wn) = (w % WM, w / WM): consecutive waves -- which the
// hardware spreads over the four SIMDs -- walk down the M direction first, so the two waves sharing a SIMD
// in an 8-wave workgroup sit in different column strips (balances the matrix pipes when the last
// column strip of a tile is only partly active, `nact` below).
// Per-lane coordinates of the accumulators inside the BM x BN block tile (C/D layout above):
//   element (i, j, r) -> tile row  lrow + i*32 + (r&3) + 8*(r>>2),  tile column  lcol + j*32.
template <typename Cfg>
struct EpiLane {
  int lrow, lcol, spcol; bool odd;
  __device__ __forceinline__ EpiLane() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lrow = (wave % Cfg::WM) * Cfg::WTM + 4 * (lane >> 5);
    lcol = (wave / Cfg::WM) * Cfg::WTN + (lane & 31);
    odd = lane & 1;
    spcol = (wave / Cfg::WM) * Cfg::WTN + (odd ? 16 : 0) + ((lane & 31) >> 1);   // + j*32
  }
  static __device__ __forceinline__ constexpr int rr(int i, int r) { return i * 32 + (r & 3) + 8 * (r >> 2); }
};

// Runs the whole K loop for the block tile at (m0, n0).  M, N are the valid extents (rows beyond
// them are clamped on load -- the caller masks them in its epilogue).  K % 32 == 0 (SP groups).
template <typename Cfg, bool CONV = false>
__device__ __forceinline__ void gemm_mainloop_regs(const ASrc& a, const sp_t* __restrict__ Bp, int ldb,
                                              int M, int N, int K, int m0, int n0,
                                              float* lds_f, f32x16 (&acc)[Cfg::TM][Cfg::TN], int ncols = 1 << 30) {
  constexpr int BK = Cfg::BK;
  constexpr int TM = Cfg::TM, TN = Cfg::TN;
  char* lds = reinterpret_cast<char*>(lds_f);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave % Cfg::WM, wn = wave / Cfg::WM;
  const int nact = min(TN, (ncols - (n0 + wn * Cfg::WTN) + 31) / 32);   // active 32-column MFMA tiles (wave-uniform)

  // ---- loader: per-thread dword offsets of its 16-B chunks (32-bit: operands are < 2^31 dwords)
  int aoff[Cfg::A_F4], boff[Cfg::B_F4];
  int a_lds_off[Cfg::A_F4], b_lds_off[Cfg::B_F4];
  int ayx[Cfg::A_F4];                             // CONV: (yo*stride - pad) << 16 | (xo*stride - pad) & 0xffff
  const int kc = tid & 7;                         // this thread's chunk inside a 128-B row group
#pragma unroll
  for (int i = 0; i < Cfg::A_F4; ++i) {
    const int r = (tid + i * Cfg::THREADS) >> 3;
    const int gr = min(m0 + r, M - 1);
    if (CONV) {
      const int hw = a.cv.Ho * a.cv.Wo;
      const int b = gr / hw, rem = gr - b * hw;
      const int yo = rem / a.cv.Wo, xo = rem - yo * a.cv.Wo;
      const int yb = yo * a.cv.stride - a.cv.pad, xb = xo * a.cv.stride - a.cv.pad;
      ayx[i] = (yb << 16) | (xb & 0xffff);
      aoff[i] = ((b * a.cv.H + yb) * a.cv.W + xb) * a.cv.Cp + kc * 4;      // tap (0,0); may point before the image
    } else {
      const int row = a.gather ? (int)a.gather[gr] : gr;
      aoff[i] = row * a.ld0 + kc * 4;
    }
    a_lds_off[i] = lds_chunk_off(r, kc);
  }
  const int cv_gpt = CONV ? a.cv.Cp / 32 : 1;     // k-tiles per filter tap
#pragma unroll
  for (int i = 0; i < Cfg::B_F4; ++i) {
    const int r = (tid + i * Cfg::THREADS) >> 3;
    const int gr = min(n0 + r, N - 1);
    boff[i] = gr * ldb + kc * 4;
    b_lds_off[i] = Cfg::TILE_A + lds_chunk_off(r, kc);
  }

  u32x4 ra0[Cfg::A_F4], ra1[Cfg::A_F4], rb[Cfg::B_F4];
  // (macros, not lambdas: capturing the staging registers by reference makes hipcc keep a
  //  scratch copy of them)
#define GEMM_LOAD_A(k0_, ra_)                                                                 \
  { if (CONV) {                                                                               \
    const int kt__ = (k0_) / BK;                    /* all block-uniform scalars */           \
    const int tap__ = kt__ / cv_gpt, cg__ = kt__ - tap__ * cv_gpt;                            \
    const int ky__ = tap__ / a.cv.KW, kx__ = tap__ - ky__ * a.cv.KW;                          \
    const int toff__ = (ky__ * a.cv.W + kx__) * a.cv.Cp + cg__ * 32;                          \
    _Pragma("unroll") for (int i = 0; i < Cfg::A_F4; ++i) {                                   \
      const int y__ = (ayx[i] >> 16) + ky__, x__ = (int)(short)(ayx[i] & 0xffff) + kx__;      \
      const bool in__ = (unsigned)y__ < (unsigned)a.cv.H && (unsigned)x__ < (unsigned)a.cv.W; \
      ra_[i] = in__ ? *reinterpret_cast<const u32x4*>(a.p0 + (aoff[i] + toff__)) : u32x4{0u, 0u, 0u, 0u}; \
    }                                                                                         \
  } else {                                                                                    \
    const int k0__ = (k0_);                                                                   \
    const bool second__ = k0__ >= a.ksplit; /* block-uniform */                               \
    const sp_t* ap__ = second__ ? a.p1 : a.p0;                                                \
    const int ka__ = second__ ? k0__ - a.ksplit : k0__;                                       \
    _Pragma("unroll") for (int i = 0; i < Cfg::A_F4; ++i)                                     \
      ra_[i] = *reinterpret_cast<const u32x4*>(ap__ + (aoff[i] + GEMM_PROBE_K(ka__)));       \
  } }
#define GEMM_LOAD_B(k0_)                                                                      \
  {                                                                                           \
    const int k0__ = (k0_);                                                                   \
    _Pragma("unroll") for (int i = 0; i < Cfg::B_F4; ++i)                                     \
      rb[i] = *reinterpret_cast<const u32x4*>(Bp + (boff[i] + GEMM_PROBE_K(k0__)));          \
  }
#define GEMM_STORE_TILE(buf_, ra_)                                                            \
  {                                                                                           \
    char* s__ = lds + (buf_) * Cfg::STAGE_BYTES;                                              \
    _Pragma("unroll") for (int i = 0; i < Cfg::A_F4; ++i)                                     \
      *reinterpret_cast<u32x4*>(s__ + a_lds_off[i]) = ra_[i];                                \
    _Pragma("unroll") for (int i = 0; i < Cfg::B_F4; ++i)                                     \
      *reinterpret_cast<u32x4*>(s__ + b_lds_off[i]) = rb[i];                                 \
  }
  // all MFMAs of one k-tile held in LDS stage buf_
#define GEMM_COMPUTE_TILE(buf_, FULL_)                                                             \
  {                                                                                           \
    const char* sA__ = lds + (buf_) * Cfg::STAGE_BYTES;                                       \
    const char* sB__ = sA__ + Cfg::TILE_A;                                                    \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                  \
      h16x8 ah[TM], al[TM], bh[TN], bl[TN];                                                   \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                        \
        ah[i] = *reinterpret_cast<const h16x8*>(sA__ + lds_chunk_off(a_r0 + i * 32, ks * 2 + g));     \
        al[i] = *reinterpret_cast<const h16x8*>(sA__ + lds_chunk_off(a_r0 + i * 32, 4 + ks * 2 + g)); \
      }                                                                                       \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                        \
        bh[j] = *reinterpret_cast<const h16x8*>(sB__ + lds_chunk_off(b_r0 + j * 32, ks * 2 + g));     \
        bl[j] = *reinterpret_cast<const h16x8*>(sB__ + lds_chunk_off(b_r0 + j * 32, 4 + ks * 2 + g)); \
      }                                                                                       \
      /* the two cross terms first, the leading term last; TM*TN independent accumulators */  \
      /* between two MFMAs on the same accumulator */                                         \
      if (GEMM_PROBE_MFMA) {                                                                  \
      if (FULL_) {               /* all column tiles live: straight-line MFMA block */        \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                        \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                        \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                        \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
      } else {                   /* wave-uniform: skip dead 32-column tiles */                \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) if (j < nact) {                        \
          _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
          }                                                                                   \
        }                                                                                     \
      }                                                                                       \
      } else {  /* probe: keep the LDS reads alive without the matrix pipe */                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                        \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
            acc[i][j][0] += (float)ah[i][0] + (float)al[i][1] + (float)bh[j][2] + (float)bl[j][3]; \
      }                                                                                       \
    }                                                                                         \
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment rows of this lane inside the A / B tiles
  const int g = lane >> 5;
  const int a_r0 = wm * Cfg::WTM + (lane & 31);
  const int b_r0 = wn * Cfg::WTN + (lane & 31);

  const int nk = K / BK;
  GEMM_LOAD_A(0, ra0);
  GEMM_LOAD_B(0);
  if (nk > 1) GEMM_LOAD_A(BK, ra1);
  GEMM_STORE_TILE(0, ra0);
  __syncthreads();
  if (nact == TN) {          // k loop instantiated twice: the common all-tiles-live case stays branch free
  for (int kt = 0; kt < nk; kt += 2) {
    // even k-tile: in LDS stage 0; A of tile kt+1 is landing in register set 1; set 0 is free
    if (kt + 2 < nk) GEMM_LOAD_A((kt + 2) * BK, ra0);
    if (kt + 1 < nk) GEMM_LOAD_B((kt + 1) * BK);
    GEMM_COMPUTE_TILE(0, 1);
    if (kt + 1 >= nk) break;
    GEMM_STORE_TILE(1, ra1);              // stage 1 was last read in iteration kt-1 (barrier since)
    __syncthreads();
    // odd k-tile: in LDS stage 1; A of tile kt+2 is landing in register set 0; set 1 is free
    if (kt + 3 < nk) GEMM_LOAD_A((kt + 3) * BK, ra1);
    if (kt + 2 < nk) GEMM_LOAD_B((kt + 2) * BK);
    GEMM_COMPUTE_TILE(1, 1);
    if (kt + 2 < nk) {
      GEMM_STORE_TILE(0, ra0);
      __syncthreads();
    }
  }
  } else {
  for (int kt = 0; kt < nk; kt += 2) {
    // even k-tile: in LDS stage 0; A of tile kt+1 is landing in register set 1; set 0 is free
    if (kt + 2 < nk) GEMM_LOAD_A((kt + 2) * BK, ra0);
    if (kt + 1 < nk) GEMM_LOAD_B((kt + 1) * BK);
    GEMM_COMPUTE_TILE(0, 0);
    if (kt + 1 >= nk) break;
    GEMM_STORE_TILE(1, ra1);              // stage 1 was last read in iteration kt-1 (barrier since)
    __syncthreads();
    // odd k-tile: in LDS stage 1; A of tile kt+2 is landing in register set 0; set 1 is free
    if (kt + 3 < nk) GEMM_LOAD_A((kt + 3) * BK, ra1);
    if (kt + 2 < nk) GEMM_LOAD_B((kt + 2) * BK);
    GEMM_COMPUTE_TILE(1, 0);
    if (kt + 2 < nk) {
      GEMM_STORE_TILE(0, ra0);
      __syncthreads();
    }
  }
  }
#undef GEMM_LOAD_A
#undef GEMM_LOAD_B
#undef GEMM_COMPUTE_TILE
#undef GEMM_STORE_TILE
}

// ---- DMA variant -----------------------------------------------------------------------------
// NS-stage LDS ring; every k-tile is brought in by global_load_lds_dwordx4 (64 lanes x 16 B = 8 tile
// rows of 128 B per instruction; the LDS destination is wave-uniform base + lane*16, so the XOR
// swizzle is applied to the per-lane GLOBAL address instead).  Tile t+NS-1 is issued right after the
// barrier that starts k-tile t, i.e. NS-1 tiles (several microseconds of matrix work) ahead of its
// use -- the register-staged loop could only afford one tile of lookahead for B and two for A, and
// measured latency-bound (profiles/, r01 v4).  One s_barrier per k-tile, explicit vmcnt waits.
#define LOFTR_WAITCNT_VM(n_) __builtin_amdgcn_s_waitcnt((((n_) & 0xF) | (((n_) >> 4) << 14) | (0x7 << 4) | (0xF << 8)))
template <typename Cfg, bool CONV = false>
__device__ __forceinline__ void gemm_mainloop_dma(const ASrc& a, const sp_t* __restrict__ Bp, int ldb,
                                                  int M, int N, int K, int m0, int n0,
                                                  float* lds_f, f32x16 (&acc)[Cfg::TM][Cfg::TN], int ncols = 1 << 30) {
  constexpr int BK = Cfg::BK, NS = Cfg::NS, WAVES = Cfg::WAVES;
  constexpr int TM = Cfg::TM, TN = Cfg::TN;
  constexpr int PER_TILE = Cfg::A_DMA + Cfg::B_DMA;            // DMA instructions per wave per k-tile
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  char* lds = reinterpret_cast<char*>(lds_f);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave % Cfg::WM, wn = wave / Cfg::WM;
  const int nact = min(TN, (ncols - (n0 + wn * Cfg::WTN) + 31) / 32);   // active 32-column MFMA tiles (wave-uniform)
  const int rsub = lane >> 3, slot = lane & 7;                 // row inside the 8-row group, LDS slot

  int aoff[Cfg::A_DMA], boff[Cfg::B_DMA], ayx[Cfg::A_DMA];
#pragma unroll
  for (int q = 0; q < Cfg::A_DMA; ++q) {
    const int r = (q * WAVES + wave) * 8 + rsub;
    const int chunk = slot ^ ((r >> 1) & 7);                   // which 16-B chunk of the row lands in this slot
    const int gr = min(m0 + r, M - 1);
    if (CONV) {
      const int hw = a.cv.Ho * a.cv.Wo;
      const int b = gr / hw, rem = gr - b * hw;
      const int yo = rem / a.cv.Wo, xo = rem - yo * a.cv.Wo;
      const int yb = yo * a.cv.stride - a.cv.pad, xb = xo * a.cv.stride - a.cv.pad;
      ayx[q] = (yb << 16) | (xb & 0xffff);
      aoff[q] = ((b * a.cv.H + yb) * a.cv.W + xb) * a.cv.Cp + chunk * 4;
    } else {
      const int row = a.gather ? (int)a.gather[gr] : gr;
      aoff[q] = row * a.ld0 + chunk * 4;
      ayx[q] = 0;
    }
  }
#pragma unroll
  for (int q = 0; q < Cfg::B_DMA; ++q) {
    const int r = (q * WAVES + wave) * 8 + rsub;
    const int chunk = slot ^ ((r >> 1) & 7);
    boff[q] = min(n0 + r, N - 1) * ldb + chunk * 4;
  }
  const int cv_gpt = CONV ? a.cv.Cp / 32 : 1;
  int cv_cg = 0, cv_ky = 0, cv_kx = 0, cv_toff = 0;             // CONV: position of the next k-tile to issue (uniform)

#define GEMM_ISSUE(kt_, stage_)                                                               \
  {                                                                                           \
    char* s__ = lds + (stage_) * Cfg::STAGE_BYTES + wave * 1024;                              \
    const int k0__ = (kt_) * BK;                                                              \
    if (CONV) {              /* k-tiles are issued in increasing order: (ky, kx, channel group) advance incrementally */ \
      _Pragma("unroll") for (int q = 0; q < Cfg::A_DMA; ++q) {                                \
        const int y__ = (ayx[q] >> 16) + cv_ky, x__ = (int)(short)(ayx[q] & 0xffff) + cv_kx;  \
        const bool in__ = (unsigned)y__ < (unsigned)a.cv.H && (unsigned)x__ < (unsigned)a.cv.W; \
        const sp_t* g__ = in__ ? a.p0 + (aoff[q] + cv_toff) : a.cv.zeros;                     \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g__, (lds_ptr_t)(s__ + q * WAVES * 1024), 16, 0, 0); \
      }                                                                                       \
      cv_toff += 32;                                                                          \
      if (++cv_cg == cv_gpt) {                                                                \
        cv_cg = 0;              /* cv_toff is now (ky*W + kx + 1) * Cp: the next tap */        \
        if (++cv_kx == a.cv.KW) { cv_kx = 0; ++cv_ky; cv_toff += (a.cv.W - a.cv.KW) * a.cv.Cp; } \
      }                                                                                       \
    } else if (GEMM_PROBE_A_EVERY == 1 || (kt_) % GEMM_PROBE_A_EVERY == 0) {                   \
      const bool second__ = k0__ >= a.ksplit; /* block-uniform */                             \
      const sp_t* ap__ = second__ ? a.p1 : a.p0;                                              \
      const int ka__ = second__ ? k0__ - a.ksplit : k0__;                                     \
      _Pragma("unroll") for (int q = 0; q < Cfg::A_DMA; ++q)                                  \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ap__ + (aoff[q] + ka__)),                \
                                         (lds_ptr_t)(s__ + q * WAVES * 1024), 16, 0, 0);     \
    } else {  /* probe: keep the per-tile DMA count (vmcnt bookkeeping) with a single cheap load */ \
      _Pragma("unroll") for (int q = 0; q < Cfg::A_DMA; ++q)                                  \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a.p0), (lds_ptr_t)(s__ + q * WAVES * 1024), 4, 0, 0); \
    }                                                                                         \
    _Pragma("unroll") for (int q = 0; q < Cfg::B_DMA; ++q)                                    \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Bp + (boff[q] + k0__)),                    \
                                       (lds_ptr_t)(s__ + Cfg::TILE_A + q * WAVES * 1024), 16, 0, 0); \
  }
#define GEMM_COMPUTE_STAGE(stage_, FULL_)                                                            \
  {                                                                                           \
    const char* sA__ = lds + (stage_) * Cfg::STAGE_BYTES;                                     \
    const char* sB__ = sA__ + Cfg::TILE_A;                                                    \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                  \
      h16x8 ah[TM], al[TM], bh[TN], bl[TN];                                                   \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                        \
        ah[i] = *reinterpret_cast<const h16x8*>(sA__ + lds_chunk_off(a_r0 + i * 32, ks * 2 + g));     \
        al[i] = *reinterpret_cast<const h16x8*>(sA__ + lds_chunk_off(a_r0 + i * 32, 4 + ks * 2 + g)); \
      }                                                                                       \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                        \
        bh[j] = *reinterpret_cast<const h16x8*>(sB__ + lds_chunk_off(b_r0 + j * 32, ks * 2 + g));     \
        bl[j] = *reinterpret_cast<const h16x8*>(sB__ + lds_chunk_off(b_r0 + j * 32, 4 + ks * 2 + g)); \
      }                                                                                       \
      if (FULL_) {               /* all column tiles live: straight-line MFMA block */        \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                        \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                        \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                        \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
      } else {                   /* wave-uniform: skip dead 32-column tiles */                \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) if (j < nact) {                        \
          _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
          }                                                                                   \
        }                                                                                     \
      }                                                                                       \
    }                                                                                         \
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int g = lane >> 5;
  const int a_r0 = wm * Cfg::WTM + (lane & 31);
  const int b_r0 = wn * Cfg::WTN + (lane & 31);

  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) GEMM_ISSUE(s, s);
  int stage = 0, istage = NS - 1;                     // stage of tile kt / of the tile issued in iteration kt
  if (nact == TN) {          // k loop instantiated twice: the common all-tiles-live case stays branch free
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most the newer tiles' DMAs are still outstanding
    const int newer = min(NS - 2, nk - 1 - kt);
    if (NS >= 4 && newer >= 2) LOFTR_WAITCNT_VM(2 * PER_TILE);
    else if (newer >= 1) LOFTR_WAITCNT_VM(PER_TILE);
    else LOFTR_WAITCNT_VM(0);
    if (GEMM_PROBE_BARRIER) __builtin_amdgcn_s_barrier();   // ... for every wave's share of it; also: all waves are done reading stage istage
    if (GEMM_PROBE_DMA && kt + NS - 1 < nk) GEMM_ISSUE(kt + NS - 1, istage);
    GEMM_COMPUTE_STAGE(GEMM_PROBE_DSREAD ? stage : 0, 1);
    stage = stage + 1 == NS ? 0 : stage + 1;
    istage = istage + 1 == NS ? 0 : istage + 1;
  }
  } else {
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most the newer tiles' DMAs are still outstanding
    const int newer = min(NS - 2, nk - 1 - kt);
    if (NS >= 4 && newer >= 2) LOFTR_WAITCNT_VM(2 * PER_TILE);
    else if (newer >= 1) LOFTR_WAITCNT_VM(PER_TILE);
    else LOFTR_WAITCNT_VM(0);
    if (GEMM_PROBE_BARRIER) __builtin_amdgcn_s_barrier();   // ... for every wave's share of it; also: all waves are done reading stage istage
    if (GEMM_PROBE_DMA && kt + NS - 1 < nk) GEMM_ISSUE(kt + NS - 1, istage);
    GEMM_COMPUTE_STAGE(GEMM_PROBE_DSREAD ? stage : 0, 0);
    stage = stage + 1 == NS ? 0 : stage + 1;
    istage = istage + 1 == NS ? 0 : istage + 1;
  }
  }
  __builtin_amdgcn_s_barrier();                       // callers reuse the LDS in their epilogues
#undef GEMM_ISSUE
#undef GEMM_COMPUTE_STAGE
}

template <typename Cfg, bool CONV = false>
__device__ __forceinline__ void gemm_mainloop(const ASrc& a, const sp_t* __restrict__ Bp, int ldb,
                                              int M, int N, int K, int m0, int n0,
                                              float* lds_f, f32x16 (&acc)[Cfg::TM][Cfg::TN], int ncols = 1 << 30) {
  // ncols: columns >= ncols are never used by the caller -> their MFMAs are skipped (32-column granularity)
  if constexpr (Cfg::NS >= 2) gemm_mainloop_dma<Cfg, CONV>(a, Bp, ldb, M, N, K, m0, n0, lds_f, acc, ncols);
  else gemm_mainloop_regs<Cfg, CONV>(a, Bp, ldb, M, N, K, m0, n0, lds_f, acc, ncols);
}

// Coordinates of accumulator element `reg` of MFMA tile (i, j) for this lane.
template <typename Cfg>
__device__ __forceinline__ int acc_row(int m0, int i, int reg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return m0 + (wave % Cfg::WM) * Cfg::WTM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
template <typename Cfg>
__device__ __forceinline__ int acc_col(int n0, int j) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return n0 + (wave / Cfg::WM) * Cfg::WTN + j * 32 + (lane & 31);
}

// ---- fp32 -> SP conversion of whole tensors (sp_convert.hip) --------------------------------
// Up to SP_MAX_JOBS independent [rows, K] tensors per launch (all weights of a transformer in one).
// K is padded up to a multiple of 32 with zeros in the destination (dst row pitch = ceil32(K)).
constexpr int SP_MAX_JOBS = 64;
struct SpJobs {
  const float* src[SP_MAX_JOBS];
  sp_t* dst[SP_MAX_JOBS];
  int rows[SP_MAX_JOBS], K[SP_MAX_JOBS], ld[SP_MAX_JOBS];     // ld = source row pitch in floats
  float* inv_scale[SP_MAX_JOBS];   // null: stored as is.  Else float[rows]: every row is stored multiplied by the power of two
                                   // that lifts its maximum to [2^13, 2^14) and the INVERSE factor is written here (gemm.h header)
  const float* tensor_inv[SP_MAX_JOBS];   // null, or a device scalar: the whole tensor is stored multiplied by 1 / *tensor_inv
  int n;
  SpJobs() : n(0) { for (int i = 0; i < SP_MAX_JOBS; ++i) { inv_scale[i] = nullptr; tensor_inv[i] = nullptr; } }
};
// power of two s with max * s in [2^13, 2^14) (1 for an all-zero row); returns s, *inv = 1 / s (both exact)
__host__ __device__ static inline float sp_row_scale(float absmax, float* inv) {
  if (!(absmax > 0.f) || !(absmax < 3.0e38f)) { *inv = 1.f; return 1.f; }
  int e;
  (void)frexpf(absmax, &e);                 // absmax = m * 2^e, m in [0.5, 1)
  int sh = 14 - e;                          // m * 2^14 in [2^13, 2^14)
  sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
  *inv = ldexpf(1.f, -sh);
  return ldexpf(1.f, sh);
}
static inline int ceil32(int k) { return (k + 31) / 32 * 32; }
int launch_sp_convert(const SpJobs& jobs, hipStream_t st);
int launch_sp_convert1(const float* src, int ld, sp_t* dst, long rows, int K, hipStream_t st, float* tensor_inv_out = nullptr);
