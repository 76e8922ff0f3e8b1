// fp32-accurate "NT" GEMM main loop on the CDNA4 matrix cores:  acc[m][n] += sum_k A[m][k] * B[n][k]
//
// Precision design ("f16x3 split").  The dual-softmax confidences must match the reference within
// 1e-4 and the sub-pixel key points within 1e-3 px; bf16/fp16/tf32-rounded operands miss that by
// 10-100x, and gfx950 has no xf32.  The fp32-input MFMA (v_mfma_f32_32x32x2_f32) is
// exact but runs at the fp32 VECTOR rate (157 TFLOP/s), 1/16 of the fp16 matrix rate.  Instead every
// fp32 operand x is split on the way into LDS into two fp16 numbers
//        hi = fp16(x),   lo = fp16(x - hi)          (x - hi is exact in fp32; |x - hi - lo| <= 2^-22 |x|)
// and each product is evaluated as three fp16 MFMAs with fp32 accumulation,
//        a*b  ~=  hi_a*hi_b + hi_a*lo_b + lo_a*hi_b          (dropped: lo_a*lo_b ~ 2^-22 |a*b|)
// i.e. ~22 mantissa bits per product at 3/16 of the cost of the fp32 MFMA (5.3x its peak).  Every
// fp16 x fp16 product is exact in fp32, so the only extra error over an fp32 fma chain is the
// 2^-22-relative representation / dropped-term error: measured end-to-end through the 8 coarse
// layers + dual-softmax it moves conf by 1.4e-5 vs an fp64 run (fp32 chain: 1.9e-5), DESIGN.md §5.
// Range: |x| must stay below the fp16 maximum 65504 (LayerNorm-bounded activations and weights are
// O(1)); tiny values degrade gracefully (absolute error <= 2^-25 through fp16 subnormals, which
// the MFMA does not flush -- tools/micro/f16_denorm.hip).
//
// Tiling (per workgroup of WM*WN waves of 64 lanes):
//   * block tile BM x BN, k-step BK = 32; each wave owns a (BM/WM) x (BN/WN) sub-tile made of
//     TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each), v_mfma_f32_32x32x16_f16.
//   * A and B tiles are staged global(fp32) -> registers -> split -> LDS(fp16 hi / lo planes),
//     LDS double buffered, registers double buffered on top: the global loads of tile t+2 are
//     issued before the MFMAs of tile t, tile t+1 is converted + written to LDS after them (one
//     barrier per k-tile).
//   * LDS rows hold the 32 k-values of one tile row as 4 chunks of 8 halfs (16 B); rows are 64 B
//     with NO padding, chunk c of row r lives at slot c ^ ((r >> 2) & 3).  A fragment read
//     (ds_read_b128, lane (i, g) -> row i, chunk 2*kstep + g) is then conflict free in each of
//     the hardware's 16-lane service groups, and the staging ds_write_b64 of 8 consecutive lanes
//     fills one row.
//   * MFMA operand mapping: v_mfma_f32_32x32x16_f16 wants A[i = lane&31][k = 8*(lane>>5) .. +7]
//     as 8 halfs per lane = exactly one 16-B chunk.
//   * C/D layout (dtype independent on gfx950): col = lane & 31,
//     row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), reg in [0,16).
#pragma once
#include "common.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Where the rows of the A operand come from.
//   * plain:            row r, k  ->  p0[r * ld0 + k]
//   * concatenated K:   k >= ksplit reads p1[r * ld1 + (k - ksplit)]   (cat([x, msg], dim=2) of
//                       transformer.py:55 without materialising the concatenation)
//   * gathered rows:    r -> gather[r]  (fine_preprocess.py:51-52 picks coarse features at
//                       (b_ids, i_ids) -- the index already folds b*L + i)
struct ASrc {
  const float* p0; int ld0;
  const float* p1; int ld1; int ksplit;     // ld1 must equal ld0
  const int64_t* gather;
};
__host__ __device__ static inline ASrc asrc_plain(const float* p, int ld) { return ASrc{p, ld, nullptr, 0, 1 << 30, nullptr}; }
__host__ __device__ static inline ASrc asrc_cat(const float* p0, int ld0, const float* p1, int ld1, int ksplit) {
  return ASrc{p0, ld0, p1, ld1, ksplit, nullptr};
}
__host__ __device__ static inline ASrc asrc_gather(const float* p, int ld, const int64_t* idx) {
  return ASrc{p, ld, nullptr, 0, 1 << 30, idx};
}

template <int BM_, int BN_, int WM_, int WN_>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, BK = 32, WM = WM_, WN = WN_;
  static constexpr int THREADS = WM * WN * 64;
  static constexpr int WTM = BM / WM, WTN = BN / WN;           // wave sub-tile
  static constexpr int TM = WTM / 32, TN = WTN / 32;           // 32x32 MFMA tiles per wave
  static constexpr int KCH = BK / 4;                           // float4 chunks per row (global side)
  static constexpr int A_F4 = BM * KCH / THREADS;              // float4 per thread per tile
  static constexpr int B_F4 = BN * KCH / THREADS;
  static constexpr int ROW_BYTES = BK * 2;                     // one plane (hi or lo) of one row: 64 B
  static constexpr int PLANE_A = BM * ROW_BYTES, PLANE_B = BN * ROW_BYTES;
  static constexpr int STAGE_BYTES = 2 * (PLANE_A + PLANE_B);  // [A hi | A lo | B hi | B lo]
  static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE_BYTES; // double buffered
  static constexpr int LDS_FLOATS = (int)(LDS_BYTES / 4);
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be a multiple of 32");
  static_assert((BM * KCH) % THREADS == 0 && (BN * KCH) % THREADS == 0, "loader mapping");
};

// byte offset of 16-B chunk `c` (0..3) of row `r` inside a plane
__device__ __forceinline__ int lds_chunk_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

// x -> (hi, lo) fp16 pair, 4 values at a time
__device__ __forceinline__ void split4(const f32x4& v, h16x4& hi, h16x4& lo) {
  hi = __builtin_convertvector(v, h16x4);                       // v_cvt_pk_f16_f32 (round to nearest even)
  const f32x4 back = __builtin_convertvector(hi, f32x4);
  lo = __builtin_convertvector(v - back, h16x4);
}

// Optional transform of the A operand between its global load and the fp16 split.  The default
// does nothing; AttnXform (below) folds the linear-attention normaliser into the merge GEMM.
struct NoXform {
  __device__ __forceinline__ void fetch(int, int) {}
  template <int NA> __device__ __forceinline__ void apply(f32x4 (&)[NA], int) {}
};

// sum over the 8 consecutive lanes that hold one 32-float tile row (DPP only, no LDS traffic)
__device__ __forceinline__ float sum8_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}

// Linear attention folded into the merge projection (linear_attention.py:44-45 + transformer.py:51):
//   merge(message)[l,:] = sum_h z[l,h] * (Q[l,h,:] @ KV_h) @ Wm[:, h-block]^T
//                       = (z (.) Q)[l,:] @ P,   P[(h,d), j] = sum_v KV[h,d,v] Wm[j, h*32+v]
// with z[l,h] = S / (Q[l,h,:] . Ksum[h,:] + eps).  D = 32 = BK, so k-tile t IS head t: the 32 Q
// values of a row sit in 8 consecutive lanes (one float4 each) and z is a DPP reduction away.
struct AttnXform {
  const float* kv;        // [H][33][32] of this batch element: row 32 of each head = Ksum
  float v_length, eps;
  f32x4 ks0, ks1;         // one per in-flight register set of the main loop
  __device__ __forceinline__ void fetch(int k0, int slot) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(kv + (k0 >> 5) * (33 * 32) + 32 * 32 + (threadIdx.x & 7) * 4);
    if (slot) ks1 = v; else ks0 = v;
  }
  template <int NA> __device__ __forceinline__ void apply(f32x4 (&ra)[NA], int slot) {
    const f32x4 ks = slot ? ks1 : ks0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float p = ra[i].x * ks.x + ra[i].y * ks.y + ra[i].z * ks.z + ra[i].w * ks.w;
      p = sum8_dpp(p);
      ra[i] *= v_length / (p + eps);
    }
  }
};

// Runs the whole K loop for the block tile at (m0, n0).  M, N are the valid extents (rows beyond
// them are clamped on load -- the caller masks them in its epilogue).  K % 4 == 0; k beyond K reads
// as zero.
template <typename Cfg, typename AX>
__device__ __forceinline__ void gemm_mainloop(const ASrc& a, const float* __restrict__ Bp, int ldb,
                                              int M, int N, int K, int m0, int n0,
                                              float* lds_f, f32x16 (&acc)[Cfg::TM][Cfg::TN], AX& ax) {
  constexpr int BK = Cfg::BK;
  constexpr int TM = Cfg::TM, TN = Cfg::TN, KCH = Cfg::KCH;
  char* lds = reinterpret_cast<char*>(lds_f);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;

  // ---- loader: per-thread element offsets of its tile rows (32-bit: operands are < 2^31 elements)
  int aoff[Cfg::A_F4], boff[Cfg::B_F4];
  int a_lds_off[Cfg::A_F4], b_lds_off[Cfg::B_F4];
  const int kc = tid % KCH;                       // this thread's float4 column inside a tile row
#pragma unroll
  for (int i = 0; i < Cfg::A_F4; ++i) {
    const int r = (tid + i * Cfg::THREADS) / KCH;
    const int gr = min(m0 + r, M - 1);
    const int row = a.gather ? (int)a.gather[gr] : gr;
    aoff[i] = row * a.ld0 + kc * 4;               // ld1 == ld0 for the concatenated source
    a_lds_off[i] = lds_chunk_off(r, kc >> 1) + (kc & 1) * 8;
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_F4; ++i) {
    const int r = (tid + i * Cfg::THREADS) / KCH;
    const int gr = min(n0 + r, N - 1);
    boff[i] = gr * ldb + kc * 4;
    b_lds_off[i] = lds_chunk_off(r, kc >> 1) + (kc & 1) * 8;
  }

  // The A operand streams from HBM: its global loads run TWO k-tiles ahead of the MFMAs (two
  // register sets).  The B operand (weights / one pair's descriptors) is L2 resident and shared by
  // every workgroup: one tile ahead is enough, one register set.  The fp16 split + LDS store of
  // tile t+1 happens after the MFMAs of tile t.
  f32x4 ra0[Cfg::A_F4], ra1[Cfg::A_F4], rb[Cfg::B_F4];
  // (macros, not lambdas: capturing the staging registers by reference makes hipcc keep a
  //  scratch copy of them)
#define GEMM_LOAD_A(k0_, ra_, slot_)                                                          \
  {                                                                                           \
    const int k0__ = (k0_);                                                                   \
    const bool second__ = k0__ >= a.ksplit; /* block-uniform */                               \
    const float* ap__ = second__ ? a.p1 : a.p0;                          /* uniform */        \
    const int ka__ = second__ ? k0__ - a.ksplit : k0__;                                       \
    const bool kin__ = k0__ + kc * 4 < K;                                                     \
    ax.fetch(k0__, slot_);                                                                    \
    _Pragma("unroll") for (int i = 0; i < Cfg::A_F4; ++i)                                     \
      ra_[i] = kin__ ? *reinterpret_cast<const f32x4*>(ap__ + (aoff[i] + ka__)) : f32x4{0.f, 0.f, 0.f, 0.f}; \
  }
#define GEMM_LOAD_B(k0_)                                                                      \
  {                                                                                           \
    const int k0__ = (k0_);                                                                   \
    const bool kin__ = k0__ + kc * 4 < K;                                                     \
    _Pragma("unroll") for (int i = 0; i < Cfg::B_F4; ++i)                                     \
      rb[i] = kin__ ? *reinterpret_cast<const f32x4*>(Bp + (boff[i] + k0__)) : f32x4{0.f, 0.f, 0.f, 0.f}; \
  }
#define GEMM_STORE_TILE(buf_, ra_, slot_)                                                \
  {                                                                                           \
    char* sA__ = lds + (buf_) * Cfg::STAGE_BYTES;                                             \
    char* sB__ = sA__ + 2 * Cfg::PLANE_A;                                                     \
    ax.apply(ra_, slot_);                                                                     \
    _Pragma("unroll") for (int i = 0; i < Cfg::A_F4; ++i) {                                   \
      h16x4 hi__, lo__;                                                                       \
      split4(ra_[i], hi__, lo__);                                                             \
      *reinterpret_cast<h16x4*>(sA__ + a_lds_off[i]) = hi__;                                 \
      *reinterpret_cast<h16x4*>(sA__ + Cfg::PLANE_A + a_lds_off[i]) = lo__;                  \
    }                                                                                         \
    _Pragma("unroll") for (int i = 0; i < Cfg::B_F4; ++i) {                                   \
      h16x4 hi__, lo__;                                                                       \
      split4(rb[i], hi__, lo__);                                                             \
      *reinterpret_cast<h16x4*>(sB__ + b_lds_off[i]) = hi__;                                 \
      *reinterpret_cast<h16x4*>(sB__ + Cfg::PLANE_B + b_lds_off[i]) = lo__;                  \
    }                                                                                         \
  }
  // all MFMAs of one k-tile held in LDS stage buf_
#define GEMM_COMPUTE_TILE(buf_)                                                               \
  {                                                                                           \
    const char* sA__ = lds + (buf_) * Cfg::STAGE_BYTES;                                       \
    const char* sB__ = sA__ + 2 * Cfg::PLANE_A;                                               \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                  \
      h16x8 ah[TM], al[TM], bh[TN], bl[TN];                                                   \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                        \
        const int off = lds_chunk_off(a_r0 + i * 32, ks * 2 + g);                             \
        ah[i] = *reinterpret_cast<const h16x8*>(sA__ + off);                                  \
        al[i] = *reinterpret_cast<const h16x8*>(sA__ + Cfg::PLANE_A + off);                   \
      }                                                                                       \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                        \
        const int off = lds_chunk_off(b_r0 + j * 32, ks * 2 + g);                             \
        bh[j] = *reinterpret_cast<const h16x8*>(sB__ + off);                                  \
        bl[j] = *reinterpret_cast<const h16x8*>(sB__ + Cfg::PLANE_B + off);                   \
      }                                                                                       \
      /* the two cross terms first, the leading term last; TM*TN independent accumulators */  \
      /* between two MFMAs on the same accumulator */                                         \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                        \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                        \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                        \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
    }                                                                                         \
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment rows of this lane inside the A / B planes
  const int g = lane >> 5;
  const int a_r0 = wm * Cfg::WTM + (lane & 31);
  const int b_r0 = wn * Cfg::WTN + (lane & 31);

  const int nk = (K + BK - 1) / BK;
  GEMM_LOAD_A(0, ra0, 0);
  GEMM_LOAD_B(0);
  if (nk > 1) GEMM_LOAD_A(BK, ra1, 1);
  GEMM_STORE_TILE(0, ra0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // even k-tile: in LDS stage 0; A of tile kt+1 is landing in register set 1; set 0 is free
    if (kt + 2 < nk) GEMM_LOAD_A((kt + 2) * BK, ra0, 0);
    if (kt + 1 < nk) GEMM_LOAD_B((kt + 1) * BK);
    GEMM_COMPUTE_TILE(0);
    if (kt + 1 >= nk) break;
    GEMM_STORE_TILE(1, ra1, 1);           // stage 1 was last read in iteration kt-1 (barrier since)
    __syncthreads();
    // odd k-tile: in LDS stage 1; A of tile kt+2 is landing in register set 0; set 1 is free
    if (kt + 3 < nk) GEMM_LOAD_A((kt + 3) * BK, ra1, 1);
    if (kt + 2 < nk) GEMM_LOAD_B((kt + 2) * BK);
    GEMM_COMPUTE_TILE(1);
    if (kt + 2 < nk) {
      GEMM_STORE_TILE(0, ra0, 0);
      __syncthreads();
    }
  }
#undef GEMM_LOAD_A
#undef GEMM_LOAD_B
#undef GEMM_COMPUTE_TILE
#undef GEMM_STORE_TILE
}

template <typename Cfg>
__device__ __forceinline__ void gemm_mainloop(const ASrc& a, const float* __restrict__ Bp, int ldb,
                                              int M, int N, int K, int m0, int n0,
                                              float* lds_f, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  NoXform nx;
  gemm_mainloop<Cfg, NoXform>(a, Bp, ldb, M, N, K, m0, n0, lds_f, acc, nx);
}

// Coordinates of accumulator element `reg` of MFMA tile (i, j) for this lane.
template <typename Cfg>
__device__ __forceinline__ int acc_row(int m0, int i, int reg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return m0 + (wave / Cfg::WN) * Cfg::WTM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
template <typename Cfg>
__device__ __forceinline__ int acc_col(int n0, int j) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return n0 + (wave % Cfg::WN) * Cfg::WTN + j * 32 + (lane & 31);
}
