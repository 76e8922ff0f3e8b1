// fp32-exact "NT" GEMM main loop on the CDNA4 matrix cores:  acc[m][n] += sum_k A[m][k] * B[n][k]
//
// Why fp32 MFMA: the dual-softmax confidences must match the reference within 1e-4 and the
// sub-pixel key points within 1e-3 px; bf16/fp16/tf32-rounded operands miss that by 10-100x
// (SURVEY.md §0).  v_mfma_f32_32x32x2_f32 is bitwise an fp32 fma chain and runs at the fp32
// vector peak (157 TFLOP/s) while leaving the VALU free for the fused epilogues.
//
// Tiling (per workgroup of WM*WN waves of 64 lanes):
//   * block tile BM x BN, k-step BK; each wave owns a (BM/WM) x (BN/WN) sub-tile made of
//     TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each).
//   * A and B tiles are staged global -> registers -> LDS (double buffered; the global loads
//     of tile t+1 are issued before the MFMAs of tile t and written to LDS after them).
//   * LDS rows are k-contiguous with a 4-float pad (row stride BK+4 floats): the 16-byte
//     fragment reads below then hit 16 different 16-B slots per 16-lane group (conflict free
//     for ds_read_b128, bank = (addr/4) % 64), and the staging ds_write_b128 of 8 consecutive
//     lanes covers one full row (conflict free, bank = (addr/4) % 32).
//   * MFMA operand mapping: v_mfma_f32_32x32x2_f32 wants A[i = lane&31][k = lane>>5].  The
//     order of k inside the contraction is free as long as A and B agree, so lane (i, h)
//     fetches ONE float4 = k in {8*kk + 4*h .. +3} and feeds component j to the j-th MFMA:
//     MFMA j contracts k = 8*kk + j (h=0) and 8*kk + 4 + j (h=1).  One ds_read_b128 per
//     operand per 4 MFMAs.
//   * C/D layout (dtype independent on gfx950): col = lane & 31,
//     row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), reg in [0,16).
#pragma once
#include "common.h"

// Where the rows of the A operand come from.
//   * plain:            row r, k  ->  p0[r * ld0 + k]
//   * concatenated K:   k >= ksplit reads p1[r * ld1 + (k - ksplit)]   (cat([x, msg], dim=2) of
//                       transformer.py:55 without materialising the concatenation)
//   * gathered rows:    r -> gather[r]  (fine_preprocess.py:51-52 picks coarse features at
//                       (b_ids, i_ids) -- the index already folds b*L + i)
struct ASrc {
  const float* p0; int ld0;
  const float* p1; int ld1; int ksplit;
  const int64_t* gather;
};
__host__ __device__ static inline ASrc asrc_plain(const float* p, int ld) { return ASrc{p, ld, nullptr, 0, 1 << 30, nullptr}; }
__host__ __device__ static inline ASrc asrc_cat(const float* p0, int ld0, const float* p1, int ld1, int ksplit) {
  return ASrc{p0, ld0, p1, ld1, ksplit, nullptr};
}
__host__ __device__ static inline ASrc asrc_gather(const float* p, int ld, const int64_t* idx) {
  return ASrc{p, ld, nullptr, 0, 1 << 30, idx};
}

template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
  static constexpr int THREADS = WM * WN * 64;
  static constexpr int LDS_STRIDE = BK + 4;                    // floats
  static constexpr int WTM = BM / WM, WTN = BN / WN;           // wave sub-tile
  static constexpr int TM = WTM / 32, TN = WTN / 32;           // 32x32 MFMA tiles per wave
  static constexpr int KCH = BK / 4;                           // float4 chunks per row
  static constexpr int A_F4 = BM * KCH / THREADS;              // float4 per thread per tile
  static constexpr int B_F4 = BN * KCH / THREADS;
  static constexpr int LDS_FLOATS = 2 * (BM + BN) * LDS_STRIDE;
  static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be a multiple of 32");
  static_assert((BM * KCH) % THREADS == 0 && (BN * KCH) % THREADS == 0, "loader mapping");
  static_assert(BK % 8 == 0, "BK must be a multiple of 8");
};

// Runs the whole K loop for the block tile at (m0, n0).  M, N are the valid extents (rows
// beyond them are clamped on load -- the caller masks them in its epilogue).  K % BK == 0.
template <typename Cfg>
__device__ __forceinline__ void gemm_mainloop(const ASrc& a, const float* __restrict__ Bp, int ldb,
                                              int M, int N, int K, int m0, int n0,
                                              float* lds, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, LS = Cfg::LDS_STRIDE;
  constexpr int TM = Cfg::TM, TN = Cfg::TN, KCH = Cfg::KCH;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;

  constexpr int STAGE = (BM + BN) * LS;       // floats per stage buffer: [A tile | B tile]

  // ---- loader: per-thread global row pointers (k-independent part) ------------------------
  const float* arow0[Cfg::A_F4];
  const float* arow1[Cfg::A_F4];
  const float* brow[Cfg::B_F4];
  int a_lds_off[Cfg::A_F4], b_lds_off[Cfg::B_F4];
#pragma unroll
  for (int i = 0; i < Cfg::A_F4; ++i) {
    int idx = tid + i * Cfg::THREADS;
    int r = idx / KCH, kc = idx % KCH;
    int gr = min(m0 + r, M - 1);
    long row = a.gather ? (long)a.gather[gr] : (long)gr;
    arow0[i] = a.p0 + row * a.ld0 + kc * 4;
    arow1[i] = a.p1 ? a.p1 + row * a.ld1 + kc * 4 : nullptr;
    a_lds_off[i] = r * LS + kc * 4;
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_F4; ++i) {
    int idx = tid + i * Cfg::THREADS;
    int r = idx / KCH, kc = idx % KCH;
    int gr = min(n0 + r, N - 1);
    brow[i] = Bp + (long)gr * ldb + kc * 4;
    b_lds_off[i] = r * LS + kc * 4;
  }

  f32x4 ra[Cfg::A_F4], rb[Cfg::B_F4];
  // (macros, not lambdas: capturing the staging registers by reference makes hipcc keep a
  //  scratch copy of them)
#define GEMM_LOAD_TILE(k0_)                                                                   \
  {                                                                                           \
    const int k0__ = (k0_);                                                                   \
    const bool second__ = k0__ >= a.ksplit; /* wave-uniform */                                \
    const int ka__ = second__ ? k0__ - a.ksplit : k0__;                                       \
    _Pragma("unroll") for (int i = 0; i < Cfg::A_F4; ++i)                                     \
      ra[i] = *reinterpret_cast<const f32x4*>((second__ ? arow1[i] : arow0[i]) + ka__);      \
    _Pragma("unroll") for (int i = 0; i < Cfg::B_F4; ++i)                                     \
      rb[i] = *reinterpret_cast<const f32x4*>(brow[i] + k0__);                               \
  }
#define GEMM_STORE_TILE(buf_)                                                                 \
  {                                                                                           \
    float* sA__ = lds + (buf_) * STAGE;                                                       \
    float* sB__ = sA__ + BM * LS;                                                             \
    _Pragma("unroll") for (int i = 0; i < Cfg::A_F4; ++i)                                     \
      *reinterpret_cast<f32x4*>(sA__ + a_lds_off[i]) = ra[i];                                \
    _Pragma("unroll") for (int i = 0; i < Cfg::B_F4; ++i)                                     \
      *reinterpret_cast<f32x4*>(sB__ + b_lds_off[i]) = rb[i];                                \
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (floats) inside a stage buffer
  const int frag_k = (lane >> 5) * 4;
  const int a_frag = (wm * Cfg::WTM + (lane & 31)) * LS + frag_k;
  const int b_frag = (wn * Cfg::WTN + (lane & 31)) * LS + frag_k;

  const int nk = K / BK;
  GEMM_LOAD_TILE(0);
  GEMM_STORE_TILE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) GEMM_LOAD_TILE((kt + 1) * BK);   // in flight during the MFMAs below
    const float* sa = lds + cur * STAGE + a_frag;
    const float* sb = lds + cur * STAGE + BM * LS + b_frag;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(sa + i * 32 * LS + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(sb + j * 32 * LS + kk * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) {
      GEMM_STORE_TILE(cur ^ 1);     // buffer cur^1 was last read in iteration kt-1 (barrier since)
      __syncthreads();
    }
  }
#undef GEMM_LOAD_TILE
#undef GEMM_STORE_TILE
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
