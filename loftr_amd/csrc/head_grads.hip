// d loss / d feat_c0, d loss / d feat_c1 from d loss / d sim_matrix: the two GEMMs behind the backward of both coarse heads.
//   reference: src/loftr/utils/coarse_matching.py:110-114 (dual-softmax: sim = einsum("nlc,nsc->nls", f0 / sqrt C, f1 / sqrt C) / T),
//              :122-123 (Sinkhorn: the same without T) -- what torch.autograd does for that einsum:
//                  g0[n] = alpha * dsim[n]   . feat_c1[n]        [L, S] x [S, C]
//                  g1[n] = alpha * dsim[n]^T . feat_c0[n]        [S, L] x [L, C]
// Until round 3 these were two torch.bmm calls (rocBLAS fp32 Tensile kernels, 2 x 810 us of the 3.5 ms dual-softmax backward).
//
// Both are fp32 x fp32 GEMMs with a LONG reduction (K = S or L = 4800) and a narrow output (C = 256 columns), and neither operand
// exists in the SP format: dsim is produced as fp32 by the heads' backward kernels (and, for the second product, is needed
// transposed), feat is the fp32 tensor autograd saved.  So this kernel splits on the fly: tiles travel global -> registers (fp32,
// one k-tile ahead of the compute) -> (hi, lo) fp16 -> LDS in the fragment layout of gemm.h, and the product is the same three fp16
// MFMAs per fp32 product with fp32 accumulation (hi*hi + hi*lo + lo*hi, csrc/gemm.h).  The transposes cost nothing: a thread
// loads a 4-column x 4- or 8-row micro tile with 16-byte loads ALONG the contiguous dimension and writes it to LDS across it.
//   workgroup: 4 waves, 128 output rows x all C <= 256 columns (a wave: 32 rows x C), k-tile 32; LDS 48 KB, two workgroups per CU.
//   Scaling (gemm.h: the lo half of |x| < 2^-3 is a subnormal fp16 number, and gradients are small): both operand tiles are staged
//   times a power of two that keeps the tile maximum in [2^10, 2^16) -- a RUNNING pair of exponents, changed only when a tile's
//   maximum leaves that window; when it changes the accumulators are rescaled by the exact ratio (a wave-uniform, rare branch), so
//   one accumulator set serves the whole reduction and the result is exact in the scales.
#include "gemm.h"
#include "head_grads.h"

namespace {
namespace hg {
constexpr int BM = 128, BK = 32, WAVES = 4, MAXT = 8;          // MAXT column tiles of 32 (C <= 256)
// Two workgroups per CU (4 waves, 48 KB of LDS, <= 256 registers each).  Until round 5 the kernel had to be held to ONE per CU (by
// LDS padding): two co-resident workgroups produced wrong partials -- a few of the 128 k-terms of a partial missing, different from
// run to run (found in round 4 at the first grids of more than 256 workgroups).  Root cause (round 5, tools/micro/hg_diag.py and the
// stand-alone reproducer tools/micro/pk_opsel_probe.hip, profiles/r05_pk_opsel_probe.txt): the staging conversions multiplied a PAIR
// of operand values by the running scale, which hipcc compiled to  v_pk_mul_f32 / v_pk_fma_f32 ... op_sel:[0,1]  (the LOW result takes
// the HIGH dword of the scale pair).  On gfx950 that form intermittently computes with a wrong operand in part of the wave when ANOTHER
// wave of the same SIMD has MFMAs in flight: 7.5 million wrong products in 5 launches of the probe with two workgroups per CU, none
// with one per CU, none without MFMAs, none for the op_sel_hi forms the rest of the library uses.  Loads, LDS stores, barriers and wait
// counts were all innocent (read-back diagnostics: registers and LDS correct, the packed product wrong).  The split below is therefore
// done with SCALAR conversions (hg_pack2); tests/test_isa_audit.py keeps the op_sel form out of every kernel of the library.
// -DHG_PACKED_SPLIT rebuilds the faulty form (A/B: tools/micro/hg_diag.py counts its bad partials).
constexpr int A_BYTES = BM * 128, B_BYTES = 256 * 128, LDS_BYTES = A_BYTES + B_BYTES + 64;     // + [2][WAVES] floats
struct __attribute__((packed, aligned(4))) F4U { f32x4 v; };

// (hi, lo) words of two consecutive values, scalar conversions only (see above); same values as sp_pack2 bit for bit
#ifndef HG_PACKED_SPLIT
__device__ __forceinline__ void hg_pack2(float a, float b, uint32_t& hi, uint32_t& lo) {
  asm volatile("" : "+v"(a));           // keep the two values apart: no SLP vectorisation into packed fp32 instructions
  asm volatile("" : "+v"(b));
  const uint32_t pa = sp_pack(a), pb = sp_pack(b);
  hi = (pa & 0xffffu) | (pb << 16);
  lo = (pa >> 16) | (pb & 0xffff0000u);
}
#else
#define hg_pack2 sp_pack2
#endif
struct Args {
  const float* a; long a_ld, a_bs;      // A source: TRANS = false: A[m][k] = a[n * a_bs + m * a_ld + k];  true: a[n * a_bs + k * a_ld + m]
  const float* b; long b_ld, b_bs;      // B[k][c] = b[n * b_bs + k * b_ld + c]
  float* out; long o_ld, o_bs;          // out[m][c] = alpha * sum_k A[m][k] B[k][c] at out[n * o_bs + m * o_ld + c]
  int M, K, C, NT;                      // NT = C / 32; K = reduction length of a batch element ...
  int Ktot;                             // ... except the last one when the batch is a split of ONE reduction of Ktot (0 = not a split)
  float alpha;
  int tiles_m, N;
  int Cr;                               // real columns (C = NT * 32 >= Cr, Cr % 4 == 0): loads and stores stop there
  // B rows through a convolution tap (gHo > 0; split-K batches only: reduction index t = n * K + k is output pixel (b, y, x) of a
  // [., gHo, gWo] map and B's row is input pixel (b, y gs + gdy, x gs + gdx) of the [., gH, gW] map b points at, or zero outside it)
  int gHo, gWo, gH, gW, gs, gdy, gdx;
  // ... all taps of a KH x KW filter in ONE launch (gtaps = KH KW > 1): batch element nb = tap * gns + n is chunk n of the reduction
  // seen through tap (ky, kx) = (tap / gKW, tap % gKW), i.e. gdy = ky - gpad, gdx = kx - gpad; its output is out[nb]
  int gtaps, gns, gKW, gpad;
};

__device__ __forceinline__ float lift_exp(float absmax, float& inv) {     // power of two that lifts absmax into [2^13, 2^14), exact inverse
  const int e = (int)((__float_as_uint(absmax) >> 23) & 0xffu) - 126;
  int sh = 14 - e;
  sh = (absmax > 1e-30f && absmax < 1e30f) ? sh : 0;
  sh = sh > 60 ? 60 : (sh < -60 ? -60 : sh);
  inv = __uint_as_float((unsigned)(127 - sh) << 23);
  return __uint_as_float((unsigned)(127 + sh) << 23);
}

template <bool TRANS>
__global__ __launch_bounds__(WAVES * 64, 2) void head_grad_kernel(Args p) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  char* const sA = lds;
  char* const sB = lds + A_BYTES;
  float* const smax = reinterpret_cast<float*>(lds + A_BYTES + B_BYTES);     // [2][WAVES] per-wave |A|, |B| maxima of the tile being staged
  int tm, tn_;
  if (!xcd_tile(p.tiles_m * p.N, 1, tm, tn_)) return;
  (void)tn_;
  const int nb = tm / p.tiles_m, m0 = (tm - nb * p.tiles_m) * BM;
  int n = nb, gdy = p.gdy, gdx = p.gdx;
  if (p.gtaps > 1) {
    const int tap = nb / p.gns;
    n = nb - tap * p.gns;
    gdy = tap / p.gKW - p.gpad; gdx = tap % p.gKW - p.gpad;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, li = lane & 31;
  const float* A = p.a + (long)n * p.a_bs;
  const float* B = p.b + (long)n * p.b_bs;
  const int M = p.M, C = p.C, NT = p.NT;
  const int K = p.Ktot ? min(p.K, p.Ktot - n * p.K) : p.K;          // split-K batches: the last chunk is shorter
  const int nk = ceil_div(K, BK);

  // ---- staging registers: A 16 floats, B 32 floats per thread and k-tile
  //   A, plain:      thread = (row tid >> 1, k half tid & 1): 16 consecutive k of one row
  //   A, transposed: thread = (4 rows m = 4 (tid & 31) .., 4 k = 4 (tid >> 5) ..): 4 loads of 4 consecutive m
  //   B:             thread = (4 columns c = 4 (tid & 63) .., 8 k = 8 (tid >> 6) ..): 8 loads of 4 consecutive c
  f32x4 ra[4], rb[8];
  const bool a_vec = TRANS ? (m0 + BM <= M) : true;
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    if (!TRANS) {
      const int row = min(m0 + (tid >> 1), M - 1), kb = k0 + (tid & 1) * 16;
      const float* src = A + (long)row * p.a_ld + kb;
      if (kb + 16 <= K) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ra[q] = reinterpret_cast<const F4U*>(src + 4 * q)->v;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[q][e] = kb + 4 * q + e < K ? src[4 * q + e] : 0.f;
      }
    } else {
      const int mb = m0 + 4 * (tid & 31), kb = k0 + 4 * (tid >> 5);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = kb + q;
        const float* src = A + (long)min(k, K - 1) * p.a_ld + mb;
        if (a_vec) ra[q] = reinterpret_cast<const F4U*>(src)->v;
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[q][e] = src[min(mb + e, M - 1) - mb];
        }
        if (k >= K) ra[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const int cb = 4 * (tid & 63), kb2 = k0 + 8 * (tid >> 6);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = kb2 + q;
      const float* brow = nullptr;
      if (k < K) {
        if (p.gHo > 0) {
          const long t = (long)n * p.K + k;
          const int hw = p.gHo * p.gWo, bb = (int)(t / hw), rem = (int)(t - (long)bb * hw);
          const int y = rem / p.gWo, x = rem - y * p.gWo;
          const int iy = y * p.gs + gdy, ix = x * p.gs + gdx;
          if ((unsigned)iy < (unsigned)p.gH && (unsigned)ix < (unsigned)p.gW) brow = p.b + ((long)(bb * p.gH + iy) * p.gW + ix) * p.b_ld;
        } else {
          brow = B + (long)k * p.b_ld;
        }
      }
      rb[q] = (cb < p.Cr && brow) ? *reinterpret_cast<const f32x4*>(brow + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // registers -> (hi, lo) halves -> LDS rows of 128 B (chunks 0..3 hi, 4..7 lo; 16-B chunk c of row r at slot c ^ ((r >> 1) & 7))
  auto store_tile = [&](float a_sc, float b_sc) {
    if (!TRANS) {
      const int r = tid >> 1, c0 = (tid & 1) * 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float v[8] = {ra[2 * h][0] * a_sc, ra[2 * h][1] * a_sc, ra[2 * h][2] * a_sc, ra[2 * h][3] * a_sc,
                            ra[2 * h + 1][0] * a_sc, ra[2 * h + 1][1] * a_sc, ra[2 * h + 1][2] * a_sc, ra[2 * h + 1][3] * a_sc};
        u32x4 hi, lo;
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hg_pack2(v[2 * e], v[2 * e + 1], hw[e], lw[e]);
        hi = u32x4{hw[0], hw[1], hw[2], hw[3]}; lo = u32x4{lw[0], lw[1], lw[2], lw[3]};
        *reinterpret_cast<u32x4*>(sA + lds_chunk_off(r, c0 + h)) = hi;
        *reinterpret_cast<u32x4*>(sA + lds_chunk_off(r, 4 + c0 + h)) = lo;
      }
    } else {
      const int rb0 = 4 * (tid & 31), kq = tid >> 5;             // rows rb0 .. +3, k = 4 kq .. +3: half of chunk kq >> 1
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t h0, l0, h1, l1;
        hg_pack2(ra[0][e] * a_sc, ra[1][e] * a_sc, h0, l0);
        hg_pack2(ra[2][e] * a_sc, ra[3][e] * a_sc, h1, l1);
        const int r = rb0 + e;
        *reinterpret_cast<uint2*>(sA + lds_chunk_off(r, kq >> 1) + (kq & 1) * 8) = uint2{h0, h1};
        *reinterpret_cast<uint2*>(sA + lds_chunk_off(r, 4 + (kq >> 1)) + (kq & 1) * 8) = uint2{l0, l1};
      }
    }
    const int cb = 4 * (tid & 63), kc = tid >> 6;                // columns cb .. +3 (rows of the B tile), k = 8 kc .. +7: chunk kc
    if (cb < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) hg_pack2(rb[2 * q][e] * b_sc, rb[2 * q + 1][e] * b_sc, hw[q], lw[q]);
        const int r = cb + e;
        *reinterpret_cast<u32x4*>(sB + lds_chunk_off(r, kc)) = u32x4{hw[0], hw[1], hw[2], hw[3]};
        *reinterpret_cast<u32x4*>(sB + lds_chunk_off(r, 4 + kc)) = u32x4{lw[0], lw[1], lw[2], lw[3]};
      }
    }
  };
  auto tile_absmax = [&](float& mb) {
    float m = 0.f;
    mb = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(ra[q][e]));
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) mb = fmaxf(mb, fabsf(rb[q][e]));
    mb = half_max(mb); mb = fmaxf(mb, swap32(mb));
    m = half_max(m);
    return fmaxf(m, swap32(m));
  };

  f32x16 acc[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int arow = wave * 32 + li;
  float a_sc = 1.f, a_inv = 1.f, b_sc = 1.f, b_inv = 1.f;      // running operand scales (powers of two) and their inverses
  load_tile(0);
  { float wmb; const float wma = tile_absmax(wmb); if (lane == 0) { smax[wave] = wma; smax[WAVES + wave] = wmb; } }
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                                     // every wave is done reading the previous tile; the staged tile's maxima are visible
    const float ma = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    const float mb = fmaxf(fmaxf(smax[4], smax[5]), fmaxf(smax[6], smax[7]));
    // keep a running scale while it holds the tile maximum in [2^10, 2^16); otherwise re-lift to [2^13, 2^14) and carry the exact
    // ratio into the accumulators (block-uniform branch)
    const bool a_bad = !(ma * a_sc >= 1024.f && ma * a_sc < 65000.f) && ma > 1e-30f;
    const bool b_bad = !(mb * b_sc >= 1024.f && mb * b_sc < 65000.f) && mb > 1e-30f;
    if (a_bad || b_bad) {
      float na_sc = a_sc, na_inv = a_inv, nb_sc = b_sc, nb_inv = b_inv;
      if (a_bad) na_sc = lift_exp(ma, na_inv);
      if (b_bad) nb_sc = lift_exp(mb, nb_inv);
      const float ratio = (na_sc * a_inv) * (nb_sc * b_inv);     // new / old, a power of two
      if (kt > 0) {
#pragma unroll
        for (int j = 0; j < MAXT; ++j)
          if (j < NT) acc[j] *= ratio;
      }
      a_sc = na_sc; a_inv = na_inv; b_sc = nb_sc; b_inv = nb_inv;
    }
    store_tile(a_sc, b_sc);
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);                  // in flight under the MFMAs below
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const h16x8 ah = *reinterpret_cast<const h16x8*>(sA + lds_chunk_off(arow, 2 * ks + g));
      const h16x8 al = *reinterpret_cast<const h16x8*>(sA + lds_chunk_off(arow, 4 + 2 * ks + g));
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        if (j < NT) {
          const int br = j * 32 + li;
          const h16x8 bh = *reinterpret_cast<const h16x8*>(sB + lds_chunk_off(br, 2 * ks + g));
          const h16x8 bl = *reinterpret_cast<const h16x8*>(sB + lds_chunk_off(br, 4 + 2 * ks + g));
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < nk) {                                   // maxima of the tile just loaded (every wave read the previous ones before the barrier above)
      float wmb;
      const float wma = tile_absmax(wmb);
      if (lane == 0) { smax[wave] = wma; smax[WAVES + wave] = wmb; }
    }
  }
  const float undo = a_inv * b_inv * p.alpha;
  // ---- out[m][c] = alpha * acc: lane = column, 16 rows per register set
  float* O = p.out + (long)nb * p.o_bs;
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    if (j >= NT) continue;
    const int c = j * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (m < M && c < p.Cr) O[(long)m * p.o_ld + c] = acc[j][r] * undo;
    }
  }
}
}  // namespace hg
}  // namespace

// Internal launcher (head_grads.h): out[n] = alpha * op(A[n]) . B[n] for nbatch batch elements; trans: A is read as [k][m].  ktot > 0:
// the batch is a split of ONE reduction of length ktot into chunks of K (split-K: the caller sums the nbatch partial outputs).
int launch_head_grad(const float* a, long a_ld, long a_bs, bool trans, const float* b, long b_ld, long b_bs, float* out, long o_ld,
                     long o_bs, int M, int K, int ktot, int C, int nbatch, float alpha, hipStream_t st) {
  if (C % 32 != 0 || C > 32 * hg::MAXT || (b_ld & 3) || M <= 0 || K <= 0 || nbatch <= 0) return LOFTR_ERR_UNSUPPORTED;
  hg::Args g{};
  g.a = a; g.a_ld = a_ld; g.a_bs = a_bs; g.b = b; g.b_ld = b_ld; g.b_bs = b_bs; g.out = out; g.o_ld = o_ld; g.o_bs = o_bs;
  g.M = M; g.K = K; g.Ktot = ktot; g.C = C; g.NT = C / 32; g.alpha = alpha; g.N = nbatch; g.tiles_m = ceil_div(M, hg::BM);
  g.Cr = C;
  if (trans) hipLaunchKernelGGL((hg::head_grad_kernel<true>), dim3(xcd_grid(g.tiles_m * nbatch, 1)), dim3(hg::WAVES * 64), 0, st, g);
  else hipLaunchKernelGGL((hg::head_grad_kernel<false>), dim3(xcd_grid(g.tiles_m * nbatch, 1)), dim3(hg::WAVES * 64), 0, st, g);
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}

// ---- shared by the layer backward passes (encoder_bwd.hip, fine_bwd.hip) -----------------------------------------------------------
namespace {
// out[i] = sum_p part[p * stride + i]   (ascending p: deterministic)
__global__ void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int P, long stride, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += part[(long)p * stride + i];
  out[i] = s;
}
// the same for P >> n: 32 columns x 8 lanes per block, lane l adds p = l, l + 8, ... (ascending), then the 8 lane sums in order
__global__ __launch_bounds__(256) void reduce_partials_tall_kernel(const float* __restrict__ part, float* __restrict__ out, int P, long stride, long n) {
  __shared__ float red[8][32];
  const int c = threadIdx.x & 31, l = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + c;
  float s = 0.f;
  if (i < n) for (int p = l; p < P; p += 8) s += part[(long)p * stride + i];
  red[l][c] = s;
  __syncthreads();
  if (l == 0 && i < n) {
    float t = red[0][c];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][c];
    out[i] = t;
  }
}
// part[block][c] = sum over the block's WG_ROWS rows of x[row][c]   (column sums, stage 1; ascending rows)
constexpr int COLSUM_ROWS = 256;
__global__ void colsum_part_kernel(const float* __restrict__ x, long rows, int C, float* __restrict__ part) {
  const long r0 = (long)blockIdx.x * COLSUM_ROWS;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (long r = r0; r < r0 + COLSUM_ROWS && r < rows; ++r) s += x[r * C + c];
    part[(long)blockIdx.x * C + c] = s;
  }
}
}  // namespace
// Tokens per split-K partial: enough partials to put ~256 workgroups on the chip (one 128-row tile of dW per workgroup and partial),
// between 128 and 512 tokens, a multiple of the k-tile.  (A fixed 512 left a 256 x 256 gradient over 9600 tokens on 38 workgroups.)
static int wgrad_chunk(long T, int O) {
  const int forced = loftr_debug_value(LOFTR_DBG_WGRAD_CHUNK);      // (debug / A-B)
  if (forced > 0) return forced;
  const int tiles = ceil_div(O, 128);
  const int target = tiles >= 256 ? 1 : 256 / tiles;
  long c = (T + target - 1) / target;
  c = (c + 31) / 32 * 32;
  return (int)(c < 128 ? 128 : (c > 512 ? 512 : c));
}
// scratch of launch_wgrad for any O' <= O (callers size one buffer for their largest matrix and reuse it)
size_t wgrad_part_floats(long T, int O, int I) {
  size_t best = 0;
  for (int o = 128;; o += 128) {
    const int oo = o < O ? o : O;
    const size_t v = (size_t)ceil_div((int)T, wgrad_chunk(T, oo)) * oo * I;
    best = v > best ? v : best;
    if (o >= O) break;
  }
  return best;
}
// dW [O][I] = dy^T act  (dy [T, O], act [T, I]): a split-K batch of launch_head_grad + an ordered sum of the partials; I in column
// blocks of <= 256.  part: part_cap floats of scratch, >= wgrad_part_floats(T, O, I) (the number of partials is NOT monotone in T: the
// chunk is rounded to the k-tile and clamped, so a buffer sized for a longer T may be too small -- checked, not assumed).
int launch_wgrad(const float* dy, int O, const float* act, int I, long T, float* dW, float* part, size_t part_cap, hipStream_t st) {
  const int kch = wgrad_chunk(T, O);
  const int ns = ceil_div((int)T, kch);
  if ((size_t)ns * O * I > part_cap) return LOFTR_ERR_WORKSPACE;
  for (int c0 = 0; c0 < I; c0 += 256) {
    const int cw = I - c0 < 256 ? I - c0 : 256;
    const int rc = launch_head_grad(dy, O, (long)kch * O, true, act + c0, I, (long)kch * I, part + c0, I, (long)O * I, O,
                                    kch, (int)T, cw, ns, 1.f, st);
    if (rc) return rc;
  }
  return launch_reduce_partials(part, dW, ns, (long)O * I, (long)O * I, st);
}
int launch_reduce_partials(const float* part, float* out, int P, long stride, long n, hipStream_t st) {
  const bool no_tall = loftr_debug_value(LOFTR_DBG_REDUCE_TALL) == 0;    // (debug / A-B)
  if (P >= 16 && !no_tall)       // many partials (split-K weight gradients, LayerNorm weight gradients: one partial per 64 rows): 8 lanes share the walk over P
    hipLaunchKernelGGL(reduce_partials_tall_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, part, out, P, stride, n);
  else
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, out, P, stride, n);
  return LOFTR_OK;
}
size_t colsum_part_floats(long rows, int C) { return (size_t)((rows + COLSUM_ROWS - 1) / COLSUM_ROWS) * C; }
// out[c] = sum_rows x[row][c]  (fixed order)
int launch_colsum(const float* x, long rows, int C, float* out, float* part, hipStream_t st) {
  const int nb = (int)((rows + COLSUM_ROWS - 1) / COLSUM_ROWS);
  hipLaunchKernelGGL(colsum_part_kernel, dim3(nb), dim3(256), 0, st, x, rows, C, part);
  return launch_reduce_partials(part, out, nb, C, C, st);
}

// g0 [N,L,C] = alpha * dsim . feat_c1,  g1 [N,S,C] = alpha * dsim^T . feat_c0 (either may be null).  dsim: [N] matrices of L x S with row
// pitch dsim_ld and batch stride dsim_bs (floats): the Sinkhorn head hands in the interior of its [L+1, S+1] gradient.
extern "C" int loftr_head_feat_grads(const float* dsim, long dsim_ld, long dsim_bs, const float* feat_c0, const float* feat_c1,
                                     int N, int L, int S, int C, float alpha, float* g0, float* g1, void* stream) {
  LOFTR_CHECK_ARG(dsim && feat_c0 && feat_c1 && (g0 || g1) && N >= 0 && L > 0 && S > 0 && C > 0 && dsim_ld >= S);
  if (C % 32 != 0 || C > 32 * hg::MAXT) return LOFTR_ERR_UNSUPPORTED;
  if (N == 0) return LOFTR_OK;
  hipStream_t st = (hipStream_t)stream;
  int rc = LOFTR_OK;
  if (g0) rc = launch_head_grad(dsim, dsim_ld, dsim_bs, false, feat_c1, C, (long)S * C, g0, C, (long)L * C, L, S, 0, C, N, alpha, st);
  if (rc == LOFTR_OK && g1)
    rc = launch_head_grad(dsim, dsim_ld, dsim_bs, true, feat_c0, C, (long)L * C, g1, C, (long)S * C, S, L, 0, C, N, alpha, st);
  return rc;
}


// ---- weight gradient of a convolution (backbone training) ---------------------------------------------------------------------------
// dW[tap][co][ci] = sum over output pixels (b, y, x) of dy[b, y, x, co] * x[b, y stride + ky - pad, x stride + kx - pad, ci]: per tap the
// split-K product of launch_wgrad with B's rows gathered through the tap (hg::Args::gHo ..).   what torch.autograd does for
// F.conv2d's weight (resnet_fpn.py: every nn.Conv2d of the backbone, bias-free)
// Tokens per partial for the convolution: all KH KW taps go in one launch, so the reduction is cut only as far as it takes to put
// ~512 workgroups (two rounds of one per CU) on the chip -- at least 128 pixels per partial, a multiple of the k-tile.
static int conv_wgrad_chunk(long T, int Cout, int taps) {
  const int forced = loftr_debug_value(LOFTR_DBG_WGRAD_CHUNK);      // (debug: tools/micro/conv_wgrad_debug.py)
  if (forced > 0) return forced;
  const int per = ceil_div(Cout, hg::BM) * taps;
  const int target = per >= 512 ? 1 : 512 / per;
  long c = (T + target - 1) / target;
  c = (c + 31) / 32 * 32;
  return (int)(c < 128 ? 128 : c);
}
extern "C" size_t loftr_conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return 0;
  const long T = (long)B * Ho * Wo;
  return (size_t)ceil_div((int)T, conv_wgrad_chunk(T, Cout, KH * KW)) * KH * KW * Cout * Cin * sizeof(float) + 1024;
}
extern "C" int loftr_conv_wgrad(const float* dy, const float* x, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                float* dw_taps, void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(dy && x && dw_taps && B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
  if (Cin % 4 != 0 || Cin > 32 * hg::MAXT) return LOFTR_ERR_UNSUPPORTED;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return LOFTR_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int taps = KH * KW;
  if (B == 0) { (void)hipMemsetAsync(dw_taps, 0, sizeof(float) * taps * Cout * Cin, st); return LOFTR_OK; }
  const long T = (long)B * Ho * Wo;
  if (T >= (1L << 31)) return LOFTR_ERR_UNSUPPORTED;
  LOFTR_CHECK_ARG(ws != nullptr);
  if (ws_bytes < loftr_conv_wgrad_workspace_bytes(B, Ho, Wo, Cin, Cout, KH, KW)) return LOFTR_ERR_WORKSPACE;
  float* part = reinterpret_cast<float*>(ws);
  const int kch = conv_wgrad_chunk(T, Cout, taps), ns = ceil_div((int)T, kch);
  hg::Args g{};
  g.a = dy; g.a_ld = Cout; g.a_bs = (long)kch * Cout;
  g.b = x; g.b_ld = Cin; g.b_bs = 0;
  g.out = part; g.o_ld = Cin; g.o_bs = (long)Cout * Cin;
  g.M = Cout; g.K = kch; g.Ktot = (int)T; g.C = ceil_div(Cin, 32) * 32; g.NT = g.C / 32; g.Cr = Cin; g.alpha = 1.f; g.N = taps * ns;
  g.tiles_m = ceil_div(Cout, hg::BM);
  g.gHo = Ho; g.gWo = Wo; g.gH = H; g.gW = W; g.gs = stride; g.gdy = -pad; g.gdx = -pad;
  g.gtaps = taps; g.gns = ns; g.gKW = KW; g.gpad = pad;
  hipLaunchKernelGGL((hg::head_grad_kernel<true>), dim3(xcd_grid(g.tiles_m * g.N, 1)), dim3(hg::WAVES * 64), 0, st, g);
  for (int t = 0; t < taps; ++t) {
    const int rc = launch_reduce_partials(part + (size_t)t * ns * Cout * Cin, dw_taps + (size_t)t * Cout * Cin, ns, (long)Cout * Cin, (long)Cout * Cin, st);
    if (rc) return rc;
  }
  LOFTR_CHECK_LAUNCH();
  return LOFTR_OK;
}
