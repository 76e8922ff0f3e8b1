// Backward of the dual-softmax confidence (coarse_matching.py:110-119; what torch.autograd derives for
//   conf = softmax(sim, 1) * softmax(sim, 2),   sim = <feat_c0, feat_c1> / (C temperature), masked_fill_(-1e9) on padding).
// With A = softmax over rows i (per column), B = softmax over columns j (per row), conf = A B and G = dL/dconf:
//   dL/dsim_ij = 2 G_ij conf_ij - B_ij r_i - A_ij c_j,    r_i = sum_j G_ij conf_ij,   c_j = sum_i G_ij conf_ij,
// and 0 on the mask-filled entries (masked_fill_ cuts the graph there).  The forward's own machinery re-creates what is
// needed -- descriptor staging, the statistics sweep (row / column (max, 1 / sum)) and the store-only sweep that writes
// sim -- so nothing but feat_c0 / feat_c1 has to stay alive between forward and backward; three streaming passes follow
// (row dot products, column dot products, the elementwise combination in place).  Included by coarse_match.hip.
namespace dsb {
constexpr int RCH = 32;          // row chunks of the column pass

__device__ __forceinline__ float soft(float s, float2 st) { return fexp(s - st.x) * st.y; }      // st = (max, 1 / sum)

// r[n, i] = sum_j G_ij conf_ij.   one wave per row; grid (ceil(N L / 4)), 256 threads
__global__ __launch_bounds__(256) void row_dot_kernel(const float* __restrict__ sim, const float* __restrict__ G, Geometry g,
                                                      const float2* __restrict__ rowstat, const float2* __restrict__ colstat,
                                                      float* __restrict__ r) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)g.N * g.L) return;
  const int n = (int)(row / g.L);
  const float2 rs = rowstat[row];
  const float2* cs = colstat + (long)n * g.S;
  const float* sr = sim + row * g.S;
  const float* gr = G + row * g.S;
  float acc = 0.f;
  for (int j = lane; j < g.S; j += 64) {
    const float gg = gr[j];
    if (gg != 0.f) { const float s = sr[j]; acc += gg * (soft(s, rs) * soft(s, cs[j])); }
  }
  acc = wave_sum(acc);
  if (lane == 0) r[row] = acc;
}

// part[n, k, j] = sum over the rows of chunk k of G_ij conf_ij.   grid (ceil(S / 256), RCH, N), 256 threads
__global__ __launch_bounds__(256) void col_dot_part_kernel(const float* __restrict__ sim, const float* __restrict__ G, Geometry g,
                                                           const float2* __restrict__ rowstat, const float2* __restrict__ colstat,
                                                           float* __restrict__ part) {
  const int n = blockIdx.z, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= g.S) return;
  const int per = ceil_div(g.L, RCH), i0 = blockIdx.y * per, i1 = min(i0 + per, g.L);
  const float2 cs = colstat[(long)n * g.S + j];
  float acc = 0.f;
  for (int i = i0; i < i1; ++i) {
    const long o = ((long)n * g.L + i) * g.S + j;
    const float gg = G[o];
    if (gg != 0.f) { const float s = sim[o]; acc += gg * (soft(s, rowstat[(long)n * g.L + i]) * soft(s, cs)); }
  }
  part[((long)n * RCH + blockIdx.y) * g.S + j] = acc;
}
__global__ void col_dot_merge_kernel(const float* __restrict__ part, Geometry g, float* __restrict__ c) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)g.N * g.S) return;
  const int n = (int)(idx / g.S), j = (int)(idx - (long)n * g.S);
  float acc = 0.f;
  for (int k = 0; k < RCH; ++k) acc += part[((long)n * RCH + k) * g.S + j];       // fixed order: deterministic
  c[idx] = acc;
}

// sim -> dL/dsim in place.   grid (ceil(N L / 4)), 256 threads, one wave per row
__global__ LOFTR_NO_PACKED_FP32 __launch_bounds__(256) void dsim_kernel(float* __restrict__ sim, const float* __restrict__ G, Geometry g,
                                                   const float2* __restrict__ rowstat, const float2* __restrict__ colstat,
                                                   const float* __restrict__ r, const float* __restrict__ c,
                                                   const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)g.N * g.L) return;
  const int n = (int)(row / g.L);
  const float2 rs = rowstat[row];
  const float ri = r[row];
  const bool row_ok = !mask0 || mask0[row];
  float* sr = sim + row * g.S;
  const float* gr = G + row * g.S;
  for (int j = lane; j < g.S; j += 64) {
    const long cj = (long)n * g.S + j;
    float d = 0.f;
    if (row_ok && (!mask1 || mask1[cj])) {
      const float s = sr[j];
      const float B = soft(s, rs), A = soft(s, colstat[cj]);
      d = 2.f * gr[j] * (A * B) - B * ri - A * c[cj];
    }
    sr[j] = d;
  }
}
}  // namespace dsb
