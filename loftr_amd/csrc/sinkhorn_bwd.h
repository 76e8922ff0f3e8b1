// Backward of the Sinkhorn head (coarse_matching.py:121-143 + SuperGlue's log_optimal_transport): what torch.autograd derives for
//   assign = exp(Zp + u_T + v_T - norm),   u_t = log_mu - LSE_j(Zp + v_{t-1}),   v_t = log_nu - LSE_i(Zp + u_t),   v_0 = 0,
// Zp = sim padded with the dustbin row / column alpha = bin_score, by reverse mode through the T unrolled iterations.  With
// G = dL/dassign and D = G assign:   dZ = D,  du_T = rowsum D,  dv_T = colsum D,  then for t = T .. 1
//   v-step:  Pc = exp(Zp + u_t + v_t - log_nu)   (the column softmax of Zp + u_t):  dZ -= dv_j Pc,   du_t,i -= sum_j dv_j Pc
//   u-step:  Pr = exp(Zp + v_{t-1} + u_t - log_mu) (the row softmax of Zp + v_{t-1}): dZ -= du_i Pr,  dv_{t-1},j = -sum_i du_i Pr
// (du_{t-1} starts at 0: u_{t-1} is read by v_{t-1} only).  No extra reductions: the log-sum-exps ARE log_mu - u_t and log_nu - v_t,
// which the forward's own iteration kernels re-create (u_t, v_t saved after every iteration: 2 T + 1 small vectors).  One streaming
// pass over (Z, dZ) per step: the row steps as a wave per padded row, the column steps as a thread per padded column over 32 row
// chunks merged in a fixed order.  dZ is the PADDED [N, L+1, S+1] volume; d bin_score is the sum over its dustbin row and column.
// Included by coarse_match.hip.
namespace otb {
constexpr int RCH = 32;

struct Pad {                       // the padded problem: interior Z [N, L, S], dustbins = alpha
  const float* z; float alpha, norm, log_mu_bin, log_nu_bin; int N, L, S;
  __device__ __forceinline__ float zp(int n, int i, int j) const {
    return (i < L && j < S) ? z[((long)n * L + i) * S + j] : alpha;
  }
  __device__ __forceinline__ float log_mu(int i) const { return i < L ? norm : log_mu_bin; }
  __device__ __forceinline__ float log_nu(int j) const { return j < S ? norm : log_nu_bin; }
};

// MODE 0: dZ = G exp(((Zp + u) + v) - norm), du_i = row sum.     MODE 1 (v-step): dZ -= dv_j Pc, du_i (-)= sum_j dv_j Pc
//   one wave per padded row; grid (ceil(N (L+1) / 4)), 256 threads
template <int MODE>
__global__ __launch_bounds__(256) void row_step_kernel(Pad p, const float* __restrict__ G, float* __restrict__ dZ,
                                                       const float* __restrict__ u, const float* __restrict__ v,
                                                       const float* __restrict__ dv, float* __restrict__ du, int accumulate) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)p.N * (p.L + 1)) return;
  const int n = (int)(row / (p.L + 1)), i = (int)(row - (long)n * (p.L + 1));
  const float ui = u[row];
  const float* vn = v + (long)n * (p.S + 1);
  float* dz = dZ + row * (p.S + 1);
  float acc = 0.f;
  for (int j = lane; j <= p.S; j += 64) {
    const float z = p.zp(n, i, j);
    if (MODE == 0) {
      const float d = G[row * (p.S + 1) + j] * expf(((z + ui) + vn[j]) - p.norm);
      dz[j] = d; acc += d;
    } else {
      const float t = dv[(long)n * (p.S + 1) + j] * expf(((z + ui) + vn[j]) - p.log_nu(j));
      dz[j] -= t; acc -= t;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) du[row] = (MODE == 1 && accumulate) ? du[row] + acc : acc;
}

// MODE 0: part = column sums of dZ.     MODE 1 (u-step): dZ -= du_i Pr, part = -sum_i du_i Pr
//   grid (ceil((S+1) / 256), RCH, N), 256 threads; part [N, RCH, S+1]
template <int MODE>
__global__ __launch_bounds__(256) void col_step_kernel(Pad p, float* __restrict__ dZ, const float* __restrict__ u,
                                                       const float* __restrict__ v, const float* __restrict__ du,
                                                       float* __restrict__ part) {
  const int n = blockIdx.z, j = blockIdx.x * 256 + threadIdx.x;
  if (j > p.S) return;
  const int rows = p.L + 1, per = ceil_div(rows, RCH), i0 = blockIdx.y * per, i1 = min(i0 + per, rows);
  const float vj = MODE == 1 ? v[(long)n * (p.S + 1) + j] : 0.f;
  float acc = 0.f;
  for (int i = i0; i < i1; ++i) {
    const long o = ((long)n * rows + i) * (p.S + 1) + j;
    if (MODE == 0) acc += dZ[o];
    else {
      const float t = du[(long)n * rows + i] * expf(((p.zp(n, i, j) + vj) + u[(long)n * rows + i]) - p.log_mu(i));
      dZ[o] -= t; acc -= t;
    }
  }
  part[((long)n * RCH + blockIdx.y) * (p.S + 1) + j] = acc;
}
__global__ void col_merge_kernel(const float* __restrict__ part, int N, int S1, float* __restrict__ dv) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * S1) return;
  const int n = (int)(idx / S1), j = (int)(idx - (long)n * S1);
  float acc = 0.f;
  for (int k = 0; k < RCH; ++k) acc += part[((long)n * RCH + k) * S1 + j];
  dv[idx] = acc;
}

// d bin_score = sum of dZ over the dustbin row (incl. the corner) and the dustbin column.   one workgroup, fixed order
__global__ __launch_bounds__(1024) void dbin_kernel(const float* __restrict__ dZ, int N, int L, int S, float* __restrict__ out) {
  __shared__ double red[1024];
  double acc = 0;
  const long per = (long)(L + 1) * (S + 1);
  for (long k = threadIdx.x; k < (long)N * (L + S + 1); k += 1024) {
    const int n = (int)(k / (L + S + 1)), r = (int)(k - (long)n * (L + S + 1));
    acc += r <= S ? dZ[n * per + (long)L * (S + 1) + r] : dZ[n * per + (long)(r - S - 1) * (S + 1) + S];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)red[0];
}
}  // namespace otb
