// Stand-alone reproducer for the fault behind head_grad_kernel's co-residency errors (round 5): does a packed-fp32 multiply whose LOW
// result selects the HIGH dword of a source pair (v_pk_mul_f32 ... op_sel:[0,1], what hipcc emits for "pair * broadcast scalar" when the
// scalar sits in the odd register of a pair) compute correctly while another wave on the SIMD is issuing MFMAs?
// Workgroups of 4 waves alternate an MFMA phase and a conversion phase; with WGS_PER_CU = 2 two workgroups share a CU (48 KB LDS each) and
// their phases overlap at random.  Every packed product is checked against the scalar product; forms: 0 = op_sel:[0,1] (the suspect),
// 1 = op_sel_hi:[0,1] (control: the form the rest of the library uses), 2 = v_mov-fed op_sel:[0,1] (exact instruction sequence of the kernel).
//   hipcc --offload-arch=gfx950 -O3 -o pk_opsel_probe pk_opsel_probe.hip && ./pk_opsel_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MINWG, int LDS_BYTES>
__global__ __launch_bounds__(256, MINWG) void probe(unsigned* bad, float* sink, int rounds, int form, int mfma_per_round) {
  __shared__ char lds[LDS_BYTES];
  lds[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1.0f + 0.001f * threadIdx.x + i); b[i] = (_Float16)(0.5f - 0.002f * threadIdx.x + i); }
  f16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  unsigned nbad = 0, selmask = 0;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < rounds; ++it) {
    // skew the phases of the waves of a SIMD against each other: waves of odd workgroups start with the conversion phase
    const bool mfma_first = ((blockIdx.x >> 3) + it) & 1;
    for (int ph = 0; ph < 2; ++ph) {
      if ((ph == 0) == mfma_first) {
        for (int u = 0; u < mfma_per_round; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      } else {
#pragma unroll 4
        for (int u = 0; u < 64; ++u) {
          const float x0 = 1.0f + 0.25f * ((lane + u) & 7), x1 = 2.0f + 0.5f * ((lane * 3 + u) & 3);
          const float s0 = 3.0f, s1 = 8192.0f;
          f2 x = {x0, x1}, s = {s0, s1}, r;
          float e0 = x0 * s1, e1 = x1 * s1;            // expected (form 0-2)
          if (form == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(s));
          else if (form == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n" : "=v"(r) : "v"(x), "v"(f2{s1, s0}));
          else if (form == 2) {
            float m0 = x0, m1 = x1;
            asm volatile("v_mov_b32 %0, %3\n v_mov_b32 %1, %4\n v_pk_mul_f32 %2, %5, %6 op_sel:[0,1]"
                         : "=&v"(x[0]), "=&v"(x[1]), "=v"(r) : "v"(m0), "v"(m1), "v"(x), "v"(s));
          } else if (form == 3) {      // fma, src0 low <- high dword: r = (x1 * s0 + 1, x1 * s1 + 1)
            const f2 one = {1.f, 1.f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(x), "v"(s), "v"(one));
            e0 = fmaf(x1, s0, 1.f); e1 = fmaf(x1, s1, 1.f);
          } else if (form == 4) {      // add, src0 low <- high: r = (x1 + s0, x1 + s1)
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(s));
            e0 = x1 + s0; e1 = x1 + s1;
          } else if (form == 5) {      // fma, src1 low <- high: r = (x0 * s1 + 1, x1 * s1 + 1)   (the other instruction of the head_grad staging)
            const f2 one = {1.f, 1.f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(x), "v"(s), "v"(one));
            e0 = fmaf(x0, s1, 1.f); e1 = fmaf(x1, s1, 1.f);
          } else if (form == 9 || form == 10) {   // v_pk_mov_b32 (kv_finalize_kernel carries op_sel:[1,0]): its result must be one of the four
            if (form == 9) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(s));      // (x0|x1, s0|s1) selections --
            else asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(s));               // whichever the ISA defines, ALWAYS the same one
            const int sel = (r[0] == x1 ? 1 : 0) | (r[1] == s1 ? 2 : 0);
            const bool valid = (r[0] == x0 || r[0] == x1) && (r[1] == s0 || r[1] == s1);
            e0 = r[0]; e1 = r[1];
            if (!valid) nbad += 1;
            selmask |= 1u << sel;
          } else if (form == 8) {      // add, src1 low <- high: r = (x0 + s1, x1 + s1)
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(s));
            e0 = x0 + s1; e1 = x1 + s1;
          } else if (form == 6) {      // fma, src2 low <- high: r = (x0 * s0 + s1', ...) with c = (3, 5): (x0 s0 + 5, x1 s1 + 5)
            const f2 c = {3.f, 5.f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(c));
            e0 = fmaf(x0, s0, 5.f); e1 = fmaf(x1, s1, 5.f);
          } else {                     // mul, full swap of src1: op_sel:[0,1] op_sel_hi:[1,0]: (x0 s1, x1 s0)
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(s));
            e0 = x0 * s1; e1 = x1 * s0;
          }
          nbad += (r[0] != e0) + (r[1] != e1);
        }
      }
    }
  }
  float t = 0.f;
  for (int r = 0; r < 16; ++r) t += acc[r];
  if (t == 1234.5f) sink[threadIdx.x] = t + lds[(threadIdx.x * 7) & 255];
  if (nbad) atomicAdd(bad, nbad);
  if (selmask) atomicOr(bad + 1, selmask);          // pk_mov forms: which (lo, hi) selections were observed; more than one bit = inconsistent
}

template <int MINWG, int LDS_BYTES> void run(const char* tag, unsigned* d_bad, float* d_sink, int mfmas) {
  for (int form = 0; form < 11; ++form) {
    unsigned total = 0, selseen = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(d_bad, 0, 8);
      hipLaunchKernelGGL((probe<MINWG, LDS_BYTES>), dim3(1024), dim3(256), 0, 0, d_bad, d_sink, 200, form, mfmas);
      unsigned h[2]; hipMemcpy(h, d_bad, 8, hipMemcpyDeviceToHost);
      total += h[0]; selseen |= h[1];
    }
    printf("%s, %2d MFMAs per round, form %d (%s): wrong packed products in 5 launches of 1024 workgroups: %u%s\n", tag, mfmas, form,
           form == 0 ? "mul op_sel:[0,1]" : form == 1 ? "mul op_sel_hi control" : form == 2 ? "v_mov-fed mul op_sel:[0,1]" : form == 3 ? "fma op_sel:[1,0,0]" :
           form == 4 ? "add op_sel:[1,0]" : form == 5 ? "fma op_sel:[0,1,0]" : form == 8 ? "add op_sel:[0,1]" : form == 9 ? "pk_mov op_sel:[1,0] (invalid results; see selections)" : form == 10 ? "pk_mov op_sel:[0,1] (invalid results; see selections)" : form == 6 ? "fma op_sel:[0,0,1]" : "mul op_sel:[0,1] op_sel_hi:[1,0]", total,
           form >= 9 ? (selseen == 1 || selseen == 2 || selseen == 4 || selseen == 8 ? "  [one selection observed: consistent]" : "  [SEVERAL selections observed: inconsistent]") : "");
  }
}
int main() {
  unsigned* d_bad; float* d_sink;
  hipMalloc(&d_bad, 8); hipMalloc(&d_sink, 4096);
  run<2, 49152>("two workgroups per CU", d_bad, d_sink, 48);
  run<1, 98304>("one workgroup per CU ", d_bad, d_sink, 48);
  run<2, 49152>("two workgroups per CU", d_bad, d_sink, 0);
  return 0;
}
