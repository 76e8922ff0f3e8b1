// LoFTREncoderLayer / LocalFeatureTransformer orchestration (host side of the C-ABI).
//   reference: src/loftr/loftr_module/transformer.py:35-58 (layer), :80-101 (layer schedule).
//
// Data flow of one coarse layer call (x attends to source; SP = split-fp16 GEMM operand format):
//   source_sp --proj(k,v) + KV reduction in the epilogue--> partials --finalize--> KV, Ksum, P (= KV folded into merge, SP)
//   x_sp --proj(q) + normaliser z--> Q' SP --GEMM with P + LN(norm1)--> message SP
//   cat[x_sp, message] --GEMM mlp.0 + ReLU--> hidden SP --GEMM mlp.2 + LN(norm2) + x--> x' (fp32 and SP)
// The residual stream is kept in fp32 (in place in the caller's buffers); its SP mirror only feeds GEMMs.
#include "linear.h"
#include "attention.h"
#include "coarse_plan.h"

namespace {

constexpr int MAX_LAYERS = 16;

struct LayerSp {                 // SP copies of one layer's matrices (workspace) + the fp32 originals needed
  const sp_t *q, *k, *v, *merge, *mlp0, *mlp2;
  const sp_t* kv;                // C == 256: [2C, C], rows interleaved per head [K_h | V_h] (linear.h: ProjKVArgs)
  const float *q_s, *k_s, *v_s, *merge_s, *mlp0_s, *mlp2_s, *kv_s;   // inverse per-row power-of-two scales (gemm.h)
  const float* merge_f32;
  const float *n1w, *n1b, *n2w, *n2b;
};

struct EncoderWs {
  sp_t* q;                       // [rows_l, C]  coarse: z-scaled Q as SP; fine: Q fp32 (same size)
  float *k, *v;                  // [rows_s, C]  fp32
  sp_t *msg, *msgn, *hid;        // [rows_l, C] (fine only), [rows_l, C], [rows_l, 2C]
  void* attn; size_t attn_bytes;
  void* attn2;                   // a second block of the same size: two calls' KV / P alive at once (scheduled coarse transformer)
  bool ok;
};

// q k v merge (4) + mlp0 (4) + mlp2 (2) + interleaved kv (2) matrices, then their 9C inverse row scales (padded to 16C)
size_t weights_sp_dwords(int C) { return (size_t)12 * C * C + (size_t)16 * C; }
constexpr int JOBS_PER_LAYER = 6 + 16;

size_t encoder_ws_bytes(int nb, int L, int S, int C) {
  size_t rows_l = (size_t)nb * L, rows_s = (size_t)nb * S;
  size_t b = 0;
  b += 3 * align_up(rows_l * C * 4, 256);          // q, msg, msgn
  b += 2 * align_up(rows_s * C * 4, 256);          // k, v
  b += align_up(rows_l * 2 * C * 4, 256);          // hidden
  b += 2 * align_up(attention_workspace_bytes(nb, S, C), 256);
  return b + 2048;
}

EncoderWs carve(WsAlloc& wa, int nb, int L, int S, int C) {
  EncoderWs e;
  size_t rows_l = (size_t)nb * L, rows_s = (size_t)nb * S;
  e.q = wa.take<sp_t>(rows_l * C);
  e.msg = wa.take<sp_t>(rows_l * C);
  e.msgn = wa.take<sp_t>(rows_l * C);
  e.k = wa.take<float>(rows_s * C);
  e.v = wa.take<float>(rows_s * C);
  e.hid = wa.take<sp_t>(rows_l * 2 * C);
  e.attn_bytes = attention_workspace_bytes(nb, S, C);
  e.attn = wa.take<char>(e.attn_bytes);
  e.attn2 = wa.take<char>(e.attn_bytes);
  e.ok = wa.ok();
  return e;
}

bool weights_ok(const loftr_layer_weights& w) {
  return w.q_proj && w.k_proj && w.v_proj && w.merge && w.mlp0 && w.mlp2 && w.norm1_w && w.norm1_b &&
         w.norm2_w && w.norm2_b;
}

// queue the fp32 -> SP conversion of one layer's six matrices; returns the SP view
LayerSp stage_layer(const loftr_layer_weights& w, sp_t* dst, int C, SpJobs& jobs) {
  LayerSp l;
  float* sc = reinterpret_cast<float*>(dst + (size_t)12 * C * C);      // the layer's scale block behind its matrices
  auto add = [&](const float* src, int rows, int K, float* inv) {
    const int i = jobs.n++;
    jobs.src[i] = src; jobs.dst[i] = dst; jobs.rows[i] = rows; jobs.K[i] = K; jobs.ld[i] = K; jobs.inv_scale[i] = inv;
    sp_t* r = dst;
    dst += (size_t)rows * K;
    return r;
  };
  l.q_s = sc; l.k_s = sc + C; l.v_s = sc + 2 * C; l.merge_s = sc + 3 * C; l.mlp0_s = sc + 4 * C; l.mlp2_s = sc + 6 * C;
  l.kv_s = sc + 7 * C;
  l.q = add(w.q_proj, C, C, sc);
  l.k = add(w.k_proj, C, C, sc + C);
  l.v = add(w.v_proj, C, C, sc + 2 * C);
  l.merge = add(w.merge, C, C, sc + 3 * C);
  l.mlp0 = add(w.mlp0, 2 * C, 2 * C, sc + 4 * C);
  l.mlp2 = add(w.mlp2, C, 2 * C, sc + 6 * C);
  l.kv = nullptr;
  if (C == 256) {                // per head: 32 rows of k_proj then 32 rows of v_proj
    l.kv = dst;
    for (int h = 0; h < 8; ++h) {
      add(w.k_proj + (size_t)h * 32 * C, 32, C, sc + 7 * C + h * 64);
      add(w.v_proj + (size_t)h * 32 * C, 32, C, sc + 7 * C + h * 64 + 32);
    }
  }
  l.merge_f32 = w.merge;
  l.n1w = w.norm1_w; l.n1b = w.norm1_b; l.n2w = w.norm2_w; l.n2b = w.norm2_b;
  return l;
}

// ---- coarse level (C = 256), the two halves of a layer call as separate steps ------------------------------------------------
// K / V projections of the source with the KV / Ksum reduction in their epilogue (K, V never reach HBM), then the finalize:
// sum of the row-tile partials + P (KV folded into merge).  Results live in the attention block `attn`.
int coarse_kv(const sp_t* src_sp, const uint8_t* src_mask, const LayerSp& w, int nb, int S, int C, int H, void* attn, size_t attn_bytes,
              const float** kv, const sp_t** pm, hipStream_t st) {
  float* part = attention_part_buffer(attn, attn_bytes, nb, S);
  if (!part) return LOFTR_ERR_WORKSPACE;
  int rc;
  ProjKVArgs pkv{src_sp, S, C, nb, w.kv, src_mask, 1.f / (float)S, part, ceil_div(S, 128), w.kv_s};
  if ((rc = launch_proj_kv(pkv, st))) return rc;
  return launch_attention_finalize(w.merge_f32, nb, S, C, H, attn, attn_bytes, kv, pm, st);
}
// everything on the x side of the layer (encoder_fused.hip), in place
EncoderXArgs coarse_x(float* x_f32, sp_t* x_sp, const uint8_t* x_mask, const LayerSp& w, int nb, int L, int S, int C, const float* kv,
                      const sp_t* pm, int skip_padded = 0) {
  return EncoderXArgs{x_sp, x_f32, x_f32, x_sp, nb, L, C, w.q, pm, (long)C * C, w.mlp0, w.mlp2, w.q_s, w.mlp0_s, w.mlp2_s,
                      kv, x_mask, w.n1w, w.n1b, w.n2w, w.n2b, (float)S, 1e-6f, 1.f / ATTN_P_SCALE, 1e-5f, skip_padded};
}

// LocalFeatureTransformer with layers [self, cross] * P at C = 256 (transformer.py:80-101) as a schedule of launches.
// Calls: A_i = self(feat0), B_i = self(feat1), C_i = cross(feat0 <- feat1), D_i = cross(feat1 <- updated feat0).  The reference's
// order A B C D is a chain C_i -> D_i -> B_i+1 -> C_i+1 with A_i+1 hanging off C_i: A_i+1 needs nothing D_i or B_i+1 produce, and
// writes only feat0, which they read through the K V summary taken BEFORE the launch.  A call's time is whole rounds of 256
// workgroups (encoder_fused.hip: Args2) and a cross call of the batch-8 configuration is 304, so A_i+1 rides in the idle slots
// of D_i's last round and the rest of it next to B_i+1:
//     [A_0 B_0]  { [C_i]  [D_i + head of A_i+1]  [B_i+1 + tail of A_i+1] } ...  [C_P-1] [D_P-1]
// 25 rounds instead of 28 at batch 8.  Every workgroup computes what it computed before: results are bit-identical to the call-by-
// call order.  LOFTR_ENCODER_SCHEDULE=0 keeps that order (A/B).  Returns LOFTR_ERR_UNSUPPORTED when the shapes are not the fused kernel's.
int coarse_transformer_scheduled(float* feat0, float* feat1, sp_t* sp0, sp_t* sp1, const uint8_t* mask0, const uint8_t* mask1,
                                 const LayerSp* lw, int P, bool stacked, int N, int L, int S, int C, int H, const EncoderWs& e,
                                 hipStream_t st, int skip = 0) {
  int rc;
  const float* kvA; const sp_t* pmA; const float* kvB; const sp_t* pmB;
  {
    // does the fused kernel take these shapes at all?  (probe with dummy summaries: only shapes are looked at)
    EncoderXArgs probe = coarse_x(feat0, sp0, mask0, lw[0], N, L, L, C, reinterpret_cast<const float*>(e.attn), reinterpret_cast<const sp_t*>(e.attn));
    if (encoder_x_workgroups(probe) == 0) return LOFTR_ERR_UNSUPPORTED;
  }
  // ---- A_0, B_0
  if (stacked) {
    if ((rc = coarse_kv(sp0, mask0, lw[0], 2 * N, L, C, H, e.attn, e.attn_bytes, &kvA, &pmA, st))) return rc;
    if ((rc = launch_encoder_x(coarse_x(feat0, sp0, mask0, lw[0], 2 * N, L, L, C, kvA, pmA, skip), st))) return rc;
  } else {
    if ((rc = coarse_kv(sp0, mask0, lw[0], N, L, C, H, e.attn, e.attn_bytes, &kvA, &pmA, st))) return rc;
    if ((rc = coarse_kv(sp1, mask1, lw[0], N, S, C, H, e.attn2, e.attn_bytes, &kvB, &pmB, st))) return rc;
    const EncoderXArgs a = coarse_x(feat0, sp0, mask0, lw[0], N, L, L, C, kvA, pmA, skip), b = coarse_x(feat1, sp1, mask1, lw[0], N, S, S, C, kvB, pmB, skip);
    if ((rc = launch_encoder_x2(a, 0, encoder_x_workgroups(a), b, 0, encoder_x_workgroups(b), st))) return rc;
  }
  for (int i = 0; i < P; ++i) {
    const LayerSp& wc = lw[2 * i + 1];
    // ---- C_i: feat0 attends to feat1
    if ((rc = coarse_kv(sp1, mask1, wc, N, S, C, H, e.attn, e.attn_bytes, &kvA, &pmA, st))) return rc;
    if ((rc = launch_encoder_x(coarse_x(feat0, sp0, mask0, wc, N, L, S, C, kvA, pmA, skip), st))) return rc;
    // ---- D_i: feat1 attends to the UPDATED feat0 (transformer.py:96-97) ...
    if ((rc = coarse_kv(sp0, mask0, wc, N, L, C, H, e.attn, e.attn_bytes, &kvA, &pmA, st))) return rc;
    const EncoderXArgs d = coarse_x(feat1, sp1, mask1, wc, N, S, L, C, kvA, pmA, skip);
    const int nd = encoder_x_workgroups(d);
    if (i + 1 == P) {
      if ((rc = launch_encoder_x(d, st))) return rc;
      break;
    }
    // ---- ... with the head of A_i+1 (self on the same updated feat0) in the slots its last round leaves idle
    const LayerSp& ws = lw[2 * i + 2];
    if ((rc = coarse_kv(sp0, mask0, ws, N, L, C, H, e.attn2, e.attn_bytes, &kvB, &pmB, st))) return rc;
    const EncoderXArgs a = coarse_x(feat0, sp0, mask0, ws, N, L, L, C, kvB, pmB, skip);
    const int na = encoder_x_workgroups(a);
    int head = (256 - nd % 256) % 256 / 8 * 8;
    if (head > na) head = na;
    if ((rc = launch_encoder_x2(d, 0, nd, a, 0, head, st))) return rc;
    // ---- B_i+1 (self on feat1, which D_i has just finished) + the tail of A_i+1
    if ((rc = coarse_kv(sp1, mask1, ws, N, S, C, H, e.attn, e.attn_bytes, &kvA, &pmA, st))) return rc;
    const EncoderXArgs b = coarse_x(feat1, sp1, mask1, ws, N, S, S, C, kvA, pmA, skip);
    if ((rc = launch_encoder_x2(b, 0, encoder_x_workgroups(b), a, head, na - head, st))) return rc;
  }
  return LOFTR_OK;
}

int encoder_layer(const float* x_f32, const sp_t* x_sp, const sp_t* src_sp, bool self,
                  const uint8_t* x_mask, const uint8_t* src_mask, const LayerSp& w,
                  float* out_f32, sp_t* out_sp, int nb, int L, int S, int C, int H,
                  const EncoderWs& e, hipStream_t st, int skip_padded = 0) {
  if (nb <= 0) return LOFTR_OK;
  const int Ml = nb * L, Ms = nb * S;
  const float inv_s = 1.f / (float)S;                 // values / v_length, linear_attention.py:41-42
  const float attn_eps = 1e-6f;                       // LinearAttention(eps=1e-6), linear_attention.py:15
  int rc;
  if (C == 256) {
    // k, v projections of the source with the KV / Ksum reduction in their epilogue (K, V never reach HBM)
    // -> finalize: sum of the row-tile partials + P (KV folded into merge)
    float* part = attention_part_buffer(e.attn, e.attn_bytes, nb, S);
    if (!part) return LOFTR_ERR_WORKSPACE;
    ProjKVArgs pkv{src_sp, S, C, nb, w.kv, src_mask, inv_s, part, ceil_div(S, 128), w.kv_s};
    if ((rc = launch_proj_kv(pkv, st))) return rc;
    const float* kv = nullptr; const sp_t* pm = nullptr;
    if ((rc = launch_attention_finalize(w.merge_f32, nb, S, C, H, e.attn, e.attn_bytes, &kv, &pm, st))) return rc;
    (void)Ms;
    // round 3: everything on the x side of the layer in ONE launch, tokens stationary in registers (encoder_fused.hip)
    {
      EncoderXArgs fx{x_sp, x_f32, out_f32, out_sp, nb, L, C, w.q, pm, (long)C * C, w.mlp0, w.mlp2, w.q_s, w.mlp0_s, w.mlp2_s,
                      kv, x_mask, w.n1w, w.n1b, w.n2w, w.n2b, (float)S, attn_eps, 1.f / ATTN_P_SCALE, 1e-5f, skip_padded};
      return launch_encoder_x(fx, st);
    }
  } else {
    float* qf = reinterpret_cast<float*>(e.q);
    if (self) {
      ProjArgs p{x_sp, Ml, C, 1, 3, {w.q, w.k, w.v}, {qf, e.k, e.v}, {0, 1, 2}, x_mask, inv_s, nullptr, 0.f, 0.f,
                 {w.q_s, w.k_s, w.v_s}};
      if ((rc = launch_proj(p, st))) return rc;
    } else {
      ProjArgs pq{x_sp, Ml, C, 1, 1, {w.q, nullptr, nullptr}, {qf, nullptr, nullptr}, {0, 0, 0}, x_mask, inv_s,
                  nullptr, 0.f, 0.f, {w.q_s, nullptr, nullptr}};
      if ((rc = launch_proj(pq, st))) return rc;
      ProjArgs pkv{src_sp, Ms, C, 1, 2, {w.k, w.v, nullptr}, {e.k, e.v, nullptr}, {1, 2, 0}, src_mask, inv_s,
                   nullptr, 0.f, 0.f, {w.k_s, w.v_s, nullptr}};
      if ((rc = launch_proj(pkv, st))) return rc;
    }
    if ((rc = launch_attention_small(qf, e.k, e.v, e.msg, nb, L, S, C, H, st))) return rc;
    // message = norm1(merge(message))                                   transformer.py:51-52
    LinearLNArgs m{asrc_plain(e.msg, C), w.merge, C, w.n1w, w.n1b, nullptr, nullptr, e.msgn, Ml, C, C, 1e-5f, 1, 0,
                   w.merge_s, 0.f};
    if ((rc = launch_linear_ln(m, st))) return rc;
  }
  // hidden = relu(mlp.0(cat[x, message]))                             transformer.py:55
  LinearArgs h{asrc_cat(x_sp, e.msgn, C, C), w.mlp0, 2 * C, nullptr, e.hid, 2 * C, Ml, 2 * C, 2 * C, nullptr, 0, 1, true,
               w.mlp0_s, nullptr};
  if ((rc = launch_linear(h, st))) return rc;
  // out = x + norm2(mlp.2(hidden))                                    transformer.py:55-58
  LinearLNArgs o{asrc_plain(e.hid, 2 * C), w.mlp2, 2 * C, w.n2w, w.n2b, x_f32, out_f32, out_sp, Ml, C, 2 * C, 1e-5f, 1, 0,
                 w.mlp2_s, 0.f};
  return launch_linear_ln(o, st);
}

}  // namespace

extern "C" size_t loftr_encoder_workspace_bytes(int nb, int L, int S, int C) {
  if (nb <= 0 || L <= 0 || S <= 0 || C <= 0) return 0;
  const int m = L > S ? L : S;
  size_t b = encoder_ws_bytes(nb, m, m, C) + 2 * align_up((size_t)nb * m * C * 4, 256) +
             align_up(weights_sp_dwords(C) * 4 * MAX_LAYERS, 256) + 4096;
  // the persistent coarse transformer (loftr_transformer_fwd_planned): counters, per-(call, pair) partials / KV / P of nb / 2 pairs
  const PctShape ps{PCT_MAX_LAYERS, nb / 2, {L, S}};
  if (C == 256 && ps.ok()) b += align_up(pct_ws_bytes(ps), 256);
  return b;
}

extern "C" int loftr_encoder_layer_fwd(const float* x, const float* source, const uint8_t* x_mask,
                                       const uint8_t* source_mask, const loftr_layer_weights* w,
                                       float* out, int nb, int L, int S, int C, int H, void* ws,
                                       size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(x && source && w && out && nb >= 0 && L > 0 && S > 0 && (ws || nb == 0));
  LOFTR_CHECK_ARG(weights_ok(*w));
  if (!((C == 256 || C == 128) && H == 8)) return LOFTR_ERR_UNSUPPORTED;
  if (nb == 0) return LOFTR_OK;
  hipStream_t st = (hipStream_t)stream;
  WsAlloc wa(ws, ws_bytes);
  sp_t* x_sp = wa.take<sp_t>((size_t)nb * L * C);
  sp_t* s_sp = wa.take<sp_t>((size_t)nb * S * C);
  sp_t* w_sp = wa.take<sp_t>(weights_sp_dwords(C));
  EncoderWs e = carve(wa, nb, L, S, C);
  if (!e.ok) return LOFTR_ERR_WORKSPACE;
  const bool self = (x == source) && (x_mask == source_mask);
  SpJobs jobs; jobs.n = 0;
  auto add = [&](const float* src, sp_t* dst, long rows) {
    const int i = jobs.n++;
    jobs.src[i] = src; jobs.dst[i] = dst; jobs.rows[i] = (int)rows; jobs.K[i] = C; jobs.ld[i] = C;
  };
  add(x, x_sp, (long)nb * L);
  if (!self) add(source, s_sp, (long)nb * S);
  const LayerSp lw = stage_layer(*w, w_sp, C, jobs);
  int rc;
  if ((rc = launch_sp_convert(jobs, st))) return rc;
  return encoder_layer(x, x_sp, self ? x_sp : s_sp, self, x_mask, source_mask, lw, out, nullptr, nb, L, S, C, H, e, st);
}

namespace {
// fp32 -> SP (row-scaled) conversion of every matrix of `n_layers` layers into `w_sp`, batched in launches of <= SP_MAX_JOBS
int convert_layers(const loftr_layer_weights* layers, int n_layers, int C, sp_t* w_sp, SpJobs& jobs, LayerSp* lw, hipStream_t st) {
  int rc;
  for (int i = 0; i < n_layers; ++i) {
    if (jobs.n + JOBS_PER_LAYER > SP_MAX_JOBS) {
      if ((rc = launch_sp_convert(jobs, st))) return rc;
      jobs = SpJobs();
    }
    lw[i] = stage_layer(layers[i], w_sp + weights_sp_dwords(C) * i, C, jobs);
  }
  return launch_sp_convert(jobs, st);
}
// the SP views of an already converted block (same layout as convert_layers produces)
void view_layers(const loftr_layer_weights* layers, int n_layers, int C, sp_t* w_sp, LayerSp* lw) {
  for (int i = 0; i < n_layers; ++i) {
    SpJobs scratch;
    lw[i] = stage_layer(layers[i], w_sp + weights_sp_dwords(C) * i, C, scratch);     // pointer arithmetic only; nothing launched
  }
}
}  // namespace

extern "C" size_t loftr_transformer_prepared_bytes(int n_layers, int C) {
  if (n_layers <= 0 || C <= 0) return 0;
  return align_up(weights_sp_dwords(C) * 4 * (size_t)n_layers, 256) + 256;
}

extern "C" int loftr_transformer_prepare(const loftr_layer_weights* layers, int n_layers, int C, void* prepared,
                                         size_t prepared_bytes, void* stream) {
  LOFTR_CHECK_ARG(layers && prepared && n_layers > 0);
  if (!(C == 256 || C == 128) || n_layers > MAX_LAYERS) return LOFTR_ERR_UNSUPPORTED;
  for (int i = 0; i < n_layers; ++i) LOFTR_CHECK_ARG(weights_ok(layers[i]));
  if (prepared_bytes < loftr_transformer_prepared_bytes(n_layers, C)) return LOFTR_ERR_WORKSPACE;
  LayerSp lw[MAX_LAYERS];
  SpJobs jobs;
  return convert_layers(layers, n_layers, C, reinterpret_cast<sp_t*>(prepared), jobs, lw, (hipStream_t)stream);
}

static int transformer_fwd(float* feat0, float* feat1, const uint8_t* mask0,
                           const uint8_t* mask1, const loftr_layer_weights* layers,
                           const int* layer_is_cross, int n_layers, int N, int L, int S,
                           int C, int H, const void* prepared, size_t prepared_bytes, void* ws, size_t ws_bytes,
                           const void* plan, size_t plan_bytes, int plan_order, void* diag, size_t diag_bytes, void* stream,
                           int skip_padded = 0) {
  LOFTR_CHECK_ARG(feat0 && feat1 && layers && layer_is_cross && n_layers >= 0 && N >= 0 && L > 0 && S > 0);
  LOFTR_CHECK_ARG((mask0 == nullptr) == (mask1 == nullptr));
  if (!((C == 256 || C == 128) && H == 8) || n_layers > MAX_LAYERS) return LOFTR_ERR_UNSUPPORTED;
  if (N == 0 || n_layers == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(ws != nullptr);
  if (prepared && prepared_bytes < loftr_transformer_prepared_bytes(n_layers, C)) return LOFTR_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n_layers; ++i) LOFTR_CHECK_ARG(weights_ok(layers[i]));
  // The two self-attention calls of a layer are independent (transformer.py:92-94): when the two
  // feature sets are one contiguous [2N, L, C] buffer they run as a single batch of 2N.
  const bool stacked = (L == S) && (feat1 == feat0 + (size_t)N * L * C) &&
                       (mask0 == nullptr || mask1 == mask0 + (size_t)N * L);
  WsAlloc wa(ws, ws_bytes);
  sp_t* sp0 = wa.take<sp_t>((size_t)N * (L + S) * C);        // SP mirror of [feat0 ; feat1], contiguous
  sp_t* sp1 = sp0 + (size_t)N * L * C;
  sp_t* w_sp = wa.take<sp_t>(weights_sp_dwords(C) * n_layers);
  EncoderWs e = carve(wa, 2 * N, L > S ? L : S, L > S ? L : S, C);
  if (!e.ok) return LOFTR_ERR_WORKSPACE;
  int rc;
  LayerSp lw[MAX_LAYERS];
  // layer matrices -> SP, unless the caller hands in the block loftr_transformer_prepare built once for these weights
  // (inference weights are constant)
  bool weights_ready = false;
  if (prepared) {
    view_layers(layers, n_layers, C, reinterpret_cast<sp_t*>(const_cast<void*>(prepared)), lw);
    weights_ready = true;
  }
  // Fine level (loftr.py:71-72): [self, cross] on window pairs of <= 32 tokens is local to a match -> ONE launch, the windows
  // never leave the registers between the layers (fine_fused.hip); no SP mirror of the residual stream is needed.
  if (C == 128 && n_layers == 2 && !layer_is_cross[0] && layer_is_cross[1] && L == S && L <= 32 && !mask0) {
    if (!weights_ready) {
      SpJobs wj;
      if ((rc = convert_layers(layers, n_layers, C, w_sp, wj, lw, st))) return rc;
      weights_ready = true;
    }
    FinePairArgs fp{};
    fp.f0 = feat0; fp.f1 = feat1; fp.M = N; fp.T = L; fp.C = C; fp.attn_eps = 1e-6f; fp.ln_eps = 1e-5f;
    for (int l = 0; l < 2; ++l) {
      fp.wq[l] = lw[l].q; fp.wk[l] = lw[l].k; fp.wv[l] = lw[l].v; fp.wm[l] = lw[l].merge; fp.w0[l] = lw[l].mlp0; fp.w2[l] = lw[l].mlp2;
      fp.sq[l] = lw[l].q_s; fp.sk[l] = lw[l].k_s; fp.sv[l] = lw[l].v_s; fp.sm[l] = lw[l].merge_s; fp.s0[l] = lw[l].mlp0_s; fp.s2[l] = lw[l].mlp2_s;
      fp.g1[l] = lw[l].n1w; fp.b1[l] = lw[l].n1b; fp.g2[l] = lw[l].n2w; fp.b2[l] = lw[l].n2b;
    }
    if ((rc = launch_fine_pair(fp, st)) != LOFTR_ERR_UNSUPPORTED) return rc;
  }
  {
    // residual stream -> SP mirror (in the same launch as the layer matrices when those are still to be converted)
    SpJobs jobs;
    auto add = [&](const float* src, sp_t* dst, long rows) {
      const int i = jobs.n++;
      jobs.src[i] = src; jobs.dst[i] = dst; jobs.rows[i] = (int)rows; jobs.K[i] = C; jobs.ld[i] = C;
    };
    add(feat0, sp0, (long)N * L);
    add(feat1, sp1, (long)N * S);
    if (weights_ready) {
      if ((rc = launch_sp_convert(jobs, st))) return rc;
    } else if ((rc = convert_layers(layers, n_layers, C, w_sp, jobs, lw, st))) return rc;
  }
  {
    bool pattern = C == 256 && n_layers >= 2 && n_layers % 2 == 0;
    for (int i = 0; pattern && i < n_layers; ++i) pattern = (layer_is_cross[i] != 0) == ((i & 1) != 0);
    // coarse level with the stock layer pattern and a plan: ONE persistent launch, dependencies per pair and tile (encoder_fused.hip)
    if (plan && pattern && n_layers <= PCT_MAX_LAYERS) {
      PctLaunch pl{};
      pl.shape = PctShape{n_layers, N, {L, S}};
      if (!pl.shape.ok()) return LOFTR_ERR_UNSUPPORTED;
      pl.f32[0] = feat0; pl.f32[1] = feat1; pl.sp[0] = sp0; pl.sp[1] = sp1; pl.mask[0] = mask0; pl.mask[1] = mask1;
      for (int i = 0; i < n_layers; ++i) {
        const LayerSp& w = lw[i];
        pl.layer[i] = PctLayerPtrs{w.q, w.mlp0, w.mlp2, w.kv, w.q_s, w.mlp0_s, w.mlp2_s, w.kv_s, w.merge_f32, w.n1w, w.n1b, w.n2w, w.n2b};
      }
      pl.plan = plan; pl.plan_bytes = plan_bytes; pl.plan_signature = pl.shape.signature(plan_order);
      pl.ws = wa.take<char>(pct_ws_bytes(pl.shape)); pl.ws_bytes = pct_ws_bytes(pl.shape);
      if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
      if (diag && diag_bytes >= 16) {
        pl.status = reinterpret_cast<unsigned*>(diag);
        if (diag_bytes >= 16 + 32 * pl.shape.n_items()) pl.trace = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(diag) + 16);
      }
      return launch_coarse_persistent(pl, st);
    }
    // ... without a plan: the scheduled launches (bit-identical to the call order, fewer rounds of workgroups)
    const bool sched_on = loftr_debug_value(LOFTR_DBG_ENCODER_SCHEDULE) != 0;
    if (sched_on && pattern) {
      rc = coarse_transformer_scheduled(feat0, feat1, sp0, sp1, mask0, mask1, lw, n_layers / 2, stacked, N, L, S, C, H, e, st, skip_padded);
      if (rc != LOFTR_ERR_UNSUPPORTED) return rc;
    }
  }
  for (int i = 0; i < n_layers; ++i) {
    if (!layer_is_cross[i]) {
      if (stacked) {
        if ((rc = encoder_layer(feat0, sp0, sp0, true, mask0, mask0, lw[i], feat0, sp0, 2 * N, L, L, C, H, e, st, skip_padded))) return rc;
      } else {
        if ((rc = encoder_layer(feat0, sp0, sp0, true, mask0, mask0, lw[i], feat0, sp0, N, L, L, C, H, e, st, skip_padded))) return rc;
        if ((rc = encoder_layer(feat1, sp1, sp1, true, mask1, mask1, lw[i], feat1, sp1, N, S, S, C, H, e, st, skip_padded))) return rc;
      }
    } else {
      // sequential dependency kept: feat1 attends to the UPDATED feat0 (transformer.py:96-97)
      if ((rc = encoder_layer(feat0, sp0, sp1, false, mask0, mask1, lw[i], feat0, sp0, N, L, S, C, H, e, st, skip_padded))) return rc;
      if ((rc = encoder_layer(feat1, sp1, sp0, false, mask1, mask0, lw[i], feat1, sp1, N, S, L, C, H, e, st, skip_padded))) return rc;
    }
  }
  return LOFTR_OK;
}

extern "C" int loftr_transformer_fwd(float* feat0, float* feat1, const uint8_t* mask0,
                                     const uint8_t* mask1, const loftr_layer_weights* layers,
                                     const int* layer_is_cross, int n_layers, int N, int L, int S,
                                     int C, int H, const void* prepared, size_t prepared_bytes, void* ws, size_t ws_bytes,
                                     void* stream) {
  return transformer_fwd(feat0, feat1, mask0, mask1, layers, layer_is_cross, n_layers, N, L, S, C, H, prepared, prepared_bytes, ws, ws_bytes,
                         nullptr, 0, 0, nullptr, 0, stream);
}

extern "C" int loftr_transformer_fwd_padded(float* feat0, float* feat1, const uint8_t* mask0,
                                            const uint8_t* mask1, const loftr_layer_weights* layers,
                                            const int* layer_is_cross, int n_layers, int N, int L, int S,
                                            int C, int H, const void* prepared, size_t prepared_bytes, void* ws, size_t ws_bytes,
                                            int skip_padded_tiles, void* stream) {
  return transformer_fwd(feat0, feat1, mask0, mask1, layers, layer_is_cross, n_layers, N, L, S, C, H, prepared, prepared_bytes, ws, ws_bytes,
                         nullptr, 0, 0, nullptr, 0, stream, skip_padded_tiles != 0 && mask0 != nullptr);
}

extern "C" int loftr_transformer_fwd_planned(float* feat0, float* feat1, const uint8_t* mask0,
                                             const uint8_t* mask1, const loftr_layer_weights* layers,
                                             const int* layer_is_cross, int n_layers, int N, int L, int S,
                                             int C, int H, const void* prepared, size_t prepared_bytes, void* ws, size_t ws_bytes,
                                             const void* plan, size_t plan_bytes, int plan_order, void* diag, size_t diag_bytes,
                                             void* stream) {
  LOFTR_CHECK_ARG(plan != nullptr && (plan_order == 0 || plan_order == 1));
  return transformer_fwd(feat0, feat1, mask0, mask1, layers, layer_is_cross, n_layers, N, L, S, C, H, prepared, prepared_bytes, ws, ws_bytes,
                         plan, plan_bytes, plan_order, diag, diag_bytes, stream);
}

// ------------------------------------------------------------------------------------------
extern "C" size_t loftr_linear_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return align_up((size_t)M * ceil32(K) * 4, 256) + align_up((size_t)N * ceil32(K) * 4, 256) +
         align_up((size_t)M * 4, 256) + align_up((size_t)N * 4, 256) + 1024;
}

extern "C" int loftr_linear_fwd(const float* a, const float* w, float* out, int M, int N, int K,
                                void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(a && w && out && M >= 0 && N > 0 && K > 0);
  if (M == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(ws != nullptr);
  hipStream_t st = (hipStream_t)stream;
  const int Kp = ceil32(K);
  WsAlloc wa(ws, ws_bytes);
  sp_t* a_sp = wa.take<sp_t>((size_t)M * Kp);
  sp_t* w_sp = wa.take<sp_t>((size_t)N * Kp);
  float* a_inv = wa.take<float>((size_t)M);          // per-row power-of-two scales of both operands (gemm.h): the product is
  float* w_inv = wa.take<float>((size_t)N);          // accurate relative to the ROWS' magnitudes, whatever the tensors' scale
  if (!wa.ok()) return LOFTR_ERR_WORKSPACE;
  SpJobs jobs; jobs.n = 2;
  jobs.src[0] = a; jobs.dst[0] = a_sp; jobs.rows[0] = M; jobs.K[0] = K; jobs.ld[0] = K; jobs.inv_scale[0] = a_inv;
  jobs.src[1] = w; jobs.dst[1] = w_sp; jobs.rows[1] = N; jobs.K[1] = K; jobs.ld[1] = K; jobs.inv_scale[1] = w_inv;
  int rc;
  if ((rc = launch_sp_convert(jobs, st))) return rc;
  LinearArgs p{asrc_plain(a_sp, Kp), w_sp, Kp, out, nullptr, N, M, N, Kp, nullptr, 0, 1, false, w_inv, a_inv};
  return launch_linear(p, st);
}
