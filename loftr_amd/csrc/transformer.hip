// LoFTREncoderLayer / LocalFeatureTransformer orchestration (host side of the C-ABI).
//   reference: src/loftr/loftr_module/transformer.py:35-58 (layer), :80-101 (layer schedule).
#include "linear.h"
#include "attention.h"

namespace {

struct EncoderWs {
  float *q, *k, *v, *msg, *msgn, *hid;
  void* attn; size_t attn_bytes;
  bool ok;
};

size_t encoder_ws_bytes(int nb, int L, int S, int C) {
  size_t rows_l = (size_t)nb * L, rows_s = (size_t)nb * S;
  size_t b = 0;
  b += 3 * align_up(rows_l * C * sizeof(float), 256);        // q, msg, msgn
  b += 2 * align_up(rows_s * C * sizeof(float), 256);        // k, v
  b += align_up(rows_l * 2 * C * sizeof(float), 256);        // hidden
  b += attention_workspace_bytes(nb, S, C);
  return b + 2048;
}

EncoderWs carve(void* ws, size_t bytes, int nb, int L, int S, int C) {
  WsAlloc wa(ws, bytes);
  EncoderWs e;
  size_t rows_l = (size_t)nb * L, rows_s = (size_t)nb * S;
  e.q = wa.take<float>(rows_l * C);
  e.msg = wa.take<float>(rows_l * C);
  e.msgn = wa.take<float>(rows_l * C);
  e.k = wa.take<float>(rows_s * C);
  e.v = wa.take<float>(rows_s * C);
  e.hid = wa.take<float>(rows_l * 2 * C);
  e.attn_bytes = attention_workspace_bytes(nb, S, C);
  e.attn = wa.take<char>(e.attn_bytes);
  e.ok = wa.ok();
  return e;
}

int encoder_layer(const float* x, const float* src, const uint8_t* x_mask, const uint8_t* src_mask,
                  const loftr_layer_weights& w, float* out, int nb, int L, int S, int C, int H,
                  void* ws, size_t ws_bytes, hipStream_t st) {
  if (nb <= 0) return LOFTR_OK;
  EncoderWs e = carve(ws, ws_bytes, nb, L, S, C);
  if (!e.ok) return LOFTR_ERR_WORKSPACE;
  const int Ml = nb * L, Ms = nb * S;
  const float inv_s = 1.f / (float)S;                 // values / v_length, linear_attention.py:41-42
  int rc;
  if (x == src && x_mask == src_mask) {
    // self attention: q, k, v share the A operand -> one launch over 3C output columns
    ProjArgs p{x, Ml, C, 3, {w.q_proj, w.k_proj, w.v_proj}, {e.q, e.k, e.v}, {0, 1, 2}, x_mask, inv_s};
    if ((rc = launch_proj(p, st))) return rc;
  } else {
    ProjArgs pq{x, Ml, C, 1, {w.q_proj, nullptr, nullptr}, {e.q, nullptr, nullptr}, {0, 0, 0}, x_mask, inv_s};
    if ((rc = launch_proj(pq, st))) return rc;
    ProjArgs pkv{src, Ms, C, 2, {w.k_proj, w.v_proj, nullptr}, {e.k, e.v, nullptr}, {1, 2, 0}, src_mask, inv_s};
    if ((rc = launch_proj(pkv, st))) return rc;
  }
  if (C == 256) {
    // coarse level: KV reduction, then attention-apply + merge + norm1 as ONE GEMM   transformer.py:50-52
    const float *kv = nullptr, *pm = nullptr;
    if ((rc = launch_attention_kv(e.k, e.v, w.merge, nb, S, C, H, e.attn, e.attn_bytes, &kv, &pm, st))) return rc;
    LinearLNArgs m{asrc_plain(e.q, C), pm, C, w.norm1_w, w.norm1_b, nullptr, e.msgn, L, C, C, 1e-5f,
                   kv, nb, (float)S, 1e-6f};             // LinearAttention(eps=1e-6), linear_attention.py:15
    if ((rc = launch_linear_ln(m, st))) return rc;
  } else {
    if ((rc = launch_linear_attention(e.q, e.k, e.v, e.msg, nb, L, S, C, H, e.attn, e.attn_bytes, st))) return rc;
    // message = norm1(merge(message))                                   transformer.py:51-52
    LinearLNArgs m{asrc_plain(e.msg, C), w.merge, C, w.norm1_w, w.norm1_b, nullptr, e.msgn, Ml, C, C, 1e-5f,
                   nullptr, 0, 0.f, 0.f};
    if ((rc = launch_linear_ln(m, st))) return rc;
  }
  // hidden = relu(mlp.0(cat[x, message]))                             transformer.py:55
  LinearArgs h{asrc_cat(x, C, e.msgn, C, C), w.mlp0, 2 * C, e.hid, 2 * C, Ml, 2 * C, 2 * C, nullptr, 1};
  if ((rc = launch_linear(h, EPI_RELU, st))) return rc;
  // out = x + norm2(mlp.2(hidden))                                    transformer.py:55-58
  LinearLNArgs o{asrc_plain(e.hid, 2 * C), w.mlp2, 2 * C, w.norm2_w, w.norm2_b, x, out, Ml, C, 2 * C, 1e-5f,
                 nullptr, 0, 0.f, 0.f};
  return launch_linear_ln(o, st);
}

bool weights_ok(const loftr_layer_weights& w) {
  return w.q_proj && w.k_proj && w.v_proj && w.merge && w.mlp0 && w.mlp2 && w.norm1_w && w.norm1_b &&
         w.norm2_w && w.norm2_b;
}

}  // namespace

extern "C" size_t loftr_encoder_workspace_bytes(int nb, int L, int S, int C) {
  if (nb <= 0 || L <= 0 || S <= 0 || C <= 0) return 0;
  return encoder_ws_bytes(nb, L > S ? L : S, L > S ? L : S, C);
}

extern "C" int loftr_encoder_layer_fwd(const float* x, const float* source, const uint8_t* x_mask,
                                       const uint8_t* source_mask, const loftr_layer_weights* w,
                                       float* out, int nb, int L, int S, int C, int H, void* ws,
                                       size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(x && source && w && out && nb >= 0 && L > 0 && S > 0 && (ws || nb == 0));
  LOFTR_CHECK_ARG(weights_ok(*w));
  if (!((C == 256 || C == 128) && H == 8)) return LOFTR_ERR_UNSUPPORTED;
  return encoder_layer(x, source, x_mask, source_mask, *w, out, nb, L, S, C, H, ws, ws_bytes,
                       (hipStream_t)stream);
}

extern "C" int loftr_transformer_fwd(float* feat0, float* feat1, const uint8_t* mask0,
                                     const uint8_t* mask1, const loftr_layer_weights* layers,
                                     const int* layer_is_cross, int n_layers, int N, int L, int S,
                                     int C, int H, void* ws, size_t ws_bytes, void* stream) {
  LOFTR_CHECK_ARG(feat0 && feat1 && layers && layer_is_cross && n_layers >= 0 && N >= 0 && L > 0 && S > 0);
  LOFTR_CHECK_ARG((mask0 == nullptr) == (mask1 == nullptr));
  if (!((C == 256 || C == 128) && H == 8)) return LOFTR_ERR_UNSUPPORTED;
  if (N == 0) return LOFTR_OK;
  LOFTR_CHECK_ARG(ws != nullptr);
  hipStream_t st = (hipStream_t)stream;
  // The two self-attention calls of a layer are independent (transformer.py:92-94): when the two
  // feature sets are one contiguous [2N, L, C] buffer they run as a single batch of 2N.
  const bool stacked = (L == S) && (feat1 == feat0 + (size_t)N * L * C) &&
                       (mask0 == nullptr || mask1 == mask0 + (size_t)N * L);
  for (int i = 0; i < n_layers; ++i) {
    LOFTR_CHECK_ARG(weights_ok(layers[i]));
    int rc;
    if (!layer_is_cross[i]) {
      if (stacked) {
        if ((rc = encoder_layer(feat0, feat0, mask0, mask0, layers[i], feat0, 2 * N, L, L, C, H, ws, ws_bytes, st))) return rc;
      } else {
        if ((rc = encoder_layer(feat0, feat0, mask0, mask0, layers[i], feat0, N, L, L, C, H, ws, ws_bytes, st))) return rc;
        if ((rc = encoder_layer(feat1, feat1, mask1, mask1, layers[i], feat1, N, S, S, C, H, ws, ws_bytes, st))) return rc;
      }
    } else {
      // sequential dependency kept: feat1 attends to the UPDATED feat0 (transformer.py:96-97)
      if ((rc = encoder_layer(feat0, feat1, mask0, mask1, layers[i], feat0, N, L, S, C, H, ws, ws_bytes, st))) return rc;
      if ((rc = encoder_layer(feat1, feat0, mask1, mask0, layers[i], feat1, N, S, L, C, H, ws, ws_bytes, st))) return rc;
    }
  }
  return LOFTR_OK;
}
