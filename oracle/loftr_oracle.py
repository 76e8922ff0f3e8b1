"""CPU oracle for the LoFTR matching hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of what zju3dv/LoFTR computes in
``LoFTR.forward`` *after* the ResNet-FPN backbone (reference ``src/loftr/loftr.py:56-75``).
It is the checker for the HIP path; nothing in ``loftr_amd/`` (the product) may import it.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it.

Pinning status: the reference ships no golden vectors or unit tests, so this
oracle is pinned against *outputs of the reference itself*, produced in the authoring
container by ``tests/golden/make_golden.py`` (reference imported through ``oracle/ref_shim.py``)
and committed as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` re-checks that pin.

Every function cites the reference lines it follows (paths relative to the reference root).
All arithmetic is float32 unless ``dtype=np.float64`` is requested (used to measure how much
fp32 re-association can move an output, i.e. to justify the test tolerances).

Weights are a flat ``dict`` keyed exactly like the reference ``state_dict``
(``loftr_coarse.layers.0.q_proj.weight`` ...), values numpy arrays.
"""
from __future__ import annotations

import math
import numpy as np

INF = 1e9  # src/loftr/utils/coarse_matching.py:6


# ----------------------------------------------------------------------------------------
# position encoding + flatten                     src/loftr/utils/position_encoding.py:22-42
# ----------------------------------------------------------------------------------------
def position_encoding_table(d_model: int, h: int, w: int, temp_bug_fix: bool = True,
                            dtype=np.float32) -> np.ndarray:
    """pe[c, y, x] for y<h, x<w.  position_encoding.py:22-33.

    Positions are 1-based (cumsum of ones, :23-24).  With ``temp_bug_fix=False`` the
    reference evaluates ``-math.log(10000.0) / d_model // 2`` == floor(-0.0359..) == -1.0 (:28).
    The table is built in float32 like the reference (torch.exp/sin/cos on float32 tensors).
    """
    y_pos = np.arange(1, h + 1, dtype=np.float32)[:, None] * np.ones((1, w), np.float32)
    x_pos = np.ones((h, 1), np.float32) * np.arange(1, w + 1, dtype=np.float32)[None, :]
    k = np.arange(0, d_model // 2, 2, dtype=np.float32)
    if temp_bug_fix:
        factor = np.float32(-math.log(10000.0) / (d_model // 2))
    else:
        factor = np.float32(-math.log(10000.0) / d_model // 2)
    div = np.exp(k * factor).astype(np.float32)[:, None, None]
    pe = np.zeros((d_model, h, w), np.float32)
    pe[0::4] = np.sin(x_pos[None] * div)
    pe[1::4] = np.cos(x_pos[None] * div)
    pe[2::4] = np.sin(y_pos[None] * div)
    pe[3::4] = np.cos(y_pos[None] * div)
    return pe.astype(dtype)


def add_pos_flatten(feat_nchw: np.ndarray, temp_bug_fix: bool = True) -> np.ndarray:
    """x + pe, then 'n c h w -> n (h w) c'.  loftr.py:58-59, position_encoding.py:42."""
    n, c, h, w = feat_nchw.shape
    pe = position_encoding_table(c, h, w, temp_bug_fix, feat_nchw.dtype)
    x = feat_nchw + pe[None]
    return np.ascontiguousarray(x.reshape(n, c, h * w).transpose(0, 2, 1))


# ----------------------------------------------------------------------------------------
# linear attention                         src/loftr/loftr_module/linear_attention.py:10-47
# ----------------------------------------------------------------------------------------
def elu_feature_map(x: np.ndarray) -> np.ndarray:
    """elu(x)+1.  linear_attention.py:10-11."""
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(x.dtype) + x.dtype.type(1)


def linear_attention(q, k, v, q_mask=None, kv_mask=None, eps=1e-6):
    """q [N,L,H,D], k,v [N,S,H,D] -> [N,L,H,D].  linear_attention.py:20-47."""
    dt = q.dtype
    Q = elu_feature_map(q)
    K = elu_feature_map(k)
    if q_mask is not None:                                   # :35-36
        Q = Q * q_mask[:, :, None, None].astype(dt)
    if kv_mask is not None:                                  # :37-39
        K = K * kv_mask[:, :, None, None].astype(dt)
        v = v * kv_mask[:, :, None, None].astype(dt)
    S = v.shape[1]
    v = v / dt.type(S)                                       # :41-42
    # einsums written as batched matmuls (BLAS) -- same contractions, much faster in numpy
    Kt = K.transpose(0, 2, 3, 1)                             # [N,H,D,S]
    Qh = Q.transpose(0, 2, 1, 3)                             # [N,H,L,D]
    KV = Kt @ v.transpose(0, 2, 1, 3)                        # :43  "nshd,nshv->nhdv"
    Ksum = K.sum(axis=1)                                     # [N,H,D]
    Z = dt.type(1) / ((Qh @ Ksum[..., None])[..., 0] + dt.type(eps))    # :44  [N,H,L]
    out = (Qh @ KV) * Z[..., None] * dt.type(S)              # :45  [N,H,L,V]
    return np.ascontiguousarray(out.transpose(0, 2, 1, 3).astype(dt))


def layer_norm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm over the last dim (biased variance).  transformer.py:32-33,52,56."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + x.dtype.type(eps)) * weight + bias).astype(x.dtype)


# ----------------------------------------------------------------------------------------
# encoder layer / transformer                  src/loftr/loftr_module/transformer.py:35-101
# ----------------------------------------------------------------------------------------
def encoder_layer(x, source, w: dict, prefix: str, nhead: int, x_mask=None, source_mask=None):
    """LoFTREncoderLayer.forward.  transformer.py:35-58.  x [N,L,C], source [N,S,C]."""
    dt = x.dtype
    g = lambda name: w[prefix + name].astype(dt)
    n, L, C = x.shape
    S = source.shape[1]
    dim = C // nhead
    q = (x @ g("q_proj.weight").T).reshape(n, L, nhead, dim)            # :47
    k = (source @ g("k_proj.weight").T).reshape(n, S, nhead, dim)       # :48
    v = (source @ g("v_proj.weight").T).reshape(n, S, nhead, dim)       # :49
    msg = linear_attention(q, k, v, x_mask, source_mask)                # :50
    msg = msg.reshape(n, L, C) @ g("merge.weight").T                    # :51
    msg = layer_norm(msg, g("norm1.weight"), g("norm1.bias"))           # :52
    hid = np.concatenate([x, msg], axis=2) @ g("mlp.0.weight").T        # :55 (Linear 2C->2C)
    hid = np.maximum(hid, 0)                                            # ReLU
    msg = hid @ g("mlp.2.weight").T                                     # Linear 2C->C
    msg = layer_norm(msg, g("norm2.weight"), g("norm2.bias"))           # :56
    return (x + msg).astype(dt)                                         # :58


def local_feature_transformer(feat0, feat1, w: dict, prefix: str, layer_names, nhead: int,
                              mask0=None, mask1=None):
    """LocalFeatureTransformer.forward.  transformer.py:80-101.

    NB the cross layer is sequential: feat1 attends to the *updated* feat0 (:96-97).
    """
    for idx, name in enumerate(layer_names):
        p = f"{prefix}layers.{idx}."
        if name == "self":
            feat0 = encoder_layer(feat0, feat0, w, p, nhead, mask0, mask0)
            feat1 = encoder_layer(feat1, feat1, w, p, nhead, mask1, mask1)
        elif name == "cross":
            feat0 = encoder_layer(feat0, feat1, w, p, nhead, mask0, mask1)
            feat1 = encoder_layer(feat1, feat0, w, p, nhead, mask1, mask0)
        else:
            raise KeyError(name)                                        # :98-99
    return feat0, feat1


# ----------------------------------------------------------------------------------------
# coarse matching                               src/loftr/utils/coarse_matching.py:87-261
# ----------------------------------------------------------------------------------------
def _softmax(x, axis):
    # in-place style: large temporaries are very slow to fault in on the test VMs
    m = x.max(axis=axis, keepdims=True)
    e = np.subtract(x, m)
    np.exp(e, out=e)
    e /= e.sum(axis=axis, keepdims=True)
    return e


def dual_softmax_conf(feat_c0, feat_c1, temperature=0.1, mask_c0=None, mask_c1=None):
    """conf_matrix [N,L,S].  coarse_matching.py:105-119."""
    dt = feat_c0.dtype
    C = feat_c0.shape[-1]
    f0 = feat_c0 / dt.type(C ** .5)                                     # :108-110
    f1 = feat_c1 / dt.type(C ** .5)
    sim = (f0 @ f1.transpose(0, 2, 1)) / dt.type(temperature)           # :113-114 "nlc,nsc->nls"
    if mask_c0 is not None:                                             # :115-118
        valid = mask_c0[:, :, None].astype(bool) & mask_c1[:, None, :].astype(bool)
        sim = np.where(valid, sim, dt.type(-INF))
    conf = _softmax(sim, 1)
    conf *= _softmax(sim, 2)                                            # :119
    return conf


def _logsumexp(x, axis):
    m = x.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def log_optimal_transport(scores, alpha, iters: int):
    """SuperGlue log-domain Sinkhorn with dustbins.

    Third-party: magicleap/SuperGluePretrainedNetwork ``models/superglue.py`` @ master
    (un-vendored, README.md:68-73 of the reference).  Restated from its published algorithm;
    call site coarse_matching.py:130-131.  scores [b,m,n] -> [b,m+1,n+1].
    """
    dt = scores.dtype
    b, m, n = scores.shape
    alpha = dt.type(alpha)
    Z = np.empty((b, m + 1, n + 1), dt)
    Z[:, :m, :n] = scores
    Z[:, :m, n] = alpha
    Z[:, m, :] = alpha
    norm = dt.type(-math.log(m + n))
    log_mu = np.concatenate([np.full(m, norm, dt), np.array([math.log(n) + norm], dt)])[None].repeat(b, 0)
    log_nu = np.concatenate([np.full(n, norm, dt), np.array([math.log(m) + norm], dt)])[None].repeat(b, 0)
    u = np.zeros_like(log_mu)
    v = np.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - _logsumexp(Z + v[:, None, :], axis=2)
        v = log_nu - _logsumexp(Z + u[:, :, None], axis=1)
    return (Z + u[:, :, None] + v[:, None, :] - norm).astype(dt)


def sinkhorn_conf(feat_c0, feat_c1, bin_score, iters=3, mask_c0=None, mask_c1=None,
                  prefilter=False):
    """(conf_matrix [N,L,S], assign_matrix [N,L+1,S+1]).  coarse_matching.py:121-143 (eval)."""
    dt = feat_c0.dtype
    C = feat_c0.shape[-1]
    L, S = feat_c0.shape[1], feat_c1.shape[1]
    f0 = feat_c0 / dt.type(C ** .5)
    f1 = feat_c1 / dt.type(C ** .5)
    sim = f0 @ f1.transpose(0, 2, 1)                                    # :123 (no temperature)
    if mask_c0 is not None:                                             # :124-127
        valid = mask_c0[:, :, None].astype(bool) & mask_c1[:, None, :].astype(bool)
        sim = np.where(valid, sim, dt.type(-INF))
    assign = np.exp(log_optimal_transport(sim, bin_score, iters))       # :130-132
    conf = assign[:, :-1, :-1]                                          # :133 -- a VIEW: the prefilter below zeroes assign too (:143 clones it later)
    if prefilter:                                                       # :136-140 (eval only)
        filter0 = (assign.argmax(axis=2) == S)[:, :-1]
        filter1 = (assign.argmax(axis=1) == L)[:, :-1]
        conf[np.broadcast_to(filter0[:, :, None], conf.shape)] = 0
        conf[np.broadcast_to(filter1[:, None, :], conf.shape)] = 0
    return conf.astype(dt).copy(), assign.astype(dt)


def coarse_match_select(conf, thr, border_rm, hw0_c, hw1_c, hw0_i, mask0=None, mask1=None,
                        scale0=None, scale1=None):
    """Eval branch of CoarseMatching.get_coarse_match.  coarse_matching.py:150-196,238-261.

    mask0/mask1: [N,h_c,w_c] bool (MegaDepth padding masks) or None.
    scale0/scale1: [N,2] or None.  Returns dict with the reference's keys.
    """
    N = conf.shape[0]
    h0, w0 = hw0_c
    h1, w1 = hw1_c
    mask = (conf > conf.dtype.type(thr)).reshape(N, h0, w0, h1, w1).copy()      # :172-175
    b = int(border_rm)
    if b > 0:
        # mask_border :8-25 / mask_border_with_padding :28-43 (near borders)
        mask[:, :b] = False
        mask[:, :, :b] = False
        mask[:, :, :, :b] = False
        mask[:, :, :, :, :b] = False
        if mask0 is None:
            mask[:, -b:] = False
            mask[:, :, -b:] = False
            mask[:, :, :, -b:] = False
            mask[:, :, :, :, -b:] = False
        else:
            h0s = mask0.sum(1).max(-1).astype(int); w0s = mask0.sum(-1).max(-1).astype(int)  # :37
            h1s = mask1.sum(1).max(-1).astype(int); w1s = mask1.sum(-1).max(-1).astype(int)  # :38
            for bi in range(N):                                                  # :39-43
                mask[bi, h0s[bi] - b:] = False
                mask[bi, :, w0s[bi] - b:] = False
                mask[bi, :, :, h1s[bi] - b:] = False
                mask[bi, :, :, :, w1s[bi] - b:] = False
    mask = mask.reshape(N, h0 * w0, h1 * w1)
    mask = mask & (conf == conf.max(axis=2, keepdims=True)) \
                & (conf == conf.max(axis=1, keepdims=True))                      # :187-189
    mask_v = mask.max(axis=2)                                                    # :193
    all_j = mask.argmax(axis=2)                 # first True index, like torch bool max
    b_ids, i_ids = np.nonzero(mask_v)                                            # :194
    j_ids = all_j[b_ids, i_ids]
    mconf = conf[b_ids, i_ids, j_ids]
    scale = hw0_i[0] / hw0_c[0]                                                  # :242
    s0 = scale * scale0[b_ids] if scale0 is not None else scale                  # :243
    s1 = scale * scale1[b_ids] if scale1 is not None else scale                  # :244
    mk0 = (np.stack([i_ids % w0, i_ids // w0], 1) * s0).astype(np.float32)       # :245-247
    mk1 = (np.stack([j_ids % w1, j_ids // w1], 1) * s1).astype(np.float32)       # :248-250
    keep = mconf != 0                                                            # :254-258
    return dict(b_ids=b_ids.astype(np.int64), i_ids=i_ids.astype(np.int64),
                j_ids=j_ids.astype(np.int64), gt_mask=(mconf == 0),
                m_bids=b_ids[keep].astype(np.int64), mkpts0_c=mk0[keep], mkpts1_c=mk1[keep],
                mconf=mconf[keep].astype(np.float32))


# ----------------------------------------------------------------------------------------
# fine preprocess                          src/loftr/loftr_module/fine_preprocess.py:29-59
# ----------------------------------------------------------------------------------------
def gather_windows(feat_f, b_ids, c_ids, w_c: int, W: int, stride: int):
    """Rows of F.unfold(k=W, stride, pad=W//2) -> 'n l ww c', picked at (b_ids, c_ids).

    fine_preprocess.py:40-47.  feat_f [N,C,Hf,Wf]; coarse cell c=(cy*w_c+cx) has its window
    centred on fine pixel (stride*cy, stride*cx); zero outside the map.  Returns [M,WW,C].
    """
    N, C, Hf, Wf = feat_f.shape
    M = len(b_ids)
    r = W // 2
    out = np.zeros((M, W * W, C), feat_f.dtype)
    cy = (c_ids // w_c) * stride
    cx = (c_ids % w_c) * stride
    for wy in range(W):
        for wx in range(W):
            y = cy + wy - r
            x = cx + wx - r
            ok = (y >= 0) & (y < Hf) & (x >= 0) & (x < Wf)
            yy = np.clip(y, 0, Hf - 1)
            xx = np.clip(x, 0, Wf - 1)
            vals = feat_f[b_ids, :, yy, xx]                     # [M,C]
            out[:, wy * W + wx, :] = np.where(ok[:, None], vals, 0)
    return out


def fine_preprocess(feat_f0, feat_f1, feat_c0, feat_c1, b_ids, i_ids, j_ids, w: dict,
                    hw0_c, hw1_c, W=5, cat_c_feat=True):
    """FinePreprocess.forward.  fine_preprocess.py:29-59.  Returns two [M,WW,Cf]."""
    dt = feat_f0.dtype
    Cf = feat_f0.shape[1]
    stride = feat_f0.shape[2] // hw0_c[0]                                        # :31
    M = len(b_ids)
    if M == 0:                                                                   # :34-37
        return np.zeros((0, W * W, Cf), dt), np.zeros((0, W * W, Cf), dt)
    u0 = gather_windows(feat_f0, b_ids, i_ids, hw0_c[1], W, stride)              # :40-46
    u1 = gather_windows(feat_f1, b_ids, j_ids, hw1_c[1], W, stride)              # :42-47
    if cat_c_feat:
        cc = np.concatenate([feat_c0[b_ids, i_ids], feat_c1[b_ids, j_ids]], 0)   # :51-52
        c_win = cc @ w["fine_preprocess.down_proj.weight"].astype(dt).T \
            + w["fine_preprocess.down_proj.bias"].astype(dt)
        cat = np.concatenate([np.concatenate([u0, u1], 0),
                              np.repeat(c_win[:, None, :], W * W, 1)], -1)       # :53-56
        merged = cat @ w["fine_preprocess.merge_feat.weight"].astype(dt).T \
            + w["fine_preprocess.merge_feat.bias"].astype(dt)
        u0, u1 = merged[:M], merged[M:]                                          # :57
    return u0.astype(dt), u1.astype(dt)


# ----------------------------------------------------------------------------------------
# fine matching                                   src/loftr/utils/fine_matching.py:15-74
# ----------------------------------------------------------------------------------------
def fine_matching(feat_f0, feat_f1, mkpts0_c, mkpts1_c, b_ids, hw0_i, hw0_f,
                  scale1=None, has_scale0=False):
    """FineMatching.forward + get_fine_match.  fine_matching.py:15-74.

    kornia 0.4.1 semantics restated: create_meshgrid(W,W,True)[y,x] =
    (2x/(W-1)-1, 2y/(W-1)-1); spatial_expectation2d = (sum p*gx, sum p*gy).
    Returns (expec_f [M,3], mkpts0_f [M,2], mkpts1_f [M,2]).
    """
    M, WW, C = feat_f0.shape
    dt = feat_f0.dtype
    W = int(math.sqrt(WW))
    scale = hw0_i[0] / hw0_f[0]                                                  # :30
    if M == 0:                                                                   # :33-41
        return np.zeros((0, 3), np.float32), mkpts0_c, mkpts1_c
    picked = feat_f0[:, WW // 2, :]                                              # :43
    sim = (feat_f1 @ picked[:, :, None])[:, :, 0]                                # :44 "mc,mrc->mr"
    heat = _softmax(dt.type(1. / C ** .5) * sim, 1)                              # :45-46
    lin = (np.arange(W, dtype=dt) * dt.type(2) / dt.type(W - 1) - dt.type(1))
    gx = np.tile(lin, W)            # x fastest
    gy = np.repeat(lin, W)
    grid = np.stack([gx, gy], -1)                                                # [WW,2]  :50
    coords = heat @ grid                                                         # :49
    var = (heat[:, :, None] * grid[None] ** 2).sum(1) - coords ** 2              # :53
    std = np.sqrt(np.maximum(var, dt.type(1e-10))).sum(-1)                       # :54
    expec = np.concatenate([coords, std[:, None]], -1).astype(np.float32)        # :57
    s1 = scale * scale1[b_ids] if has_scale0 else scale                          # :68 (sic: keyed on scale0)
    mk1 = mkpts1_c + (coords * (W // 2) * s1)[:len(mkpts1_c)]                    # :69
    return expec, mkpts0_c, mk1.astype(np.float32)


# ----------------------------------------------------------------------------------------
# the whole hot path                                        src/loftr/loftr.py:56-75
# ----------------------------------------------------------------------------------------
def loftr_hot_path(feat_c0, feat_c1, feat_f0, feat_f1, w: dict, cfg: dict, hw0_i, hw1_i,
                   mask0=None, mask1=None, scale0=None, scale1=None, dtype=np.float32,
                   keep_intermediates=False):
    """Everything in LoFTR.forward after the backbone.  loftr.py:56-75.

    feat_c* [N,C,h_c,w_c], feat_f* [N,Cf,h_f,w_f] (backbone outputs, NCHW).
    cfg: the reference's lower-cased config dict (cvpr_ds_config.py:10-50 keys).
    mask0/mask1 [N,h_c,w_c] bool.  Returns a dict with the reference's batch-dict keys.
    """
    cast = lambda a: None if a is None else np.asarray(a).astype(dtype)
    feat_c0, feat_c1, feat_f0, feat_f1 = map(cast, (feat_c0, feat_c1, feat_f0, feat_f1))
    w = {k: np.asarray(v).astype(dtype) for k, v in w.items()}
    out = {}
    hw0_c, hw1_c = feat_c0.shape[2:], feat_c1.shape[2:]
    hw0_f, hw1_f = feat_f0.shape[2:], feat_f1.shape[2:]
    out.update(bs=feat_c0.shape[0], hw0_i=tuple(hw0_i), hw1_i=tuple(hw1_i), hw0_c=tuple(hw0_c),
               hw1_c=tuple(hw1_c), hw0_f=tuple(hw0_f), hw1_f=tuple(hw1_f))
    tbf = cfg["coarse"]["temp_bug_fix"]
    fc0 = add_pos_flatten(feat_c0, tbf)                                          # :58
    fc1 = add_pos_flatten(feat_c1, tbf)                                          # :59
    m0 = m1 = None
    if mask0 is not None:                                                        # :61-63
        m0 = np.asarray(mask0).reshape(mask0.shape[0], -1).astype(bool)
        m1 = np.asarray(mask1).reshape(mask1.shape[0], -1).astype(bool)
    fc0, fc1 = local_feature_transformer(fc0, fc1, w, "loftr_coarse.", cfg["coarse"]["layer_names"],
                                         cfg["coarse"]["nhead"], m0, m1)         # :64
    mc = cfg["match_coarse"]
    if mc["match_type"] == "dual_softmax":                                       # :67
        conf = dual_softmax_conf(fc0, fc1, mc["dsmax_temperature"], m0, m1)
    elif mc["match_type"] == "sinkhorn":
        conf, assign = sinkhorn_conf(fc0, fc1, w["coarse_matching.bin_score"], mc["skh_iters"],
                                     m0, m1, prefilter=mc["skh_prefilter"])
        if mc.get("sparse_spvs", False):
            out["conf_matrix_with_bin"] = assign
    else:
        raise NotImplementedError(mc["match_type"])
    out["conf_matrix"] = conf
    sel = coarse_match_select(conf, mc["thr"], mc["border_rm"], hw0_c, hw1_c, hw0_i,
                              None if mask0 is None else np.asarray(mask0).astype(bool),
                              None if mask1 is None else np.asarray(mask1).astype(bool),
                              scale0, scale1)
    out.update(sel)
    W = cfg["fine_window_size"]
    out["W"] = W
    u0, u1 = fine_preprocess(feat_f0, feat_f1, fc0, fc1, sel["b_ids"], sel["i_ids"], sel["j_ids"],
                             w, hw0_c, hw1_c, W, cfg["fine_concat_coarse_feat"])  # :70
    if keep_intermediates:
        out["feat_c0"], out["feat_c1"] = fc0, fc1
        out["feat_f0_unfold_pre"], out["feat_f1_unfold_pre"] = u0, u1
    if u0.shape[0] != 0:                                                         # :71-72
        u0, u1 = local_feature_transformer(u0, u1, w, "loftr_fine.", cfg["fine"]["layer_names"],
                                           cfg["fine"]["nhead"])
    if keep_intermediates:
        out["feat_f0_unfold"], out["feat_f1_unfold"] = u0, u1
    expec, mk0f, mk1f = fine_matching(u0, u1, sel["mkpts0_c"], sel["mkpts1_c"], sel["b_ids"],
                                      hw0_i, hw0_f, scale1, has_scale0=scale0 is not None)  # :75
    out.update(expec_f=expec, mkpts0_f=mk0f, mkpts1_f=mk1f)
    return out
