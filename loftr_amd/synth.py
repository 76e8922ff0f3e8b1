"""Deterministic synthetic weights / feature maps for tests and bench (numpy-seeded).

There are no checkpoints or datasets on the box, so every parity test and the
bench run on seeded random-init weights and synthetic inputs; this module is the single
recipe for them.  The arrays are pure functions of (seed, shapes) through
``np.random.default_rng`` so the authoring container (where the golden vectors were made with
the real reference) and the GPU box regenerate identical bytes; each golden file stores
checksums to detect drift.

Initialisation follows the reference's schemes: xavier-uniform for transformer matrices
(src/loftr/loftr_module/transformer.py:75-78), kaiming-normal(fan_out) for the fine
preprocess linears (src/loftr/loftr_module/fine_preprocess.py:24-27).  LayerNorm affine
parameters and biases are randomised slightly (instead of 1/0) so the tests exercise them.
"""
import numpy as np


def _xavier(rng, out_f, in_f):
    bound = np.sqrt(6.0 / (in_f + out_f))
    return rng.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32)


def _encoder_layer(rng, prefix, d, w):
    for name in ("q_proj", "k_proj", "v_proj", "merge"):
        w[f"{prefix}{name}.weight"] = _xavier(rng, d, d)
    w[f"{prefix}mlp.0.weight"] = _xavier(rng, 2 * d, 2 * d)
    w[f"{prefix}mlp.2.weight"] = _xavier(rng, d, 2 * d)
    for n in ("norm1", "norm2"):
        w[f"{prefix}{n}.weight"] = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
        w[f"{prefix}{n}.bias"] = (0.1 * rng.standard_normal(d)).astype(np.float32)


def make_weights(seed=0, cfg=None, with_bin_score=False):
    """Hot-path weights keyed like the reference state_dict (backbone excluded)."""
    from .config import default_cfg
    cfg = cfg or default_cfg
    rng = np.random.default_rng(seed)
    w = {}
    dc, df = cfg["coarse"]["d_model"], cfg["fine"]["d_model"]
    for i in range(len(cfg["coarse"]["layer_names"])):
        _encoder_layer(rng, f"loftr_coarse.layers.{i}.", dc, w)
    for i in range(len(cfg["fine"]["layer_names"])):
        _encoder_layer(rng, f"loftr_fine.layers.{i}.", df, w)
    if cfg["fine_concat_coarse_feat"]:
        w["fine_preprocess.down_proj.weight"] = (rng.standard_normal((df, dc)) * np.sqrt(2.0 / df)).astype(np.float32)
        w["fine_preprocess.down_proj.bias"] = (0.05 * rng.standard_normal(df)).astype(np.float32)
        w["fine_preprocess.merge_feat.weight"] = (rng.standard_normal((df, 2 * df)) * np.sqrt(2.0 / df)).astype(np.float32)
        w["fine_preprocess.merge_feat.bias"] = (0.05 * rng.standard_normal(df)).astype(np.float32)
    if with_bin_score or cfg["match_coarse"]["match_type"] == "sinkhorn":
        w["coarse_matching.bin_score"] = np.float32(cfg["match_coarse"]["skh_init_bin_score"])
    return w


def make_features(seed, n, hw0_c, hw1_c, dc=256, df=128, fine_ratio=4, corr=0.0, scale_c=1.0):
    """Backbone-like outputs: feat_c0/1 [n,dc,h,w], feat_f0/1 [n,df,4h,4w] float32.

    ``corr`` in [0,1] mixes image-0 content into image-1 (shifted by one coarse cell), so that
    some pairs produce confident mutual matches.  ``scale_c`` multiplies the coarse maps: with a random-weight transformer the
    residual stream then dominates its (LayerNorm-bounded) updates and corresponding cells keep near-identical descriptors --
    the peaked score statistics of a TRAINED network (conf close to 1, |sim / temperature| in the hundreds).
    """
    rng = np.random.default_rng(seed)
    h0, w0 = hw0_c
    h1, w1 = hw1_c
    c0 = rng.standard_normal((n, dc, h0, w0)).astype(np.float32)
    c1 = rng.standard_normal((n, dc, h1, w1)).astype(np.float32)
    f0 = rng.standard_normal((n, df, h0 * fine_ratio, w0 * fine_ratio)).astype(np.float32)
    f1 = rng.standard_normal((n, df, h1 * fine_ratio, w1 * fine_ratio)).astype(np.float32)
    if corr > 0 and (h0, w0) == (h1, w1):
        a = np.float32(corr)
        b = np.float32(np.sqrt(1 - corr ** 2))
        c1 = a * np.roll(c0, (1, 1), (2, 3)) + b * c1
        f1 = a * np.roll(f0, (fine_ratio, fine_ratio), (2, 3)) + b * f1
    if scale_c != 1.0:
        c0, c1 = (np.float32(scale_c) * c0).astype(np.float32), (np.float32(scale_c) * c1).astype(np.float32)
    return c0, c1, f0, f1


def make_images(seed, n, h, w):
    """Synthetic grayscale pairs in [0,1): image1 is image0 rolled by (8,16) px + noise."""
    rng = np.random.default_rng(seed)
    img0 = rng.random((n, 1, h, w), dtype=np.float32)
    img1 = np.roll(img0, (8, 16), (2, 3)) + 0.02 * rng.random((n, 1, h, w), dtype=np.float32)
    return img0, np.clip(img1, 0, 1).astype(np.float32)


def checksum(a) -> float:
    a = np.asarray(a, dtype=np.float64).ravel()
    k = np.arange(a.size, dtype=np.float64) % 97 + 1
    return float((a * k).sum())


def make_backbone_weights(seed, module, bn_strength=1.0):
    """Seeded state_dict for a ResNet-FPN backbone (ours or the reference's: same parameter names / shapes), with
    NON-trivial BatchNorm statistics so that eval-mode BN actually scales and shifts.  Filled in state_dict order from
    one numpy Generator -> loading the result into two modules with strict=True also proves their layouts agree.
    ``bn_strength`` scales the spread of the BN affine parameters / statistics around the identity (1.0: strong
    per-channel scales and offsets, descriptors share a large common component -> few, confident matches;
    0.3: closer to a freshly initialised network -> many low-confidence matches)."""
    import torch
    rng = np.random.default_rng(seed)
    s = np.float64(bn_strength)
    out = {}
    for name, t in module.state_dict().items():
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros(shape, dtype=t.dtype)
        elif name.endswith("running_var"):
            out[name] = torch.from_numpy(((0.5 + rng.random(shape)) if bn_strength == 1.0 else (1.0 + s * (rng.random(shape) - 0.5))).astype(np.float32))
        elif name.endswith("running_mean") or (name.endswith(".bias") and t.dim() == 1):
            out[name] = torch.from_numpy((s * 0.1 * rng.standard_normal(shape)).astype(np.float32))
        elif t.dim() == 1:                                   # BN weight
            out[name] = torch.from_numpy((1.0 + s * 0.2 * rng.standard_normal(shape)).astype(np.float32))
        else:                                                # conv filters: kaiming-like scale (fan_out)
            fan_out = shape[0] * int(np.prod(shape[2:]))
            out[name] = torch.from_numpy((rng.standard_normal(shape) * np.sqrt(2.0 / fan_out)).astype(np.float32))
    return out
