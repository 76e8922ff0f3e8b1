"""Goldens of the encoder-layer backward: gradients of the reference's OWN LoFTREncoderLayer / LocalFeatureTransformer under torch.autograd.

    python tests/golden/make_golden_layer_grad.py       # authoring container only (needs /root/reference)

`glayer_*`: one `LoFTREncoderLayer.forward(x, source, x_mask, source_mask)` (src/loftr/loftr_module/transformer.py:35-58) with a random
upstream gradient G: d<G, out>/d x, d source and the ten weight gradients.  `gtf_*`: the whole `LocalFeatureTransformer.forward`
(:80-101: self layers on both maps, cross layers with feat1 attending to the UPDATED feat0), leaves = the two inputs and every weight.
Inputs and weights are regenerated from the seeds by `build` (the tests use the same function); float32 forward and backward, as the
reference trains."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "glayer_self": dict(seed=31, kind="layer", nb=2, L=48, S=48, C=256, H=8, masks=False),
    "glayer_cross_mask": dict(seed=32, kind="layer", nb=2, L=40, S=56, C=256, H=8, masks=True),
    "glayer_fine": dict(seed=33, kind="layer", nb=5, L=25, S=25, C=128, H=8, masks=False),
    # FinePreprocess (fine_preprocess.py:29-59): border cells (clipped windows), a cell matched twice, unequal maps
    "gfpre": dict(seed=35, kind="fpre", N=2, hc0=(6, 8), hc1=(5, 9), Cf=128, Cc=256, W=5, stride=4, M=40),
    "gtf_coarse": dict(seed=34, kind="tf", N=2, L=48, S=35, C=256, H=8, masks=True, layers=["self", "cross", "self", "cross"]),
}
FIELDS = (("q_proj", "q_proj.weight"), ("k_proj", "k_proj.weight"), ("v_proj", "v_proj.weight"), ("merge", "merge.weight"),
          ("mlp0", "mlp.0.weight"), ("mlp2", "mlp.2.weight"), ("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"),
          ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"))


def digest(name, g):
    """What the npz keeps of a weight-gradient MATRIX (the fixtures stay small): a strided sub-matrix plus all row and column sums
    (every entry enters two of them); vectors are stored whole."""
    g = np.asarray(g)
    if g.ndim < 2:
        return {name: g}
    sr, sc = max(1, g.shape[0] // 16), max(1, g.shape[1] // 32)
    return {f"{name}/sub": g[::sr, ::sc].copy(), f"{name}/rowsum": g.astype(np.float64).sum(1), f"{name}/colsum": g.astype(np.float64).sum(0),
            f"{name}/absmax": np.float64(np.abs(g).max())}


def layer_weights(rng, C):
    """One layer's state_dict (reference names), seeded: xavier-sized matrices, LayerNorm scales around 1."""
    xav = lambda o, i: (rng.uniform(-1, 1, (o, i)) * np.sqrt(6.0 / (o + i))).astype(np.float32)
    return {"q_proj.weight": xav(C, C), "k_proj.weight": xav(C, C), "v_proj.weight": xav(C, C), "merge.weight": xav(C, C),
            "mlp.0.weight": xav(2 * C, 2 * C), "mlp.2.weight": xav(C, 2 * C),
            "norm1.weight": (1 + 0.2 * rng.standard_normal(C)).astype(np.float32), "norm1.bias": (0.1 * rng.standard_normal(C)).astype(np.float32),
            "norm2.weight": (1 + 0.2 * rng.standard_normal(C)).astype(np.float32), "norm2.bias": (0.1 * rng.standard_normal(C)).astype(np.float32)}


def build(rc):
    rng = np.random.default_rng(rc["seed"])
    if rc["kind"] == "fpre":
        N, (h0, w0), (h1, w1), Cf, Cc, st, M = rc["N"], rc["hc0"], rc["hc1"], rc["Cf"], rc["Cc"], rc["stride"], rc["M"]
        b = np.sort(rng.integers(0, N, M)).astype(np.int64)
        i = rng.integers(0, h0 * w0, M).astype(np.int64)
        j = rng.integers(0, h1 * w1, M).astype(np.int64)
        i[:4] = [0, w0 - 1, (h0 - 1) * w0, h0 * w0 - 1]                      # the four corner cells: windows clipped on two sides
        b[:4] = 0
        i[5], b[5], j[5] = i[4], b[4], (j[4] + 1) % (h1 * w1)                # cell (b, i) carries two matches
        kai = lambda o, ii: (rng.standard_normal((o, ii)) * np.sqrt(2.0 / o)).astype(np.float32)
        return dict(feat_f0=rng.standard_normal((N, Cf, h0 * st, w0 * st)).astype(np.float32),
                    feat_f1=rng.standard_normal((N, Cf, h1 * st, w1 * st)).astype(np.float32),
                    feat_c0=rng.standard_normal((N, h0 * w0, Cc)).astype(np.float32), feat_c1=rng.standard_normal((N, h1 * w1, Cc)).astype(np.float32),
                    b_ids=b, i_ids=i, j_ids=j, G0=rng.standard_normal((M, rc["W"] ** 2, Cf)).astype(np.float32),
                    G1=rng.standard_normal((M, rc["W"] ** 2, Cf)).astype(np.float32),
                    w={"down_proj.weight": kai(Cf, Cc), "down_proj.bias": (0.1 * rng.standard_normal(Cf)).astype(np.float32),
                       "merge_feat.weight": kai(Cf, 2 * Cf), "merge_feat.bias": (0.1 * rng.standard_normal(Cf)).astype(np.float32)})
    C = rc["C"]
    if rc["kind"] == "layer":
        nb, L, S = rc["nb"], rc["L"], rc["S"]
        out = dict(x=rng.standard_normal((nb, L, C)).astype(np.float32), source=rng.standard_normal((nb, S, C)).astype(np.float32),
                   G=rng.standard_normal((nb, L, C)).astype(np.float32), w=layer_weights(rng, C), x_mask=None, source_mask=None)
        if rc["masks"]:
            xm, sm = np.ones((nb, L), bool), np.ones((nb, S), bool)
            xm[0, L - 7:], sm[0, S - 11:], sm[1, S - 3:] = False, False, False
            out.update(x_mask=xm, source_mask=sm)
        return out
    N, L, S = rc["N"], rc["L"], rc["S"]
    out = dict(feat0=rng.standard_normal((N, L, C)).astype(np.float32), feat1=rng.standard_normal((N, S, C)).astype(np.float32),
               G0=rng.standard_normal((N, L, C)).astype(np.float32), G1=rng.standard_normal((N, S, C)).astype(np.float32),
               w=[layer_weights(rng, C) for _ in rc["layers"]], mask0=None, mask1=None)
    if rc["masks"]:
        m0, m1 = np.ones((N, L), bool), np.ones((N, S), bool)
        m0[0, L - 9:], m1[1, S - 6:] = False, False
        out.update(mask0=m0, mask1=m1)
    return out


def make(name):
    import importlib
    import torch
    from oracle.ref_shim import import_reference
    import_reference()
    tfm = importlib.import_module("src.loftr.loftr_module.transformer")
    rc = CASES[name]
    inp = build(rc)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
    store = dict(recipe=np.array(json.dumps(rc)))
    if rc["kind"] == "fpre":
        fpm = importlib.import_module("src.loftr.loftr_module.fine_preprocess")
        cfg = {"fine_concat_coarse_feat": True, "fine_window_size": rc["W"], "coarse": {"d_model": rc["Cc"]}, "fine": {"d_model": rc["Cf"]}}
        fp = fpm.FinePreprocess(cfg)
        fp.load_state_dict({k: t(v) for k, v in inp["w"].items()}, strict=True)
        leaf = {k: t(inp[k]).requires_grad_(True) for k in ("feat_f0", "feat_f1", "feat_c0", "feat_c1")}
        (h0, w0), (h1, w1), stp = rc["hc0"], rc["hc1"], rc["stride"]
        data = {"hw0_f": (h0 * stp, w0 * stp), "hw0_c": (h0, w0), "b_ids": t(inp["b_ids"]), "i_ids": t(inp["i_ids"]), "j_ids": t(inp["j_ids"])}
        o0, o1 = fp(leaf["feat_f0"], leaf["feat_f1"], leaf["feat_c0"], leaf["feat_c1"], data)
        ((o0 * t(inp["G0"])).sum() + (o1 * t(inp["G1"])).sum()).backward()
        store.update(out0=o0.detach().numpy(), out1=o1.detach().numpy(), **{f"grad_{k}": v.grad.numpy() for k, v in leaf.items()})
        for n, prm in fp.named_parameters():
            store.update(digest("grad_" + n.replace(".", "_"), prm.grad.numpy()))
    elif rc["kind"] == "layer":
        layer = tfm.LoFTREncoderLayer(rc["C"], rc["H"], "linear")
        layer.load_state_dict({k: t(v) for k, v in inp["w"].items()}, strict=True)
        x, s = t(inp["x"]).requires_grad_(True), t(inp["source"]).requires_grad_(True)
        out = layer(x, s, t(inp["x_mask"]), t(inp["source_mask"]))
        (out * t(inp["G"])).sum().backward()
        store.update(out=out.detach().numpy(), grad_x=x.grad.numpy(), grad_source=s.grad.numpy())
        for f, n in FIELDS:
            store.update(digest(f"grad_{f}", dict(layer.named_parameters())[n].grad.numpy()))
    else:
        cfg = dict(d_model=rc["C"], nhead=rc["H"], layer_names=rc["layers"], attention="linear")
        tf = tfm.LocalFeatureTransformer(cfg)
        tf.load_state_dict({f"layers.{i}.{k}": t(v) for i, w in enumerate(inp["w"]) for k, v in w.items()}, strict=True)
        f0, f1 = t(inp["feat0"]).requires_grad_(True), t(inp["feat1"]).requires_grad_(True)
        o0, o1 = tf(f0, f1, t(inp["mask0"]), t(inp["mask1"]))
        ((o0 * t(inp["G0"])).sum() + (o1 * t(inp["G1"])).sum()).backward()
        store.update(out0=o0.detach().numpy(), out1=o1.detach().numpy(), grad_feat0=f0.grad.numpy(), grad_feat1=f1.grad.numpy())
        for i, layer in enumerate(tf.layers):
            for f, n in FIELDS:
                store.update(digest(f"grad_l{i}_{f}", dict(layer.named_parameters())[n].grad.numpy()))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **store)
    print(name, {k: float(np.abs(v).max()) for k, v in store.items() if k.startswith("grad_") and "/" not in k},
          "%.1f kB" % (os.path.getsize(os.path.join(HERE, f"{name}.npz")) / 1e3))


if __name__ == "__main__":
    for nm in sys.argv[1:] or list(CASES):
        make(nm)
