"""Shared helpers for the parity tests: golden loading, input regeneration, comparisons.

Tolerances (BASELINE.json north_star): key points within 1e-3 px, confidences within 1e-4.
"""
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(HERE, "golden")

TOL_CONF = 1e-4      # |mconf - ref|            (north_star)
TOL_PX = 1e-3        # |mkpts*_f - ref| in px   (north_star)

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN_DIR, "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

SMALL_CASES = ["small_ds", "small_ds_corr", "small_mask", "small_ot", "small_ot_prefilter",
               "small_ot_mask", "small_empty",
               # round 3: peaked ("trained-like") statistics; Sinkhorn prefilter with surviving matches
               "peaked_small_ds", "peaked_ot_prefilter", "peaked_ot_mask_prefilter"]
MID_CASES = ["mid_ds"]
FULL_CASES = ["full_ds_thr0", "full_ds_thr02", "full_ot", "outdoor_mask", "peaked_ds"]


def load_case(name, check=True):
    """(recipe, inputs, golden dict) for a golden case; regenerates inputs from the seeds and
    verifies their checksums against the ones stored when the reference was run."""
    g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))
    rc = json.loads(str(g["recipe"]))
    inp = make_golden.build_case_inputs(rc)
    if check:
        want = json.loads(str(g["checksums"]))
        got = make_golden.input_checksums(inp)
        for k in want:
            assert np.isclose(got[k], want[k], rtol=1e-9, atol=1e-6), \
                f"synthetic input '{k}' of case {name} drifted: {got[k]} vs {want[k]}"
    return rc, inp, g


def match_keys(d):
    return list(zip(np.asarray(d["b_ids"]).tolist(), np.asarray(d["i_ids"]).tolist(),
                    np.asarray(d["j_ids"]).tolist()))


def compare_to_golden(out, g, thr, tol_conf=TOL_CONF, tol_px=TOL_PX, max_flips=0):
    """Compare a hot-path result dict with a golden (reference) dict.

    Matches are keyed by (b,i,j).  Common matches must agree within the tolerances.  A match
    present on one side only is a *flip*; it is accepted only if it is borderline -- its
    confidence lies within 10*tol_conf of the threshold, or (mutual-NN near tie) within
    10*tol_conf of the row/col maximum on the side where it is absent -- and the number of
    flips is <= max_flips.  Returns a report dict.
    """
    ko, kg = match_keys(out), match_keys(g)
    so, sg = {k: n for n, k in enumerate(ko)}, {k: n for n, k in enumerate(kg)}
    common = [k for k in ko if k in sg]
    io = np.array([so[k] for k in common], int)
    ig = np.array([sg[k] for k in common], int)
    rep = dict(M_out=len(ko), M_ref=len(kg), common=len(common),
               only_out=[k for k in ko if k not in sg], only_ref=[k for k in kg if k not in so])
    # order: ascending (b,i) on both sides
    assert ko == sorted(ko), "matches not in ascending (b,i) order"
    if len(common):
        d = lambda key: np.abs(np.asarray(out[key], np.float64)[io] - np.asarray(g[key], np.float64)[ig]).max()
        rep.update(d_mconf=d("mconf"), d_mkpts0_c=d("mkpts0_c"), d_mkpts1_c=d("mkpts1_c"),
                   d_mkpts0_f=d("mkpts0_f"), d_mkpts1_f=d("mkpts1_f"), d_expec_xy=np.abs(
                       np.asarray(out["expec_f"], np.float64)[io, :2] - np.asarray(g["expec_f"], np.float64)[ig, :2]).max())
        assert rep["d_mconf"] <= tol_conf, rep
        assert rep["d_mkpts0_f"] <= tol_px and rep["d_mkpts1_f"] <= tol_px, rep
        assert rep["d_mkpts0_c"] <= tol_px and rep["d_mkpts1_c"] <= tol_px, rep
    flips = len(rep["only_out"]) + len(rep["only_ref"])
    assert flips <= max_flips, f"{flips} match flips (allowed {max_flips}): {rep}"
    for k in rep["only_out"]:
        c = float(np.asarray(out["mconf"])[so[k]])
        assert abs(c - thr) <= 10 * tol_conf or _near_tie(g, k, c, tol_conf), (k, c, rep)
    for k in rep["only_ref"]:
        c = float(np.asarray(g["mconf"])[sg[k]])
        assert abs(c - thr) <= 10 * tol_conf or _near_tie(out, k, c, tol_conf), (k, c, rep)
    return rep


def _near_tie(side, key, c, tol):
    """True if on `side` the row/col maximum of conf is within 10*tol of c (a near tie)."""
    b, i, j = key
    if "conf_row_max" in side:
        rm, cm = side["conf_row_max"][b, i], side["conf_col_max"][b, j]
    elif "conf_matrix" in side:
        cmx = np.asarray(side["conf_matrix"])
        rm, cm = cmx[b, i].max(), cmx[b, :, j].max()
    else:
        return False
    return abs(rm - c) <= 10 * tol or abs(cm - c) <= 10 * tol


def check_conf_digest(conf_or_digest, g, tol=TOL_CONF):
    """conf_matrix (or its digest) against the golden digest."""
    d = conf_or_digest
    if not isinstance(d, dict):
        d = make_golden.conf_digest(d)
    for k in ("conf_row_max", "conf_col_max", "conf_sample_val"):
        err = np.abs(np.asarray(d[k], np.float64) - np.asarray(g[k], np.float64)).max()
        assert err <= tol, (k, err)
    for k in ("conf_row_sum", "conf_col_sum"):
        err = np.abs(np.asarray(d[k], np.float64) - np.asarray(g[k], np.float64)).max()
        assert err <= 20 * tol, (k, err)


# ---------------------------------------------------------------------------------------------
# HIP path runner (GPU tests, smoke, bench share it through this module or their own copy)
def build_hip_matcher(cfg, w, device="cuda:0"):
    """loftr_amd.LoFTR with the synthetic hot-path weights loaded (backbone left random)."""
    import copy
    import torch
    from loftr_amd import LoFTR
    model = LoFTR(copy.deepcopy(cfg)).eval()
    sd = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in w.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("backbone.") for k in missing), [k for k in missing if not k.startswith("backbone.")]
    return model.to(device)


def run_hip(inp, device="cuda:0", model=None, materialize_conf=True, channels_last=False):
    """The HIP hot path on a case's inputs -> dict of numpy arrays with the reference's keys."""
    import torch
    model = model or build_hip_matcher(inp["cfg"], inp["w"], device)
    model.coarse_matching.materialize_conf = materialize_conf
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    tf = (lambda a: t(a).contiguous(memory_format=torch.channels_last)) if channels_last else t
    n = inp["feat_c0"].shape[0]
    data = {"bs": n, "hw0_i": torch.Size(inp["hw0_i"]), "hw1_i": torch.Size(inp["hw1_i"])}
    if inp["mask0"] is not None:
        data.update(mask0=t(inp["mask0"]), mask1=t(inp["mask1"]), scale0=t(inp["scale0"]), scale1=t(inp["scale1"]))
    with torch.no_grad():
        model.match_from_features(tf(inp["feat_c0"]), tf(inp["feat_c1"]), tf(inp["feat_f0"]), tf(inp["feat_f1"]), data)
    torch.cuda.synchronize()
    out = {}
    for k, v in data.items():
        if torch.is_tensor(v):
            out[k] = v.detach().cpu().numpy()
        elif v is not None:
            out[k] = v
    return out
