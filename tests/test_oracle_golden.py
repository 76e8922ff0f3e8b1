"""Pins oracle/loftr_oracle.py against the golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import loftr_oracle as O
from _cases import SMALL_CASES, MID_CASES, FULL_CASES, load_case, compare_to_golden, check_conf_digest


def run_oracle(inp, **kw):
    return O.loftr_hot_path(inp["feat_c0"], inp["feat_c1"], inp["feat_f0"], inp["feat_f1"], inp["w"],
                            inp["cfg"], inp["hw0_i"], inp["hw1_i"], inp["mask0"], inp["mask1"],
                            inp["scale0"], inp["scale1"], **kw)


@pytest.mark.parametrize("name", SMALL_CASES + MID_CASES)
def test_oracle_matches_reference(name):
    rc, inp, g = load_case(name)
    out = run_oracle(inp, keep_intermediates=True)
    rep = compare_to_golden(out, g, inp["cfg"]["match_coarse"]["thr"])
    assert rep["M_out"] == rep["M_ref"]
    check_conf_digest(out["conf_matrix"], g)
    if "conf_matrix" in g:
        assert np.abs(out["conf_matrix"] - g["conf_matrix"]).max() <= 1e-4
    if "conf_matrix_with_bin" in g:
        assert np.abs(out["conf_matrix_with_bin"] - g["conf_matrix_with_bin"]).max() <= 1e-4
    if "feat_c0" in g:
        assert np.abs(out["feat_c0"] - g["feat_c0"]).max() <= 2e-4
        assert np.abs(out["feat_c1"] - g["feat_c1"]).max() <= 2e-4
    if "feat_f0_unfold" in g and len(g["feat_f0_unfold"]):
        h = len(g["feat_f0_unfold"])
        for k in ("feat_f0_unfold_pre", "feat_f1_unfold_pre", "feat_f0_unfold", "feat_f1_unfold"):
            assert np.abs(out[k][:h] - g[k]).max() <= 5e-4, k
    assert out["expec_f"].shape == g["expec_f"].shape
    if len(g["expec_f"]):
        # std column is ill-conditioned near sqrt(clamp(var,1e-10)): loose bound
        assert np.abs(out["expec_f"][:, 2] - g["expec_f"][:, 2]).max() <= 5e-3


@pytest.mark.slow
@pytest.mark.parametrize("name", [
    "full_ds_thr0",                                                     # BASELINE geometry (L = S = 4800), ~50 s of numpy
    pytest.param("full_ds_thr02", marks=pytest.mark.skipif(            # N = 2, 2.5 min: opt-in
        os.environ.get("LOFTR_SLOW_TESTS") != "1", reason="set LOFTR_SLOW_TESTS=1 (N=2 full-size oracle run, ~2.5 min)")),
])
def test_oracle_matches_reference_full(name):
    rc, inp, g = load_case(name)
    out = run_oracle(inp)
    compare_to_golden(out, g, inp["cfg"]["match_coarse"]["thr"], max_flips=2)
    check_conf_digest(out["conf_matrix"], g)


def test_fp64_headroom():
    """fp32 oracle vs fp64 oracle: how far fp32 re-association can move the graded outputs."""
    rc, inp, g = load_case("small_ds_corr")
    o32 = run_oracle(inp)
    o64 = run_oracle(inp, dtype=np.float64)
    assert len(o32["mconf"]) == len(o64["mconf"])
    assert np.abs(o32["mconf"] - o64["mconf"]).max() < 1e-4
    assert np.abs(o32["mkpts1_f"] - o64["mkpts1_f"]).max() < 1e-3


def test_position_encoding_quirk():
    """temp_bug_fix=False uses exp(-k): position_encoding.py:28 evaluates to a factor of -1.0."""
    pe = O.position_encoding_table(256, 4, 4, temp_bug_fix=False)
    assert np.isclose(pe[4, 0, 0], np.sin(np.float32(1.0) * np.exp(np.float32(-2.0))), atol=1e-6)
    pe = O.position_encoding_table(256, 4, 4, temp_bug_fix=True)
    assert np.isclose(pe[4, 0, 1], np.sin(2.0 * np.exp(-2.0 * np.log(10000.0) / 128)), atol=1e-6)
