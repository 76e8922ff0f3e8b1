"""Evaluation caller: the numpy oracle and the product's host-side aggregation against
golden vectors produced by the reference's own src/utils/metrics.py (tests/golden/make_golden_metrics.py)."""
import os

import numpy as np
import pytest

from oracle import metrics_oracle as mo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def epi_close(got, ref):
    """Tolerance for squared epipolar distances evaluated in fp32: d = (p1' E p0)^2 * w, and p1' E p0 is a 3-term
    dot product that cancels to ~0 for inliers, so its ABSOLUTE error (~1e-8: association order of the terms) is
    what two fp32 evaluations share -- compared on the distance sqrt(d): 1e-6 absolute + 1e-4 relative (the
    thresholds applied downstream are d < 1e-4 .. 5e-4, i.e. sqrt(d) ~ 1e-2)."""
    a, b = np.sqrt(np.asarray(got, np.float64)), np.sqrt(np.asarray(ref, np.float64))
    return a.shape == b.shape and bool(np.all(np.abs(a - b) <= 1e-6 + 1e-4 * b))


@pytest.fixture(scope="module")
def epi():
    return np.load(os.path.join(GOLD, "metrics_epi.npz"))


@pytest.fixture(scope="module")
def agg():
    return np.load(os.path.join(GOLD, "metrics_agg.npz"), allow_pickle=False)


def _metrics_from(agg):
    lens = agg["epi_lens"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    epi = [agg["epi_flat"][offs[i]:offs[i + 1]] for i in range(len(lens))]
    return dict(identifiers=[str(s) for s in agg["ids"]], R_errs=list(agg["R_errs"]), t_errs=list(agg["t_errs"]), epi_errs=epi)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_oracle_epipolar_errors_match_reference(epi, case):
    got = mo.compute_symmetrical_epipolar_errors(epi[f"{case}_mkpts0_f"], epi[f"{case}_mkpts1_f"], epi[f"{case}_m_bids"],
                                                 epi[f"{case}_T_0to1"], epi[f"{case}_K0"], epi[f"{case}_K1"])
    ref = epi[f"{case}_epi_errs"]
    assert got.shape == ref.shape and got.dtype == np.float32
    assert epi_close(got, ref)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_relative_pose_error_matches_reference(epi, impl):
    if impl == "oracle":
        m = mo
    else:
        from loftr_amd import evaluation as m
    for key, thr in (("rpe_errs", 0.0), ("rpe_errs_thr", 10.0)):
        got = np.array([m.relative_pose_error(epi["rpe_T"][i], epi["rpe_R"][i], epi["rpe_t"][i], thr) for i in range(6)], dtype=np.float64)
        assert np.allclose(got, epi[key], rtol=0, atol=1e-9)
    assert np.all(epi["rpe_errs_thr"][:, 0] == 0)              # |t_gt| < 10: translation error ignored
    assert epi["rpe_errs"][2, 1] < 5e-2                         # identical rotation (fp32-rounded R_gt: ~1e-2 deg)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_aggregation_matches_reference(agg, impl):
    if impl == "oracle":
        m = mo
    else:
        from loftr_amd import evaluation as m
    metrics = _metrics_from(agg)
    for thr in (5e-4, 1e-4):
        got = m.aggregate_metrics(dict(metrics), thr)
        keys = [k for k in agg.files if k.startswith(f"agg_thr_{thr:g}_")]
        assert len(keys) == 4 and len(got) == 4
        for k in keys:
            name = k[len(f"agg_thr_{thr:g}_"):]
            assert abs(got[name] - float(agg[k])) <= 1e-12, (name, got[name], float(agg[k]))
    auc = m.error_auc(agg["auc_errs"], [5, 10, 20])
    for t in (5, 10, 20):
        assert abs(auc[f"auc@{t}"] - float(agg[f"auc_auc@{t}"])) <= 1e-12
    prec = m.epidist_prec(metrics["epi_errs"], [1e-4, 5e-4, 1e-3], False)
    assert np.allclose(prec, agg["prec"], rtol=0, atol=1e-12)
    # duplicates: 40 items, 31 unique identifiers
    assert len(set(metrics["identifiers"])) == 31
