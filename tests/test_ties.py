"""Exact ties at the row maximum of conf_matrix (SURVEY.md §4 edge case; coarse_matching.py:187-193): the reference
ANDs the threshold / border / mutual-maximum masks over the row and emits the FIRST surviving column, which may be a
later tied column when the first one falls to the border mask.  Golden: tests/golden/ties_ds.npz, produced by the
reference's own CoarseMatching on descriptors with duplicated rows (make_golden_ties.py)."""
import os

import numpy as np
import pytest

from _cases import GOLDEN_DIR, TOL_CONF

H, W = 8, 10


def _golden(match_type="dual_softmax"):
    return dict(np.load(os.path.join(GOLDEN_DIR, "ties_ds.npz" if match_type == "dual_softmax" else "ties_ot.npz")))


def _ids(d):
    return list(zip(np.asarray(d["b_ids"]).tolist(), np.asarray(d["i_ids"]).tolist(), np.asarray(d["j_ids"]).tolist()))


@pytest.mark.parametrize("match_type", ["dual_softmax", "sinkhorn"])
def test_oracle_takes_first_surviving_tied_column(match_type):
    from oracle import loftr_oracle as O
    g = _golden(match_type)
    if match_type == "dual_softmax":
        conf = O.dual_softmax_conf(g["feat_c0"], g["feat_c1"], 0.1)
    else:                                                 # coarse_matching.py:121-143, bin_score 1.0, 3 iterations, no prefilter
        conf = O.sinkhorn_conf(g["feat_c0"], g["feat_c1"], np.float32(1.0), iters=3)
        conf = conf[0] if isinstance(conf, tuple) else conf
    sel = O.coarse_match_select(conf, float(g["thr"]), int(g["border_rm"]), (H, W), (H, W), (H * 8, W * 8))
    assert _ids(sel) == _ids(g)
    assert np.abs(sel["mconf"] - g["mconf"]).max() <= TOL_CONF


@pytest.mark.gpu
@pytest.mark.parametrize("match_type", ["dual_softmax", "sinkhorn"])
def test_hip_takes_first_surviving_tied_column(match_type):
    import torch
    from loftr_amd import ops
    from oracle import loftr_oracle as O
    g = _golden(match_type)
    dev = "cuda:0"
    f0, f1 = torch.from_numpy(g["feat_c0"]).to(dev), torch.from_numpy(g["feat_c1"]).to(dev)
    r = ops.coarse_match(f0, f1, (H, W), (H, W), thr=float(g["thr"]), border_rm=int(g["border_rm"]), scale=8.0,
                         match_type=match_type, **(dict(temperature=0.1) if match_type == "dual_softmax" else dict(bin_score=1.0, skh_iters=3)))
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in r.items() if torch.is_tensor(v)}
    # 1. identical to the reference, later tied columns included
    assert _ids(out) == _ids(g), (sorted(set(_ids(g)) - set(_ids(out))), sorted(set(_ids(out)) - set(_ids(g))))
    assert np.abs(out["mconf"] - g["mconf"]).max() <= TOL_CONF
    assert np.array_equal(out["mkpts1_c"], g["mkpts1_c"]) and np.array_equal(out["mkpts0_c"], g["mkpts0_c"])
    # 2. the duplicated columns of the device conf_matrix are bitwise equal (what makes these ties exact) ...
    conf = out["conf_matrix"]
    assert np.array_equal(conf[:, :, 3], conf[:, :, 34]) and np.array_equal(conf[:, :, 5], conf[:, :, 34])
    # 3. ... and the selection kernels implement exactly the reference semantics on that matrix
    sel = O.coarse_match_select(conf, float(g["thr"]), int(g["border_rm"]), (H, W), (H, W), (H * 8, W * 8))
    assert _ids(sel) == _ids(out)
    later = [(b, i, j) for b, i, j in _ids(out) if (conf[b, i, :j] == conf[b, i, j]).any()]
    assert len(later) >= 2, "the tie path was not exercised"


@pytest.mark.gpu
def test_match_counts_small_grids_many_pairs():
    """Per-pair match counts when a wave's 64 consecutive rows span MORE than two pairs (L < 32): ADVICE r1."""
    import torch
    from loftr_amd import ops
    rng = np.random.default_rng(3)
    N, h, w = 7, 4, 5
    f0 = rng.standard_normal((N, h * w, 256)).astype(np.float32)
    f1 = (0.7 * f0 + 0.7 * rng.standard_normal((N, h * w, 256))).astype(np.float32)
    dev = "cuda:0"
    r = ops.coarse_match(torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev), (h, w), (h, w), thr=0.0, border_rm=0,
                         scale=8.0, match_type="dual_softmax", temperature=0.1)
    torch.cuda.synchronize()
    counts = r["counts"].cpu().numpy()
    b = r["b_ids"].cpu().numpy()
    assert counts[0] == len(b) > N
    assert np.array_equal(counts[1:], np.bincount(b, minlength=N))
