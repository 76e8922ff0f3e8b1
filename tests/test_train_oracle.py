"""Training-side consumers of the matching path, forward values: the numpy restatement
(oracle/train_oracle.py) against goldens produced by the reference's own spvs_coarse / spvs_fine / LoFTRLoss
(tests/golden/make_golden_train.py)."""
import importlib.util
import json
import os

import numpy as np
import pytest

from _cases import GOLDEN_DIR

_spec = importlib.util.spec_from_file_location("make_golden_train", os.path.join(GOLDEN_DIR, "make_golden_train.py"))
MG = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MG)
CASES = list(MG.CASES)
LOSS_CFGS = {"focal_sparse_ds": ("focal", True, "dual_softmax", "l2_with_std"), "focal_sparse_ot": ("focal", True, "sinkhorn", "l2_with_std"),
             "focal_dense_ds": ("focal", False, "dual_softmax", "l2"), "ce_dense_ds": ("cross_entropy", False, "dual_softmax", "l2_with_std")}


def load(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))
    rc = json.loads(str(g["recipe"]))
    return rc, MG.make_inputs(rc), g


def spv_keys(d):
    return list(zip(d["spv_b_ids"].tolist(), d["spv_i_ids"].tolist(), d["spv_j_ids"].tolist()))


def check_spvs(out, g, max_flips=2):
    """GT match sets equal up to a couple of cells whose warped coordinate sits on a rounding boundary (the 3x3
    intrinsics inverse is not evaluated in the same order as torch.inverse)."""
    a, b = set(spv_keys(out)), set(spv_keys(g))
    assert len(a ^ b) <= max_flips, sorted(a ^ b)
    assert np.abs(out["spv_w_pt0_i"] - g["spv_w_pt0_i"]).max() <= 2e-3 * max(1.0, np.abs(g["spv_w_pt0_i"]).max() / 100)
    assert np.array_equal(out["spv_pt1_i"], g["spv_pt1_i"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_supervision_and_losses(name):
    from oracle import train_oracle as T
    rc, inp, g = load(name)
    out = T.spvs_coarse((rc["H"], rc["W"]), (rc["H"], rc["W"]), inp["depth0"], inp["depth1"], inp["T_0to1"], inp["T_1to0"],
                        inp["K0"], inp["K1"], 8, inp.get("scale0"), inp.get("scale1"), inp.get("mask0"), inp.get("mask1"))
    check_spvs(out, g)
    assert np.allclose(out["conf_matrix_gt"].sum((1, 2)), g["conf_gt_sum"], atol=2)
    ef = T.spvs_fine(g["spv_w_pt0_i"], g["spv_pt1_i"], g["b_ids"], g["i_ids"], g["j_ids"], 2, 2, inp.get("scale1"))
    assert np.abs(ef - g["expec_f_gt"]).max() <= 1e-5 * max(1.0, np.abs(g["expec_f_gt"]).max())
    # losses on the reference's own GT
    N, L = rc["N"], (rc["H"] // 8) * (rc["W"] // 8)
    gt = np.zeros((N, L, L), np.float32)
    if g["conf_gt_sum"].sum() > 0:                        # else: the reference's placeholder ids (0, 0, 0), conf_matrix_gt all zero
        gt[g["spv_b_ids"], g["spv_i_ids"], g["spv_j_ids"]] = 1
    conf, conf_bin = MG.replay_matcher_outputs(rc, g["spv_b_ids"], g["spv_i_ids"], g["spv_j_ids"])
    weight = None
    if "mask0" in inp:
        weight = (inp["mask0"].reshape(N, -1)[..., None] * inp["mask1"].reshape(N, -1)[:, None]).astype(np.float32)
    want = json.loads(str(g["losses"]))
    for tag, (ctype, sparse, mtype, ftype) in LOSS_CFGS.items():
        c = conf_bin if (sparse and mtype == "sinkhorn") else conf
        lc = T.coarse_loss(c, gt, weight, ctype, sparse, mtype)
        lf = T.fine_loss(g["expec_f"], g["expec_f_gt"], ftype)
        assert abs(lc - want[tag]["loss_c"]) <= 2e-6 * max(1, abs(want[tag]["loss_c"])), (tag, lc, want[tag])
        if lf is None:                                     # eval mode, no correct coarse match: loss_scalars carry 1.0 (loftr_loss.py:186-188)
            assert want[tag]["loss_f"] == 1.0 and abs(want[tag]["loss"] - want[tag]["loss_c"]) <= 1e-7
        else:
            assert abs(lf - want[tag]["loss_f"]) <= 2e-6, (tag, lf, want[tag])
