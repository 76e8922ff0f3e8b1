"""Golden for EXACT TIES in coarse matching: the reference's own CoarseMatching (dual-softmax, eval) on descriptors
with deliberately duplicated rows in image 1 (coarse_matching.py:187-193: threshold / border / mutual-maximum masks
are ANDed and the FIRST surviving column of a row is taken).

    python tests/golden/make_golden_ties.py        # authoring container only (needs /root/reference)

Construction (8 x 10 coarse cells, L = S = 80, border_rm = 1, thr = 0): several BORDER cells of image 1 receive a
copy of the descriptor of a later INTERIOR cell, so the two columns of the score matrix -- and of conf_matrix -- are
bitwise identical and every row whose maximum sits there attains it twice, first in a column the border mask removes.
The reference then emits the later, interior column; an implementation that takes the first arg-max and tests it
afterwards drops the row.  The script asserts that the reference run really contains such rows."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

H0 = H1 = 8
W0 = W1 = 10
C = 256
DUPS = [(3, 34), (9, 47), (20, 55), (70, 61), (5, 34)]     # (border cell <- copy of interior cell); 34 is copied twice


def make_inputs(seed=5):
    rng = np.random.default_rng(seed)
    f0 = rng.standard_normal((2, H0 * W0, C)).astype(np.float32)
    f1 = (0.6 * np.roll(f0, 7, axis=1) + 0.8 * rng.standard_normal((2, H1 * W1, C))).astype(np.float32)
    for dst, src in DUPS:
        f1[:, dst] = f1[:, src]
    f1[1, 12] = f1[1, 77]          # pair 1: an interior duplicate BEFORE its twin (first one must win), and
    f0[1, 40] = f0[1, 41]          # two identical ROWS (each keeps its own first column)
    return f0, f1


def run_reference(f0, f1, thr=0.0, border_rm=1, match_type="dual_softmax"):
    from oracle.ref_shim import import_reference
    import_reference()
    from src.loftr.utils.coarse_matching import CoarseMatching
    cfg = dict(thr=thr, border_rm=border_rm, train_coarse_percent=0.4, train_pad_num_gt_min=200, match_type=match_type, dsmax_temperature=0.1)
    if match_type == "sinkhorn":                         # coarse_matching.py:121-143 (round 3: the tie golden for the OT branch)
        cfg.update(skh_init_bin_score=1.0, skh_iters=3, skh_prefilter=False, sparse_spvs=False)
    cm = CoarseMatching(cfg).eval()
    data = {"hw0_i": (H0 * 8, W0 * 8), "hw1_i": (H1 * 8, W1 * 8), "hw0_c": (H0, W0), "hw1_c": (H1, W1)}
    with torch.no_grad():
        cm(torch.from_numpy(f0), torch.from_numpy(f1), data)
    return {k: data[k].numpy() for k in ("conf_matrix", "b_ids", "i_ids", "j_ids", "mconf", "mkpts0_c", "mkpts1_c")}


if __name__ == "__main__":
    f0, f1 = make_inputs()
    out = run_reference(f0, f1)
    conf = out["conf_matrix"]
    for dst, src in DUPS:
        assert np.array_equal(conf[:, :, dst], conf[:, :, src]), "the reference's conf columns of duplicated descriptors are not bitwise equal"
    later = sum(1 for b, i, j in zip(out["b_ids"], out["i_ids"], out["j_ids"])
                if any(j == src and conf[b, i, dst] == conf[b, i, j] and dst < src for dst, src in DUPS))
    assert later >= 2, later
    print(f"M={len(out['mconf'])}, matches that are a LATER tied column: {later}")
    np.savez_compressed(os.path.join(HERE, "ties_ds.npz"), feat_c0=f0, feat_c1=f1, thr=0.0, border_rm=1,
                        **{k: v for k, v in out.items() if k != "conf_matrix"}, conf_row_max=conf.max(2), conf_col_max=conf.max(1))
    print("->", os.path.join(HERE, "ties_ds.npz"), os.path.getsize(os.path.join(HERE, "ties_ds.npz")) // 1000, "kB")
    # ---- the same descriptors through the Sinkhorn branch: duplicated descriptors give bitwise equal columns of the assignment too
    out = run_reference(f0, f1, match_type="sinkhorn")
    conf = out["conf_matrix"]
    for dst, src in DUPS:
        assert np.array_equal(conf[:, :, dst], conf[:, :, src]), "sinkhorn: conf columns of duplicated descriptors are not bitwise equal"
    later = sum(1 for b, i, j in zip(out["b_ids"], out["i_ids"], out["j_ids"])
                if any(j == src and conf[b, i, dst] == conf[b, i, j] and dst < src for dst, src in DUPS))
    assert later >= 2, later
    print(f"sinkhorn: M={len(out['mconf'])}, matches that are a LATER tied column: {later}")
    np.savez_compressed(os.path.join(HERE, "ties_ot.npz"), feat_c0=f0, feat_c1=f1, thr=0.0, border_rm=1,
                        **{k: v for k, v in out.items() if k != "conf_matrix"}, conf_row_max=conf.max(2), conf_col_max=conf.max(1))
    print("->", os.path.join(HERE, "ties_ot.npz"), os.path.getsize(os.path.join(HERE, "ties_ot.npz")) // 1000, "kB")
