"""CPU restatement of the reference's evaluation caller: what `test_step`
(src/lightning/lightning_loftr.py:205-229) computes from the matcher's output.

TEST INFRASTRUCTURE ONLY -- imported by tests/ (and nothing else).  The product code is
loftr_amd/evaluation.py + csrc/eval.hip; it never imports this module.

Pinned against the real `src/utils/metrics.py` (imported through oracle/ref_shim.py) by the golden vectors
in tests/golden/metrics_*.npz (generator: tests/golden/make_golden_metrics.py).

NOT restated: `estimate_pose` / `compute_pose_errors` (metrics.py:71-140).  They are thin wrappers around
OpenCV's `cv2.findEssentialMat(method=RANSAC)` + `cv2.recoverPose` (opencv-python 4.4.0.46 in the reference's
environment.yaml); OpenCV is not installed in this image, its RANSAC draws from a library-internal RNG and
cannot be pinned without the library -> out of scope, `relative_pose_error` below is the part
of that path that is reference code.
"""
import numpy as np


def cross_product_matrix(t):
    """kornia.geometry.epipolar.numeric.cross_product_matrix (kornia 0.4.1): [N,3] -> [N,3,3]."""
    t = np.asarray(t, np.float32)
    z = np.zeros_like(t[:, 0])
    return np.stack([z, -t[:, 2], t[:, 1], t[:, 2], z, -t[:, 0], -t[:, 1], t[:, 0], z], -1).reshape(-1, 3, 3)


def essential_from_pose(T_0to1):
    """metrics.py:55-56: E = [t]_x R of the ground-truth relative pose, fp32."""
    T = np.asarray(T_0to1, np.float32)
    return np.matmul(cross_product_matrix(T[:, :3, 3]), T[:, :3, :3]).astype(np.float32)


def symmetric_epipolar_distance(pts0, pts1, E, K0, K1):
    """metrics.py:31-47, fp32: squared symmetric epipolar distance of [M,2] pixel matches."""
    pts0 = np.asarray(pts0, np.float32)
    pts1 = np.asarray(pts1, np.float32)
    E, K0, K1 = np.asarray(E, np.float32), np.asarray(K0, np.float32), np.asarray(K1, np.float32)
    p0 = (pts0 - np.array([K0[0, 2], K0[1, 2]], np.float32)[None]) / np.array([K0[0, 0], K0[1, 1]], np.float32)[None]
    p1 = (pts1 - np.array([K1[0, 2], K1[1, 2]], np.float32)[None]) / np.array([K1[0, 0], K1[1, 1]], np.float32)[None]
    one = np.ones((p0.shape[0], 1), np.float32)
    p0 = np.concatenate([p0, one], 1)
    p1 = np.concatenate([p1, one], 1)
    Ep0 = (p0 @ E.T).astype(np.float32)
    p1Ep0 = np.sum(p1 * Ep0, -1, dtype=np.float32)
    Etp1 = (p1 @ E).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        d = p1Ep0 ** 2 * (np.float32(1.0) / (Ep0[:, 0] ** 2 + Ep0[:, 1] ** 2) + np.float32(1.0) / (Etp1[:, 0] ** 2 + Etp1[:, 1] ** 2))
    return d.astype(np.float32)


def compute_symmetrical_epipolar_errors(mkpts0_f, mkpts1_f, m_bids, T_0to1, K0, K1):
    """metrics.py:50-68: per-match error, concatenated pair by pair in ascending pair order (torch.cat over bs).
    Because the matcher emits matches grouped by ascending pair id, that equals the match order."""
    E = essential_from_pose(T_0to1)
    m_bids = np.asarray(m_bids)
    out = []
    for b in range(E.shape[0]):
        mask = m_bids == b
        out.append(symmetric_epipolar_distance(np.asarray(mkpts0_f)[mask], np.asarray(mkpts1_f)[mask], E[b], K0[b], K1[b]))
    return np.concatenate(out, 0) if out else np.zeros((0,), np.float32)


def relative_pose_error(T_0to1, R, t, ignore_gt_t_thr=0.0):
    """metrics.py:12-28 (float64 numpy like the reference): angular errors in degrees -> (t_err, R_err)."""
    T_0to1, R, t = np.asarray(T_0to1), np.asarray(R), np.asarray(t)
    t_gt = T_0to1[:3, 3]
    n = np.linalg.norm(t) * np.linalg.norm(t_gt)
    t_err = np.rad2deg(np.arccos(np.clip(np.dot(t, t_gt) / n, -1.0, 1.0)))
    t_err = np.minimum(t_err, 180 - t_err)
    if np.linalg.norm(t_gt) < ignore_gt_t_thr:
        t_err = 0
    R_gt = T_0to1[:3, :3]
    cos = (np.trace(np.dot(R.T, R_gt)) - 1) / 2
    cos = np.clip(cos, -1.0, 1.0)
    R_err = np.rad2deg(np.abs(np.arccos(cos)))
    return t_err, R_err


def error_auc(errors, thresholds=(5, 10, 20)):
    """metrics.py:143-160.  (The reference overwrites its `thresholds` argument with [5, 10, 20].)"""
    thresholds = [5, 10, 20]
    errors = [0] + sorted(list(errors))
    recall = list(np.linspace(0, 1, len(errors)))
    aucs = []
    for thr in thresholds:
        last_index = int(np.searchsorted(errors, thr))
        y = recall[:last_index] + [recall[last_index - 1]]
        x = errors[:last_index] + [thr]
        # trapezoid rule written out (np.trapz was removed in numpy 2)
        area = 0.0
        for i in range(1, len(x)):
            area += (x[i] - x[i - 1]) * (y[i] + y[i - 1]) / 2.0
        aucs.append(area / thr)
    return {f"auc@{t}": auc for t, auc in zip(thresholds, aucs)}


def epidist_prec(errors, thresholds, ret_dict=False):
    """metrics.py:163-174: mean over pairs of the fraction of matches with epipolar error < thr."""
    precs = []
    for thr in thresholds:
        prec_ = []
        for errs in errors:
            correct = np.asarray(errs) < thr
            prec_.append(np.mean(correct) if len(correct) > 0 else 0)
        precs.append(np.mean(prec_) if len(prec_) > 0 else 0)
    if ret_dict:
        return {f"prec@{t:.0e}": prec for t, prec in zip(thresholds, precs)}
    return precs


def aggregate_metrics(metrics, epi_err_thr=5e-4):
    """metrics.py:177-198: de-duplicate by identifier (LAST occurrence's index wins, first occurrence's
    position orders -- OrderedDict semantics), pose AUC on max(R_err, t_err), matching precision."""
    unq = {}
    for idx, iden in enumerate(metrics["identifiers"]):
        unq[iden] = idx                                   # dict keeps first-insertion order, last value
    unq_ids = list(unq.values())
    pose_errors = np.max(np.stack([metrics["R_errs"], metrics["t_errs"]]), axis=0)[unq_ids]
    aucs = error_auc(pose_errors)
    epi = [metrics["epi_errs"][i] for i in unq_ids]
    precs = epidist_prec(epi, [epi_err_thr], True)
    return {**aucs, **precs}
