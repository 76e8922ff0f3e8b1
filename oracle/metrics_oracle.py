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
    """metrics.py:12-28 in float64: (t_err, R_err) in degrees.
    t_err: angle between estimated and true translation DIRECTION, folded into [0, 90] (the essential matrix leaves the
    sign open), forced to 0 when the true baseline is shorter than `ignore_gt_t_thr`;
    R_err: rotation angle of R^T R_gt, from its trace."""
    T = np.asarray(T_0to1, np.float64)
    R = np.asarray(R, np.float64)
    t = np.asarray(t, np.float64)
    t_gt = T[:3, 3]
    len_gt = float(np.sqrt(np.sum(t_gt * t_gt)))
    len_t = float(np.sqrt(np.sum(t * t)))
    c = float(np.sum(t * t_gt)) / (len_t * len_gt)
    c = max(-1.0, min(1.0, c))
    ang = np.arccos(c) * 180.0 / np.pi
    t_err = ang if ang <= 180.0 - ang else 180.0 - ang
    if len_gt < ignore_gt_t_thr:
        t_err = 0
    tr = 0.0
    for i in range(3):
        for k in range(3):
            tr += R[k, i] * T[k, i]                      # trace(R^T R_gt)
    c = max(-1.0, min(1.0, (tr - 1.0) / 2.0))
    R_err = abs(np.arccos(c)) * 180.0 / np.pi
    return t_err, R_err


def error_auc(errors, thresholds=(5, 10, 20)):
    """metrics.py:143-160.  Recall curve: sorted errors e_1..e_n with a leading 0, recall = linspace(0, 1, n+1); for each
    threshold the curve is cut at the first sample >= thr, extended flat to thr, integrated by the trapezoid rule and
    divided by thr.  (The reference replaces whatever `thresholds` it is given by [5, 10, 20].)"""
    xs = [0.0] + sorted(float(e) for e in errors)
    ys = [float(v) for v in np.linspace(0, 1, len(xs))]
    out = {}
    for thr in (5, 10, 20):
        cut = 0
        while cut < len(xs) and xs[cut] < thr:          # == np.searchsorted(xs, thr), side='left'
            cut += 1
        px = xs[:cut] + [float(thr)]
        py = ys[:cut] + [ys[cut - 1]]
        area = 0.0
        for i in range(1, len(px)):
            area += (px[i] - px[i - 1]) * (py[i] + py[i - 1]) / 2.0
        out[f"auc@{thr}"] = area / thr
    return out


def epidist_prec(errors, thresholds, ret_dict=False):
    """metrics.py:163-174: per threshold, the average over pairs of (#matches with error < thr) / (#matches), a pair
    without matches contributing 0."""
    result = []
    for thr in thresholds:
        total = 0.0
        for errs in errors:
            n_ok = sum(1 for e in errs if e < thr)
            total += (n_ok / len(errs)) if len(errs) > 0 else 0.0
        result.append(total / len(errors) if len(errors) > 0 else 0)
    if ret_dict:
        return {f"prec@{t:.0e}": r for t, r in zip(thresholds, result)}
    return result


def aggregate_metrics(metrics, epi_err_thr=5e-4):
    """metrics.py:177-198.  Duplicated identifiers (sampler padding) are collapsed: the LAST occurrence supplies the
    values, the FIRST occurrence fixes the position (OrderedDict((iden, id) ...) semantics); pose error of a pair =
    max(R_err, t_err)."""
    order, last = [], {}
    for idx, iden in enumerate(metrics["identifiers"]):
        if iden not in last:
            order.append(iden)
        last[iden] = idx
    picked = [last[iden] for iden in order]
    pose = [max(metrics["R_errs"][i], metrics["t_errs"][i]) for i in picked]
    res = dict(error_auc(pose))
    res.update(epidist_prec([metrics["epi_errs"][i] for i in picked], [epi_err_thr], True))
    return res
