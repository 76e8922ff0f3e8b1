"""Pose estimation of the evaluation caller (estimate_pose, src/utils/metrics.py:72-98).

The reference calls cv2.findEssentialMat(RANSAC) + cv2.recoverPose; OpenCV is absent from this image, so
csrc/pose.hip restates the published algorithms (Nister's five-point solver, Sampson RANSAC, cheirality) as host code.
PARITY UNPINNED against OpenCV: what is checked here is the mathematics -- the minimal solver on exact data, and the
recovered pose / inlier set on synthetic two-view scenes with known ground truth, through the reference's own error
measure (relative_pose_error restated in loftr_amd/evaluation.py, itself pinned to the reference's metrics.py)."""
import ctypes as C

import numpy as np
import pytest

from loftr_amd import _lib, build as build_mod
from loftr_amd import evaluation as EV


@pytest.fixture(scope="module")
def lib():
    build_mod.build(verbose=False)
    return _lib.load()


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _scene(rng, n, noise_px=0.0, outliers=0.0):
    R = _rot(rng.standard_normal(3), 0.1 + 0.4 * rng.random())
    t = rng.standard_normal(3)
    t /= np.linalg.norm(t)
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(3, 9, n)]
    Y = X @ R.T + t
    keep = Y[:, 2] > 0.5
    X, Y = X[keep], Y[keep]
    K0 = np.array([[580.0, 0, 320], [0, 585.0, 240], [0, 0, 1]])
    K1 = np.array([[575.0, 0, 318], [0, 578.0, 243], [0, 0, 1]])
    p0 = (X / X[:, 2:]) @ K0.T
    p1 = (Y / Y[:, 2:]) @ K1.T
    p0, p1 = p0[:, :2] + noise_px * rng.standard_normal((len(X), 2)), p1[:, :2] + noise_px * rng.standard_normal((len(X), 2))
    n_out = int(outliers * len(X))
    is_out = np.zeros(len(X), bool)
    if n_out:
        sel = rng.choice(len(X), n_out, replace=False)
        p1[sel] = np.c_[rng.uniform(0, 640, n_out), rng.uniform(0, 480, n_out)]
        is_out[sel] = True
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return p0.astype(np.float32), p1.astype(np.float32), K0, K1, T, is_out


def test_five_point_exact_minimal_problems(lib):
    rng = np.random.default_rng(0)
    hits = 0
    for _ in range(40):
        R = _rot(rng.standard_normal(3), 0.5 * rng.random())
        t = rng.standard_normal(3)
        t /= np.linalg.norm(t)
        X = np.c_[rng.uniform(-2, 2, 5), rng.uniform(-2, 2, 5), rng.uniform(4, 8, 5)]
        Y = X @ R.T + t
        q0, q1 = np.ascontiguousarray(X[:, :2] / X[:, 2:]), np.ascontiguousarray(Y[:, :2] / Y[:, 2:])
        E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
        E /= np.linalg.norm(E)
        out, ns = np.zeros((10, 9)), C.c_int(0)
        assert lib.loftr_five_point(q0.ctypes.data_as(C.c_void_p), q1.ctypes.data_as(C.c_void_p), 5,
                                    out.ctypes.data_as(C.c_void_p), C.byref(ns)) == 0
        assert 1 <= ns.value <= 10
        for k in range(ns.value):                       # every solution satisfies the constraints it was derived from
            Ek = out[k].reshape(3, 3)
            assert abs(np.linalg.det(Ek)) < 1e-6
            assert np.abs(2 * Ek @ Ek.T @ Ek - np.trace(Ek @ Ek.T) * Ek).max() < 1e-5
            assert np.abs(np.einsum("ni,ij,nj->n", np.c_[q1, np.ones(5)], Ek, np.c_[q0, np.ones(5)])).max() < 1e-7
        d = min(min(np.abs(out[k] - E.ravel()).max(), np.abs(out[k] + E.ravel()).max()) for k in range(ns.value))
        hits += d < 1e-6
    assert hits == 40


@pytest.mark.parametrize("noise,outliers,r_tol,t_tol", [(0.0, 0.0, 0.01, 0.05), (0.3, 0.0, 0.5, 2.5), (0.3, 0.4, 0.7, 3.5)])
def test_estimate_pose_recovers_ground_truth(lib, noise, outliers, r_tol, t_tol):
    rng = np.random.default_rng(7)
    r_errs, t_errs = [], []
    for trial in range(8):
        p0, p1, K0, K1, T, is_out = _scene(rng, 400, noise, outliers)
        ret = EV.estimate_pose_native(p0, p1, K0, K1, 0.5, conf=0.99999, seed=trial)
        assert ret is not None
        R, t, inl = ret
        assert abs(np.linalg.det(R) - 1) < 1e-5 and abs(np.linalg.norm(t) - 1) < 1e-5
        t_err, R_err = EV.relative_pose_error(T, R, t, ignore_gt_t_thr=0.0)
        r_errs.append(R_err)
        t_errs.append(t_err)
        if outliers:
            assert inl[is_out].mean() < 0.05            # outliers rejected
            assert inl[~is_out].mean() > 0.6            # 0.5 px threshold on 0.3 px noise per coordinate, both images
        elif noise == 0:
            assert inl.mean() > 0.99
    assert np.median(r_errs) <= r_tol and np.median(t_errs) <= t_tol, (r_errs, t_errs)


def test_estimate_pose_none_cases(lib):
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    assert EV.estimate_pose_native(np.zeros((4, 2), np.float32), np.zeros((4, 2), np.float32), K, K, 0.5) is None      # < 5 points


def test_compute_pose_errors_uses_native_estimator_without_cv2(lib):
    import torch
    rng = np.random.default_rng(3)
    scenes = [_scene(rng, 300, 0.2, 0.2) for _ in range(2)]
    data = {"m_bids": torch.cat([torch.full((len(s[0]),), b, dtype=torch.int64) for b, s in enumerate(scenes)]),
            "mkpts0_f": torch.from_numpy(np.concatenate([s[0] for s in scenes])),
            "mkpts1_f": torch.from_numpy(np.concatenate([s[1] for s in scenes])),
            "K0": torch.from_numpy(np.stack([s[2] for s in scenes])), "K1": torch.from_numpy(np.stack([s[3] for s in scenes])),
            "T_0to1": torch.from_numpy(np.stack([s[4] for s in scenes]))}
    try:
        import cv2  # noqa: F401
        pytest.skip("OpenCV present: the reference's estimator is used")
    except ImportError:
        pass
    with pytest.raises(ImportError):                      # the default keeps the reference's behaviour: no cv2, no pose
        EV.compute_pose_errors(dict(data))
    d_inf = dict(data)
    EV.compute_pose_errors(d_inf, on_missing="inf")
    assert d_inf["R_errs"] == [np.inf, np.inf] and d_inf["pose_estimator"] == "none (inf)"
    with pytest.warns(UserWarning, match="parity"):        # explicit opt-in to the unpinned estimator, flagged once
        EV._WARNED_NATIVE.clear()
        EV.compute_pose_errors(data, on_missing="native")
    assert data["pose_estimator"] == "estimate_pose_native"
    assert len(data["R_errs"]) == 2 and max(data["R_errs"]) < 1.0 and max(data["t_errs"]) < 5.0
    assert all(len(i) == len(s[0]) for i, s in zip(data["inliers"], scenes))
    d2 = dict(data)
    EV.compute_pose_errors(d2, estimator=EV.estimate_pose_native)          # the explicit form
    assert d2["pose_estimator"] == "estimate_pose_native" and max(d2["R_errs"]) < 1.0
