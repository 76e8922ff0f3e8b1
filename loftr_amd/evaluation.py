"""Lightning-free evaluation caller of the matching path.

Mirrors what the reference's ``PL_LoFTR.test_step`` / ``test_epoch_end`` do around ``matcher(batch)``
(src/lightning/lightning_loftr.py:95-111, 205-249) with the same function names, batch-dict keys, return
structure and ``LoFTR_pred_eval.npy`` dump format, so an evaluation script can swap

    from src.utils.metrics import compute_symmetrical_epipolar_errors, compute_pose_errors, aggregate_metrics

for ``from loftr_amd.evaluation import ...``.

* per-match epipolar errors: HIP kernel ``loftr_epipolar_errors`` (csrc/eval.hip), device tensors in, device
  tensor out -- no host round trip between the matcher and its first consumer;
* aggregation (AUC / precision over a dataset: a few thousand scalars, once per dataset) is host-side numpy like
  the reference's;
* pose estimation (metrics.py:71-140) is OpenCV in the reference (`cv2.findEssentialMat` RANSAC + `cv2.recoverPose`):
  used when cv2 is importable; otherwise `estimate_pose_native` -- the library's own five-point RANSAC + cheirality
  (csrc/pose.hip: Nister's solver, Sampson distance, OpenCV's documented parameters; host code like cv2's).  PARITY
  UNPINNED against OpenCV (absent from this image; its sampling sequence cannot be reproduced): tests/test_pose.py
  checks the solver on exact data and the recovered pose on synthetic scenes with known ground truth.
"""
import os

import numpy as np
import torch

from . import ops


# ---- per-batch metrics ------------------------------------------------------------------------------------
def compute_symmetrical_epipolar_errors(data):
    """metrics.py:50-68.  Update: data['epi_errs'] float32 [M] (device tensor)."""
    data.update({"epi_errs": ops.epipolar_errors(data["mkpts0_f"], data["mkpts1_f"], data["m_bids"],
                                                 data["T_0to1"].to(torch.float32), data["K0"].to(torch.float32),
                                                 data["K1"].to(torch.float32))})


def _angle_deg(cosine):
    return float(np.degrees(np.arccos(np.clip(cosine, -1.0, 1.0))))


def relative_pose_error(T_0to1, R, t, ignore_gt_t_thr=0.0):
    """metrics.py:12-28 -> (t_err, R_err) in degrees: angle between the translation directions (sign-ambiguous, so
    folded to [0, 90]; ignored when the ground-truth baseline is shorter than `ignore_gt_t_thr`) and the geodesic
    angle between the rotations."""
    gt_R, gt_t = T_0to1[:3, :3], T_0to1[:3, 3]
    baseline = np.linalg.norm(gt_t)
    if baseline < ignore_gt_t_thr:
        t_err = 0
    else:
        ang = _angle_deg(np.dot(t, gt_t) / (np.linalg.norm(t) * baseline))
        t_err = min(ang, 180.0 - ang)
    R_err = abs(_angle_deg((np.trace(R.T @ gt_R) - 1.0) / 2.0))
    return t_err, R_err


def _normalise(kpts, K):
    """pixels -> normalised camera coordinates (the reference indexes K[[0,1],[2,2]] / K[[0,1],[0,1]], metrics.py:75-76)."""
    return (kpts - K[:2, 2][None]) / np.array([K[0, 0], K[1, 1]])[None]


def estimate_pose_cv2(kpts0, kpts1, K0, K1, thresh, conf=0.99999):
    """metrics.py:71-100: OpenCV 5-point RANSAC on normalised points (threshold = pixels / mean focal length as the
    reference computes it), then the cheirality vote over the returned essential matrices.  Needs cv2."""
    import cv2
    if len(kpts0) < 5:
        return None
    n0, n1 = _normalise(kpts0, K0), _normalise(kpts1, K1)
    focal = np.mean([K0[0, 0], K1[1, 1], K0[0, 0], K1[1, 1]])          # (sic) the reference's choice of entries
    E, mask = cv2.findEssentialMat(n0, n1, np.eye(3), threshold=thresh / focal, prob=conf, method=cv2.RANSAC)
    if E is None:
        return None
    candidates = []
    for Ei in np.split(E, len(E) // 3):
        votes, R, t, _ = cv2.recoverPose(Ei, n0, n1, np.eye(3), 1e9, mask=mask)
        candidates.append((votes, R, t[:, 0]))
    votes, R, t = max(candidates, key=lambda c: c[0], default=(0, None, None))   # first maximum, like the reference's `>`
    return None if votes <= 0 else (R, t, mask.ravel() > 0)


def _cfg_get(config, path, default):
    node = config
    for key in path:
        if node is None:
            return default
        node = node.get(key) if isinstance(node, dict) else getattr(node, key, None)
    return default if node is None else node


def estimate_pose_native(kpts0, kpts1, K0, K1, thresh, conf=0.99999, seed=0):
    """estimate_pose (metrics.py:72-98) on the library's own five-point RANSAC + cheirality (csrc/pose.hip, host code;
    parity against cv2 unpinned).  Returns (R [3,3], t [3], inlier mask [M] bool) or None like the reference."""
    import ctypes as C
    from . import _lib
    k0 = np.ascontiguousarray(kpts0, np.float32).reshape(-1, 2)
    k1 = np.ascontiguousarray(kpts1, np.float32).reshape(-1, 2)
    M = k0.shape[0]
    if M < 5:
        return None
    K0c, K1c = np.ascontiguousarray(K0, np.float32), np.ascontiguousarray(K1, np.float32)
    R, t = np.empty((3, 3), np.float32), np.empty(3, np.float32)
    inl = np.zeros(M, np.uint8)
    n = C.c_long(-1)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(_lib.load().loftr_estimate_pose(ptr(k0), ptr(k1), M, ptr(K0c), ptr(K1c), float(thresh), float(conf), int(seed),
                                               ptr(R), ptr(t), ptr(inl), C.byref(n)), "loftr_estimate_pose")
    if n.value < 0:
        return None
    return R.astype(np.float64), t.astype(np.float64), inl.astype(bool)


_WARNED_NATIVE = []


def compute_pose_errors(data, config=None, estimator=None, on_missing="raise"):
    """metrics.py:103-140.  Update: data['R_errs'], ['t_errs'] (lists of float), ['inliers'] (list of bool arrays) and
    data['pose_estimator'] (which estimator produced them: the AUCs depend on it).

    estimator(kpts0, kpts1, K0, K1, pixel_thr, conf=) -> (R, t, inlier_mask) | None.  Default: OpenCV, as the reference
    uses.  When OpenCV is not importable, `on_missing` decides -- never silently:
      'raise'  (default) ImportError, like the reference's `import cv2`;
      'inf'    record R_err = t_err = inf for every pair (the reference's value for a failed estimate);
      'native' the library's five-point RANSAC (csrc/pose.hip).  PARITY UNPINNED against cv2.findEssentialMat /
               recoverPose (own sampling sequence): a one-time warning says so; passing estimator=estimate_pose_native is
               the explicit form of the same opt-in."""
    pixel_thr = _cfg_get(config, ("TRAINER", "RANSAC_PIXEL_THR"), 0.5)
    conf = _cfg_get(config, ("TRAINER", "RANSAC_CONF"), 0.99999)
    if on_missing not in ("raise", "inf", "native"):
        raise ValueError(f"on_missing={on_missing!r}: expected 'raise', 'inf' or 'native'")
    name = getattr(estimator, "__name__", "custom") if estimator is not None else None
    if estimator is None:
        try:
            import cv2  # noqa: F401
            estimator, name = estimate_pose_cv2, "cv2"
        except ImportError:
            if on_missing == "raise":
                raise ImportError("compute_pose_errors needs OpenCV (cv2.findEssentialMat / recoverPose, metrics.py:72-98); pass "
                                  "on_missing='native' (library five-point RANSAC, parity unpinned) or 'inf', or an estimator")
            if on_missing == "native":
                estimator, name = estimate_pose_native, "estimate_pose_native"
                if not _WARNED_NATIVE:
                    _WARNED_NATIVE.append(True)
                    import warnings
                    warnings.warn("OpenCV is not importable: pose errors / AUC come from loftr_amd's five-point RANSAC, whose "
                                  "parity with cv2.findEssentialMat(RANSAC) + recoverPose is unpinned", stacklevel=2)
            else:
                name = "none (inf)"
    data["pose_estimator"] = name
    data.update({"R_errs": [], "t_errs": [], "inliers": []})
    m_bids = data["m_bids"].cpu().numpy()
    pts0, pts1 = data["mkpts0_f"].cpu().numpy(), data["mkpts1_f"].cpu().numpy()
    K0, K1, T = data["K0"].cpu().numpy(), data["K1"].cpu().numpy(), data["T_0to1"].cpu().numpy()
    for bs in range(K0.shape[0]):
        mask = m_bids == bs
        ret = None if estimator is None else estimator(pts0[mask], pts1[mask], K0[bs], K1[bs], pixel_thr, conf=conf)
        if ret is None:
            data["R_errs"].append(np.inf)
            data["t_errs"].append(np.inf)
            data["inliers"].append(np.array([]).astype(bool))
        else:
            R, t, inliers = ret
            t_err, R_err = relative_pose_error(T[bs], R, t, ignore_gt_t_thr=0.0)
            data["R_errs"].append(R_err)
            data["t_errs"].append(t_err)
            data["inliers"].append(inliers)


# ---- dataset-level aggregation (host, once per dataset) ----------------------------------------------------
def _area_under_recall(sorted_errors, recall, thr):
    """Area (trapezoid rule) under the recall-vs-error step curve from 0 to `thr`, the curve held flat from the last
    error below `thr` up to `thr`."""
    k = int(np.searchsorted(sorted_errors, thr))
    x = np.append(sorted_errors[:k], thr)
    y = np.append(recall[:k], recall[k - 1])
    return float(np.sum(np.diff(x) * (y[1:] + y[:-1]) * 0.5))


def error_auc(errors, thresholds=(5, 10, 20)):
    """metrics.py:143-160: normalised AUC of the cumulative pose-error curve at 5 / 10 / 20 degrees (the reference
    discards its `thresholds` argument in favour of these three)."""
    e = np.concatenate([[0.0], np.sort(np.asarray(list(errors), dtype=np.float64))])
    recall = np.linspace(0, 1, len(e))
    return {f"auc@{t}": _area_under_recall(e, recall, t) / t for t in (5, 10, 20)}


def epidist_prec(errors, thresholds, ret_dict=False):
    """metrics.py:163-174: for each threshold, the mean over pairs of the fraction of that pair's matches whose
    epipolar error is below it (a pair without matches counts 0)."""
    def pair_precision(errs, thr):
        errs = np.asarray(errs)
        return float(np.count_nonzero(errs < thr)) / errs.size if errs.size else 0

    precs = [np.mean([pair_precision(e, thr) for e in errors]) if len(errors) else 0 for thr in thresholds]
    return {f"prec@{t:.0e}": p for t, p in zip(thresholds, precs)} if ret_dict else precs


def aggregate_metrics(metrics, epi_err_thr=5e-4):
    """metrics.py:177-198: one entry per identifier (a DistributedSampler pads the last batch with repeats; a repeat
    overrides the earlier entry but keeps its position), pose AUC of max(R_err, t_err), matching precision at
    `epi_err_thr` (5e-4 ScanNet, 1e-4 MegaDepth)."""
    index_of = {}
    for i, ident in enumerate(metrics["identifiers"]):
        index_of[ident] = i
    keep = list(index_of.values())
    worst = np.maximum(np.asarray(metrics["R_errs"], dtype=np.float64), np.asarray(metrics["t_errs"], dtype=np.float64))[keep]
    out = error_auc(worst)
    out.update(epidist_prec([metrics["epi_errs"][i] for i in keep], [epi_err_thr], ret_dict=True))
    return out


# ---- the loop ----------------------------------------------------------------------------------------------
def _pair_names(batch):
    names = batch.get("pair_names")
    bs = batch["image0"].size(0)
    if names is None:
        return [(f"pair{b}_0", f"pair{b}_1") for b in range(bs)]
    return list(zip(*names))


def compute_metrics(batch, config=None, estimator=None, on_missing="raise"):
    """PL_LoFTR._compute_metrics (lightning_loftr.py:95-111) -> ({'metrics': {...}}, rel_pair_names)."""
    compute_symmetrical_epipolar_errors(batch)
    compute_pose_errors(batch, config, estimator=estimator, on_missing=on_missing)
    rel_pair_names = _pair_names(batch)
    bs = batch["image0"].size(0)
    epi, bids = batch["epi_errs"].cpu().numpy(), batch["m_bids"].cpu().numpy()      # one device->host copy, not one per pair
    metrics = {"identifiers": ["#".join(rel_pair_names[b]) for b in range(bs)],
               "epi_errs": [epi[bids == b] for b in range(bs)],
               "R_errs": batch["R_errs"], "t_errs": batch["t_errs"], "inliers": batch["inliers"]}
    return {"metrics": metrics}, rel_pair_names


@torch.no_grad()
def test_step(matcher, batch, config=None, dump=True, estimator=None, on_missing="raise"):
    """PL_LoFTR.test_step (lightning_loftr.py:205-229): matcher forward, metrics, optional per-pair dumps."""
    matcher(batch)
    ret_dict, rel_pair_names = compute_metrics(batch, config, estimator=estimator, on_missing=on_missing)
    if dump:
        pair_names = _pair_names(batch)
        bids = batch["m_bids"].cpu().numpy()
        host = {k: batch[k].cpu().numpy() for k in ("mkpts0_f", "mkpts1_f", "mconf", "epi_errs")}
        dumps = []
        for b in range(batch["image0"].shape[0]):
            mask = bids == b
            item = {"pair_names": pair_names[b], "identifier": "#".join(rel_pair_names[b])}
            for k, v in host.items():
                item[k] = v[mask]
            for k in ("R_errs", "t_errs", "inliers"):
                item[k] = batch[k][b]
            dumps.append(item)
        ret_dict["dumps"] = dumps
    return ret_dict


def gather(items):
    """src/utils/comm.py:gather as test_epoch_end uses it: the per-rank python lists concatenated in rank order on
    every rank (one process per GPU; metric lists are small host objects -> all_gather_object over the default
    group, RCCL-free: gloo or the object path of nccl).  Identity without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(items)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, list(items))
    return [x for part in parts for x in part]


def test_epoch_end(outputs, config=None, dump_dir=None):
    """PL_LoFTR.test_epoch_end (lightning_loftr.py:231-249): flatten the per-step metrics, gather them over the
    ranks (duplicates padded in by a DistributedSampler are dropped by identifier in aggregate_metrics), aggregate;
    rank 0 optionally saves ``LoFTR_pred_eval.npy``.  Every rank returns the aggregated metrics."""
    import torch.distributed as dist
    keys = outputs[0]["metrics"].keys()
    metrics = {k: gather([x for o in outputs for x in o["metrics"][k]]) for k in keys}
    epi_thr = _cfg_get(config, ("TRAINER", "EPI_ERR_THR"), 5e-4)
    result = aggregate_metrics(metrics, epi_thr)
    if dump_dir is not None:
        dumps = gather([d for o in outputs for d in o.get("dumps", [])])
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if rank == 0:
            os.makedirs(dump_dir, exist_ok=True)
            np.save(os.path.join(dump_dir, "LoFTR_pred_eval"), np.array(dumps, dtype=object), allow_pickle=True)
    return result
