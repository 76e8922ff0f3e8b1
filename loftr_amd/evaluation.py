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
* pose estimation (metrics.py:71-140) is OpenCV (`cv2.findEssentialMat` RANSAC + `cv2.recoverPose`): used when
  cv2 is importable, otherwise `compute_pose_errors` raises unless an `estimator` is supplied or
  ``on_missing="inf"`` asks for the reference's own failure values (R_err = t_err = inf, no inliers,
  metrics.py:128-131).  Not re-implemented here: without the library its RANSAC cannot be pinned.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops


# ---- per-batch metrics ------------------------------------------------------------------------------------
def compute_symmetrical_epipolar_errors(data):
    """metrics.py:50-68.  Update: data['epi_errs'] float32 [M] (device tensor)."""
    data.update({"epi_errs": ops.epipolar_errors(data["mkpts0_f"], data["mkpts1_f"], data["m_bids"],
                                                 data["T_0to1"].to(torch.float32), data["K0"].to(torch.float32),
                                                 data["K1"].to(torch.float32))})


def relative_pose_error(T_0to1, R, t, ignore_gt_t_thr=0.0):
    """metrics.py:12-28: (t_err, R_err) in degrees."""
    t_gt = T_0to1[:3, 3]
    n = np.linalg.norm(t) * np.linalg.norm(t_gt)
    t_err = np.rad2deg(np.arccos(np.clip(np.dot(t, t_gt) / n, -1.0, 1.0)))
    t_err = np.minimum(t_err, 180 - t_err)              # E ambiguity
    if np.linalg.norm(t_gt) < ignore_gt_t_thr:          # pure rotation
        t_err = 0
    R_gt = T_0to1[:3, :3]
    cos = np.clip((np.trace(np.dot(R.T, R_gt)) - 1) / 2, -1.0, 1.0)
    R_err = np.rad2deg(np.abs(np.arccos(cos)))
    return t_err, R_err


def estimate_pose_cv2(kpts0, kpts1, K0, K1, thresh, conf=0.99999):
    """metrics.py:71-100 verbatim in behaviour: needs OpenCV."""
    import cv2
    if len(kpts0) < 5:
        return None
    kpts0 = (kpts0 - K0[[0, 1], [2, 2]][None]) / K0[[0, 1], [0, 1]][None]
    kpts1 = (kpts1 - K1[[0, 1], [2, 2]][None]) / K1[[0, 1], [0, 1]][None]
    ransac_thr = thresh / np.mean([K0[0, 0], K1[1, 1], K0[0, 0], K1[1, 1]])
    E, mask = cv2.findEssentialMat(kpts0, kpts1, np.eye(3), threshold=ransac_thr, prob=conf, method=cv2.RANSAC)
    if E is None:
        return None
    best, ret = 0, None
    for _E in np.split(E, len(E) / 3):
        n, R, t, _ = cv2.recoverPose(_E, kpts0, kpts1, np.eye(3), 1e9, mask=mask)
        if n > best:
            ret, best = (R, t[:, 0], mask.ravel() > 0), n
    return ret


def _cfg_get(config, path, default):
    node = config
    for key in path:
        if node is None:
            return default
        node = node.get(key) if isinstance(node, dict) else getattr(node, key, None)
    return default if node is None else node


def compute_pose_errors(data, config=None, estimator=None, on_missing="raise"):
    """metrics.py:103-140.  Update: data['R_errs'], ['t_errs'] (lists of float), ['inliers'] (list of bool arrays).

    estimator(kpts0, kpts1, K0, K1, pixel_thr, conf=) -> (R, t, inlier_mask) | None; default: OpenCV as in the
    reference.  on_missing: 'raise' (default) or 'inf' = record the reference's failure values when no estimator
    is available."""
    pixel_thr = _cfg_get(config, ("TRAINER", "RANSAC_PIXEL_THR"), 0.5)
    conf = _cfg_get(config, ("TRAINER", "RANSAC_CONF"), 0.99999)
    if estimator is None:
        try:
            import cv2  # noqa: F401
            estimator = estimate_pose_cv2
        except ImportError:
            if on_missing != "inf":
                raise ImportError("compute_pose_errors needs OpenCV (cv2.findEssentialMat / recoverPose, as the reference) "
                                  "or an explicit estimator=; pass on_missing='inf' to record failed poses instead")
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
def error_auc(errors, thresholds=(5, 10, 20)):
    """metrics.py:143-160 (which ignores its `thresholds` argument in favour of [5, 10, 20])."""
    thresholds = [5, 10, 20]
    errors = [0] + sorted(list(errors))
    recall = list(np.linspace(0, 1, len(errors)))
    aucs = []
    for thr in thresholds:
        last = int(np.searchsorted(errors, thr))
        y = np.asarray(recall[:last] + [recall[last - 1]], dtype=np.float64)
        x = np.asarray(errors[:last] + [thr], dtype=np.float64)
        aucs.append(float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) / 2.0)) / thr)
    return {f"auc@{t}": auc for t, auc in zip(thresholds, aucs)}


def epidist_prec(errors, thresholds, ret_dict=False):
    """metrics.py:163-174."""
    precs = []
    for thr in thresholds:
        per_pair = [np.mean(np.asarray(e) < thr) if len(e) > 0 else 0 for e in errors]
        precs.append(np.mean(per_pair) if len(per_pair) > 0 else 0)
    if ret_dict:
        return {f"prec@{t:.0e}": p for t, p in zip(thresholds, precs)}
    return precs


def aggregate_metrics(metrics, epi_err_thr=5e-4):
    """metrics.py:177-198: drop the duplicates a DistributedSampler pads with, pose AUC @5/10/20 deg of
    max(R_err, t_err), mean matching precision at `epi_err_thr` (5e-4 ScanNet, 1e-4 MegaDepth)."""
    unq_ids = list(OrderedDict((iden, i) for i, iden in enumerate(metrics["identifiers"])).values())
    pose_errors = np.max(np.stack([metrics["R_errs"], metrics["t_errs"]]), axis=0)[unq_ids]
    aucs = error_auc(pose_errors, [5, 10, 20])
    epi = [metrics["epi_errs"][i] for i in unq_ids]
    return {**aucs, **epidist_prec(epi, [epi_err_thr], True)}


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
