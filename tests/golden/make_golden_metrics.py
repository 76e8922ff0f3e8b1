"""Golden vectors for the evaluation caller, produced by the REAL reference's
src/utils/metrics.py (imported through oracle/ref_shim.py).  Authoring container only:

    python tests/golden/make_golden_metrics.py        ->  tests/golden/metrics_epi.npz, metrics_agg.npz

Inputs are synthetic and seeded (no dataset / checkpoint here): a random two-view geometry per pair
(intrinsics, relative pose), matches = projections of random 3-D points with pixel noise plus gross outliers,
ragged per-pair counts including an empty pair, matches grouped by ascending pair id like the matcher emits them.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.ref_shim import import_reference_metrics   # noqa: E402
from _scenes import make_scene, random_rotation        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = import_reference_metrics()
    if not hasattr(np, "trapz"):                       # numpy >= 2.4 dropped the alias the reference calls
        np.trapz = np.trapezoid
    # ---- per-match epipolar errors (metrics.py:31-68) ----------------------------------------------
    out = {}
    for name, seed, counts in (("a", 101, [257, 0, 64, 1000]), ("b", 102, [5, 3]), ("c", 103, [0, 0, 33])):
        sc = make_scene(seed, counts)
        data = {k: torch.from_numpy(v) for k, v in sc.items()}
        ref.compute_symmetrical_epipolar_errors(data)
        for k, v in sc.items():
            out[f"{name}_{k}"] = v
        out[f"{name}_epi_errs"] = data["epi_errs"].numpy()
    # ---- relative_pose_error (metrics.py:12-28) ---------------------------------------------------
    rng = np.random.default_rng(7)
    T = np.stack([make_scene(200 + i, [1])["T_0to1"][0] for i in range(6)]).astype(np.float64)
    Rs = np.stack([random_rotation(rng, 40) for _ in range(6)])
    ts = rng.normal(size=(6, 3))
    ts[1] = -T[1, :3, 3] * 2.5                          # opposite direction: E ambiguity branch (180 - err)
    Rs[2] = T[2, :3, :3]                                # exact rotation: cos clipped at 1
    errs = np.array([ref.relative_pose_error(T[i], Rs[i], ts[i], ignore_gt_t_thr=0.0) for i in range(6)])
    errs_thr = np.array([ref.relative_pose_error(T[i], Rs[i], ts[i], ignore_gt_t_thr=10.0) for i in range(6)], dtype=np.float64)
    out.update(rpe_T=T, rpe_R=Rs, rpe_t=ts, rpe_errs=errs, rpe_errs_thr=errs_thr)
    np.savez_compressed(os.path.join(HERE, "metrics_epi.npz"), **out)

    # ---- aggregation (metrics.py:143-198) ---------------------------------------------------------
    rng = np.random.default_rng(11)
    n = 40
    ids = [f"scene{i % 31}#img{i % 31}" for i in range(n)]               # 9 duplicated identifiers (DistributedSampler padding)
    R_errs = list(rng.gamma(1.5, 4.0, n)); t_errs = list(rng.gamma(1.5, 6.0, n))
    R_errs[3] = np.inf; t_errs[3] = np.inf                                # a failed pose (metrics.py:128-131)
    epi = [rng.gamma(0.6, 4e-4, int(rng.integers(0, 300))).astype(np.float32) for _ in range(n)]
    epi[5] = np.zeros((0,), np.float32)                                   # a pair without matches
    metrics = dict(identifiers=ids, R_errs=R_errs, t_errs=t_errs, epi_errs=epi)
    agg = {f"thr_{thr:g}": ref.aggregate_metrics(dict(metrics), thr) for thr in (5e-4, 1e-4)}
    auc_errs = np.random.default_rng(11 + 1).gamma(2.0, 5.0, 200)
    auc_only = ref.error_auc(auc_errs, [5, 10, 20])
    prec = ref.epidist_prec(epi, [1e-4, 5e-4, 1e-3], False)
    flat = dict(ids=np.array(ids), R_errs=np.array(R_errs), t_errs=np.array(t_errs),
                epi_lens=np.array([len(e) for e in epi]), epi_flat=np.concatenate(epi), auc_errs=auc_errs,
                prec=np.array(prec, dtype=np.float64))
    for k, d in agg.items():
        for kk, v in d.items():
            flat[f"agg_{k}_{kk}"] = np.float64(v)
    for kk, v in auc_only.items():
        flat[f"auc_{kk}"] = np.float64(v)
    np.savez_compressed(os.path.join(HERE, "metrics_agg.npz"), **flat)
    print("wrote metrics_epi.npz, metrics_agg.npz;", {k: float(v) for k, v in agg["thr_0.0005"].items()})


if __name__ == "__main__":
    main()
