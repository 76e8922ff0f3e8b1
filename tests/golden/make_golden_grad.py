"""Goldens of the heads' backward passes: gradients of the reference's OWN modules under torch.autograd.

    python tests/golden/make_golden_grad.py       # authoring container only (needs /root/reference)

Chain (src/lightning/lightning_loftr.py:112-133 back-propagates batch['loss']):
    feat_c0, feat_c1 -> CoarseMatching.forward (coarse_matching.py:105-119) -> conf_matrix -> LoFTRLoss.compute_coarse_loss
    feat_f0, feat_f1 -> FineMatching.forward   (fine_matching.py:43-57)     -> expec_f     -> LoFTRLoss.compute_fine_loss
The inputs are regenerated from the seeds by `build_inputs` (tests use the same function); the npz stores the reference's
forward values (loss_c, loss_f, conf at the ground truth, expec_f) and the four input gradients (+ d loss_c / d conf and
d loss_f / d expec_f, the intermediate nodes)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # sparse focal supervision (the shipped training config: src/config/default.py sparse_spvs True), l2_with_std
    "grad_ds": dict(seed=11, N=2, hc=(6, 8), masks=False, coarse_type="focal", sparse=True, fine_type="l2_with_std", M=40, n_gt=30),
    # MegaDepth-style padding masks
    "grad_ds_mask": dict(seed=12, N=2, hc=(6, 8), masks=True, coarse_type="focal", sparse=True, fine_type="l2_with_std", M=24, n_gt=20),
    # dense focal (negatives supervised), masks as loss weights
    "grad_dense_mask": dict(seed=13, N=2, hc=(6, 8), masks=True, coarse_type="focal", sparse=False, fine_type="l2", M=24, n_gt=20),
    # dense cross-entropy, plain l2
    "grad_ce": dict(seed=14, N=1, hc=(7, 9), masks=False, coarse_type="cross_entropy", sparse=False, fine_type="l2", M=16, n_gt=25),
    # no ground truth and no correct fine match (.train(): loftr_loss.py:31-36, :113-117, :138-143)
    "grad_nogt": dict(seed=15, N=1, hc=(6, 8), masks=False, coarse_type="focal", sparse=False, fine_type="l2", M=12, n_gt=0),
    "grad_nogt_std": dict(seed=16, N=1, hc=(6, 8), masks=False, coarse_type="focal", sparse=True, fine_type="l2_with_std", M=12, n_gt=0),
    # Sinkhorn head (configs/loftr/*/loftr_ot*.py): sparse focal on conf_matrix_with_bin incl. the dustbin negatives; bin_score is a parameter
    "grad_ot": dict(seed=17, N=2, hc=(6, 8), masks=False, coarse_type="focal", sparse=True, fine_type="l2_with_std", M=12, n_gt=22,
                    match_type="sinkhorn"),
    "grad_ot_mask": dict(seed=18, N=2, hc=(6, 8), masks=True, coarse_type="focal", sparse=True, fine_type="l2_with_std", M=12, n_gt=18,
                         match_type="sinkhorn"),
    "grad_ot_nogt": dict(seed=19, N=1, hc=(6, 8), masks=False, coarse_type="focal", sparse=True, fine_type="l2_with_std", M=12, n_gt=0,
                         match_type="sinkhorn"),
}
BIN_SCORE, SKH_ITERS = 1.0, 3
TEMPERATURE, C_COARSE, C_FINE, WW = 0.1, 256, 128, 25


def build_inputs(rc):
    """Seeded head inputs: correlated coarse descriptors (peaked but not saturated conf at the ground truth), fine windows."""
    rng = np.random.default_rng(rc["seed"])
    N, (h, w) = rc["N"], rc["hc"]
    L = S = h * w
    ot = rc.get("match_type") == "sinkhorn"                                 # no temperature there: sim = <f0, f1> / C
    a, c = (4.0, 0.45) if ot else (1.5, 0.2)
    f0 = (a * rng.standard_normal((N, L, C_COARSE))).astype(np.float32)
    perm = np.stack([rng.permutation(L) for _ in range(N)])
    f1 = np.empty_like(f0)
    for n in range(N):
        f1[n, perm[n]] = c * f0[n] + (a * rng.standard_normal((L, C_COARSE))).astype(np.float32)
    mask0 = mask1 = None
    if rc["masks"]:
        mask0, mask1 = np.ones((N, h, w), bool), np.ones((N, h, w), bool)
        mask0[0, h - 2:, :], mask1[0, :, w - 2:] = False, False
        if N > 1:
            mask0[1, :, w - 3:], mask1[1, h - 1:, :] = False, False
    conf_gt = np.zeros((N, L, S), np.float32)
    for n in range(N):
        cand = np.arange(1, L)                                              # cell 0 is never supervised (supervision.py:76-81)
        if mask0 is not None:
            ok = mask0[n].reshape(-1)[cand] & mask1[n].reshape(-1)[perm[n][cand]]
            cand = cand[ok]
        pick = rng.choice(cand, size=min(rc["n_gt"], len(cand)), replace=False) if rc["n_gt"] else []
        for i in pick:
            conf_gt[n, i, perm[n][i]] = 1
    M = rc["M"]
    ff0 = rng.standard_normal((M, WW, C_FINE)).astype(np.float32)
    ff1 = (0.6 * ff0[:, WW // 2:WW // 2 + 1, :] * (rng.random((M, WW, 1)) < 0.2) + rng.standard_normal((M, WW, C_FINE))).astype(np.float32)
    if rc["n_gt"]:
        gt = rng.uniform(-1.4, 1.4, (M, 2)).astype(np.float32)              # some beyond fine_correct_thr = 1
    else:
        gt = (1.5 + rng.random((M, 2))).astype(np.float32)                  # none correct
    return dict(feat_c0=f0, feat_c1=f1, mask0=mask0, mask1=mask1, conf_gt=conf_gt, feat_f0=ff0, feat_f1=ff1, expec_f_gt=gt)


def loss_cfg(rc):
    return {"loftr": {"loss": dict(coarse_type=rc["coarse_type"], coarse_weight=1.0, focal_alpha=0.25, focal_gamma=2.0, pos_weight=1.0,
                                   neg_weight=1.0, fine_type=rc["fine_type"], fine_weight=1.0, fine_correct_thr=1.0),
                      "match_coarse": dict(match_type=rc.get("match_type", "dual_softmax"), sparse_spvs=rc["sparse"])}}


def matcher_cfg(rc):
    return dict(thr=0.2, border_rm=2, match_type=rc.get("match_type", "dual_softmax"), dsmax_temperature=TEMPERATURE, train_coarse_percent=0.4,
                train_pad_num_gt_min=200, sparse_spvs=rc["sparse"], skh_init_bin_score=BIN_SCORE, skh_iters=SKH_ITERS, skh_prefilter=False)


def make(name):
    import importlib
    import torch
    from oracle.ref_shim import import_reference, import_reference_training
    import_reference()
    _, RefLoss = import_reference_training()
    RefCoarse = importlib.import_module("src.loftr.utils.coarse_matching").CoarseMatching
    RefFine = importlib.import_module("src.loftr.utils.fine_matching").FineMatching
    rc = CASES[name]
    inp = build_inputs(rc)
    h, w = rc["hc"]
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
    leaf = {k: t(inp[k]).requires_grad_(True) for k in ("feat_c0", "feat_c1", "feat_f0", "feat_f1")}
    data = {"hw0_c": (h, w), "hw1_c": (h, w), "hw0_i": (8 * h, 8 * w), "hw1_i": (8 * h, 8 * w), "hw0_f": (4 * h, 4 * w), "hw1_f": (4 * h, 4 * w)}
    m0 = m1 = None
    if rc["masks"]:
        data.update(mask0=t(inp["mask0"]), mask1=t(inp["mask1"]))
        m0, m1 = data["mask0"].flatten(-2), data["mask1"].flatten(-2)
    cm = RefCoarse(matcher_cfg(rc)).eval()                  # eval: no sampling; conf_matrix is built the same way (:105-119)
    cm(leaf["feat_c0"], leaf["feat_c1"], data, mask_c0=m0, mask_c1=m1)
    ot = rc.get("match_type") == "sinkhorn"
    conf = data["conf_matrix_with_bin"] if ot else data["conf_matrix"]          # the tensor the loss reads (loftr_loss.py:174-177)
    if ot:                                                  # .clone() of assign_matrix (:143): re-derive it WITH the graph
        log_assign = cm.log_optimal_transport(torch.einsum("nlc,nsc->nls", leaf["feat_c0"] / 16, leaf["feat_c1"] / 16) if m0 is None else
                                              torch.einsum("nlc,nsc->nls", leaf["feat_c0"] / 16, leaf["feat_c1"] / 16).masked_fill(
                                                  ~(m0[..., None] * m1[:, None]).bool(), -1e9), cm.bin_score, cm.skh_iters)
        assert torch.equal(log_assign.exp().detach(), conf.detach())
        conf = log_assign.exp()
    conf.retain_grad()
    M = rc["M"]
    data.update(mkpts0_c=torch.zeros(M, 2), mkpts1_c=torch.zeros(M, 2), mconf=torch.zeros(M), b_ids=torch.zeros(M, dtype=torch.long))
    RefFine().train()(leaf["feat_f0"], leaf["feat_f1"], data)
    expec = data["expec_f"]
    expec.retain_grad()
    loss = RefLoss(loss_cfg(rc)).train()
    weight = loss.compute_c_weight(data)
    loss_c = loss.compute_coarse_loss(conf, t(inp["conf_gt"]), weight=weight)
    # plain l2 takes [M, 2] in the reference (:113-120 subtracts expec_f as a whole): the (x, y) columns of FineMatching's [M, 3]
    loss_f = loss.compute_fine_loss(expec if rc["fine_type"] == "l2_with_std" else expec[:, :2], t(inp["expec_f_gt"]))
    (loss_c + loss_f).backward()
    z = lambda v, like: (v.grad if v.grad is not None else torch.zeros_like(like)).numpy()
    b, i, j = np.nonzero(inp["conf_gt"])
    extra = {"grad_bin_score": np.float64(cm.bin_score.grad)} if ot else {}
    store = dict(recipe=np.array(json.dumps(rc)), loss_c=float(loss_c.detach()), loss_f=float(loss_f.detach()), **extra,
                 conf_at_gt=conf.detach().numpy()[b, i, j], expec_f=expec.detach().numpy(),
                 grad_conf=z(conf, conf), grad_expec=z(expec, expec),
                 **{f"grad_{k}": z(v, v) for k, v in leaf.items()})
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **store)
    print(name, "loss_c %.5f loss_f %.5f" % (float(loss_c), float(loss_f)), "conf@gt", store["conf_at_gt"][:4],
          {k: float(np.abs(v).max()) for k, v in store.items() if k.startswith("grad_")})


if __name__ == "__main__":
    for nm in sys.argv[1:] or list(CASES):
        make(nm)
