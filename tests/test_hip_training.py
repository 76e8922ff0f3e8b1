"""GPU parity of the training-side consumers (forward values): loftr_amd.training through the
C-ABI (csrc/train.hip) against the goldens produced by the reference's own spvs_coarse / spvs_fine / LoFTRLoss
(tests/golden/train_*.npz, generator: make_golden_train.py) and the numpy oracle for the corner cases."""
import json

import numpy as np
import pytest
import torch

from test_train_oracle import CASES, LOSS_CFGS, MG, check_spvs, load

pytestmark = pytest.mark.gpu
CFG = {"LOFTR": {"RESOLUTION": (8, 2), "FINE_WINDOW_SIZE": 5}}


def batch(rc, inp, dev):
    N, H, W = rc["N"], rc["H"], rc["W"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return {"image0": torch.zeros(N, 1, H, W, device=dev), "image1": torch.zeros(N, 1, H, W, device=dev),
            "dataset_name": ["scannet"] * N, **{k: t(v) for k, v in inp.items()}}


def loss_cfg(ctype, sparse, mtype, ftype):
    return {"loftr": {"loss": dict(coarse_type=ctype, coarse_weight=1.0, focal_alpha=0.25, focal_gamma=2.0, pos_weight=1.0, neg_weight=1.0,
                                   fine_type=ftype, fine_weight=1.0, fine_correct_thr=1.0),
                      "match_coarse": dict(match_type=mtype, sparse_spvs=sparse)}}


@pytest.mark.parametrize("name", CASES)
def test_supervision_against_reference_golden(name):
    from loftr_amd.training import compute_supervision_coarse, compute_supervision_fine
    dev = torch.device("cuda", 0)
    rc, inp, g = load(name)
    data = batch(rc, inp, dev)
    compute_supervision_coarse(data, CFG)
    out = {k: data[k].cpu().numpy() for k in ("spv_b_ids", "spv_i_ids", "spv_j_ids", "spv_w_pt0_i", "spv_pt1_i")}
    check_spvs(out, g)
    key = out["spv_b_ids"] * 10**6 + out["spv_i_ids"]
    assert np.all(np.diff(key) > 0)                                    # torch.where order: ascending (b, i)
    gt = data["conf_matrix_gt"]
    if data["_spv_count"] == 0:                                        # placeholder ids (0, 0, 0), conf_matrix_gt all zero (supervision.py:94-99)
        assert gt.sum().item() == 0 and g["conf_gt_sum"].sum() == 0
    else:
        assert gt.sum().item() == len(out["spv_b_ids"]) and gt[data["spv_b_ids"], data["spv_i_ids"], data["spv_j_ids"]].min().item() == 1
    # fine supervision on the reference's own coarse supervision and predictions
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data.update(spv_w_pt0_i=t(g["spv_w_pt0_i"]), spv_pt1_i=t(g["spv_pt1_i"]), b_ids=t(g["b_ids"]), i_ids=t(g["i_ids"]), j_ids=t(g["j_ids"]))
    compute_supervision_fine(data, CFG)
    ef = data["expec_f_gt"].cpu().numpy()
    assert np.abs(ef - g["expec_f_gt"]).max() <= 1e-5 * max(1.0, np.abs(g["expec_f_gt"]).max())


@pytest.mark.parametrize("name", CASES)
def test_losses_against_reference_golden(name):
    from loftr_amd.training import LoFTRLoss
    dev = torch.device("cuda", 0)
    rc, inp, g = load(name)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    conf, conf_bin = MG.replay_matcher_outputs(rc, g["spv_b_ids"], g["spv_i_ids"], g["spv_j_ids"])
    data = {"conf_matrix": t(conf), "conf_matrix_with_bin": t(conf_bin), "expec_f_gt": t(g["expec_f_gt"]),
            **{k: t(g[k]) for k in ("spv_b_ids", "spv_i_ids", "spv_j_ids")}}
    if "mask0" in inp:
        data.update(mask0=t(inp["mask0"]), mask1=t(inp["mask1"]))
    want = json.loads(str(g["losses"]))
    for tag, (ctype, sparse, mtype, ftype) in LOSS_CFGS.items():
        d = dict(data)
        d["expec_f"] = t(g["expec_f"][:, :2].copy() if ftype == "l2" else g["expec_f"])
        LoFTRLoss(loss_cfg(ctype, sparse, mtype, ftype)).eval()(d)
        got = {k: float(v) for k, v in d["loss_scalars"].items()}
        for k in ("loss_c", "loss_f", "loss"):
            assert abs(got[k] - want[tag][k]) <= 2e-6 * max(1.0, abs(want[tag][k])), (tag, k, got, want[tag])
        assert d["loss"].is_cuda


def test_corner_cases_against_oracle():
    """No ground truth at all (supervision.py:94-99, loftr_loss.py:32-36), no correct fine entry (:113-117, :138-143),
    empty prediction list."""
    from oracle import train_oracle as T
    from loftr_amd.training import LoFTRLoss, compute_supervision_coarse, compute_supervision_fine
    dev = torch.device("cuda", 0)
    rc, inp, g = load("train_sc")
    inp = dict(inp)
    inp["depth0"] = np.zeros_like(inp["depth0"])                        # every cell of image 0 warps to the corner
    data = batch(rc, inp, dev)
    compute_supervision_coarse(data, CFG)
    assert data["_spv_count"] == 0 and data["spv_b_ids"].tolist() == [0] and data["spv_i_ids"].tolist() == [0]
    assert data["conf_matrix_gt"].sum().item() == 0
    N, L = rc["N"], (rc["H"] // 8) * (rc["W"] // 8)
    rng = np.random.default_rng(5)
    conf = rng.random((N, L, L)).astype(np.float32) ** 4
    gt0 = np.zeros((N, L, L), np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data.update(conf_matrix=t(conf), b_ids=torch.zeros(0, dtype=torch.int64, device=dev), i_ids=torch.zeros(0, dtype=torch.int64, device=dev),
                j_ids=torch.zeros(0, dtype=torch.int64, device=dev))
    compute_supervision_fine(data, CFG)
    assert tuple(data["expec_f_gt"].shape) == (0, 2)
    for ctype, sparse in (("focal", True), ("focal", False), ("cross_entropy", False)):
        crit = LoFTRLoss(loss_cfg(ctype, sparse, "dual_softmax", "l2_with_std")).eval()
        lc = float(crit.compute_coarse_loss(data["conf_matrix"], data))
        want = T.coarse_loss(conf, gt0, None, ctype, sparse, "dual_softmax")
        assert abs(lc - want) <= 2e-6 * max(1.0, abs(want)), (ctype, sparse, lc, want)
    # fine loss without a correct entry
    ef = t(np.array([[0.1, 0.2, 0.5], [0.3, -0.1, 0.7]], np.float32))
    egt = t(np.array([[1.5, 0.0], [0.2, -2.0]], np.float32))
    for ftype in ("l2", "l2_with_std"):
        crit = LoFTRLoss(loss_cfg("focal", True, "dual_softmax", ftype))
        assert crit.eval().compute_fine_loss(ef, egt) is None
        lf = float(crit.train().compute_fine_loss(ef, egt))
        assert abs(lf - (0.0 if ftype == "l2_with_std" else (1.4 ** 2 + 0.2 ** 2))) <= 1e-6


def test_rejects_cpu_tensors():
    from loftr_amd import _lib
    from loftr_amd.training import compute_supervision_coarse
    rc, inp, g = load("train_sc")
    with pytest.raises((_lib.LoftrHipError, RuntimeError)):
        compute_supervision_coarse(batch(rc, inp, torch.device("cpu")), CFG)


@pytest.mark.parametrize("name", list(MG.COARSE_TRAIN))
def test_coarse_matching_train_branch(name, monkeypatch):
    """CoarseMatching.train(): the sampled / ground-truth-padded fine-level training set of the reference module
    (coarse_matching.py:200-259), torch.randint replaced by the same deterministic sequence on both sides."""
    import os
    from _cases import GOLDEN_DIR
    from loftr_amd.loftr import CoarseMatching
    dev = torch.device("cuda", 0)
    g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))
    rc = json.loads(str(g["recipe"]))
    inp = MG.coarse_train_inputs(rc)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    H, W = rc["h"] * 8, rc["w"] * 8
    data = {"hw0_i": torch.Size([H, W]), "hw1_i": torch.Size([H, W]), "hw0_c": torch.Size([rc["h"], rc["w"]]),
            "hw1_c": torch.Size([rc["h"], rc["w"]]), **{k: t(v) for k, v in inp.items() if not k.startswith("feat")}}
    mod = CoarseMatching(MG.coarse_train_config(rc)).to(dev).train()
    monkeypatch.setattr(torch, "randint", MG.det_randint)
    m0 = data["mask0"].flatten(-2) if rc["masks"] else None
    m1 = data["mask1"].flatten(-2) if rc["masks"] else None
    mod(t(inp["feat_c0"]), t(inp["feat_c1"]), data, mask_c0=m0, mask_c1=m1)
    for k in ("b_ids", "i_ids", "j_ids", "gt_mask", "m_bids"):
        assert np.array_equal(data[k].cpu().numpy(), g[k]), k
    for k in ("mkpts0_c", "mkpts1_c"):
        assert np.array_equal(data[k].cpu().numpy(), g[k]), k
    # the contract's bar; these logits are O(200) (features x 4), where the log-sum-exp form rounds conf = 1 by up to 3e-5
    assert np.abs(data["mconf"].cpu().numpy() - g["mconf"]).max() <= 1e-4
    assert data["gt_mask"].sum().item() >= rc["pad_min"]


@pytest.mark.parametrize("name", list(MG.STEP_CASES))
def test_training_step_chain_against_reference(name, monkeypatch):
    """PL_LoFTR._trainval_inference (lightning_loftr.py:82-93) end to end: coarse supervision -> matcher in .train() mode
    (sampling / ground-truth padding, FineMatching's [:len(mconf)] slice) -> fine supervision -> losses, against the same
    chain of the reference (tests/golden/tstep_*.npz).  The backbone runs as the PyTorch mirror on the CPU in train mode
    (BatchNorm on batch statistics: the reference's own arithmetic, no MIOpen noise); everything after it is the HIP path."""
    import copy
    import importlib.util
    import os
    from _cases import GOLDEN_DIR
    from loftr_amd import LoFTR
    from loftr_amd.training import LoFTRLoss, compute_supervision_coarse, compute_supervision_fine
    spec = importlib.util.spec_from_file_location("make_golden_e2e", os.path.join(GOLDEN_DIR, "make_golden_e2e.py"))
    E2E = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(E2E)
    dev = torch.device("cuda", 0)
    g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))
    rc = json.loads(str(g["recipe"]))
    batch, geo = MG.step_batch(rc)
    N = geo["N"]
    cfg = MG.step_matcher_cfg(rc)
    cpu = LoFTR(copy.deepcopy(cfg))
    sd = E2E.e2e_state_dict(cpu, cfg, 0.3)
    cpu.load_state_dict(sd, strict=True)
    cpu.train()
    with torch.no_grad():
        fc, ff = cpu.backbone(torch.from_numpy(np.concatenate([batch["image0"], batch["image1"]], 0)))
    model = LoFTR(copy.deepcopy(cfg))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data = {"dataset_name": ["scannet"] * N, **{k: t(v) for k, v in batch.items()}}
    monkeypatch.setattr(torch, "randint", MG.det_randint)
    compute_supervision_coarse(data, CFG)
    data.update({"bs": N, "hw0_i": data["image0"].shape[2:], "hw1_i": data["image1"].shape[2:]})
    with torch.no_grad():
        model.match_from_features(fc[:N].to(dev), fc[N:].to(dev), ff[:N].to(dev), ff[N:].to(dev), data)
    compute_supervision_fine(data, CFG)
    LoFTRLoss(MG.step_loss_cfg(rc)).train()(data)
    for k in ("spv_b_ids", "spv_i_ids", "spv_j_ids", "b_ids", "i_ids", "j_ids", "gt_mask", "m_bids"):
        assert np.array_equal(data[k].cpu().numpy(), g[k]), k
    for k, tol in (("mconf", 1e-4), ("mkpts0_c", 0.0), ("mkpts1_c", 0.0), ("mkpts0_f", 1e-3), ("mkpts1_f", 1e-3), ("expec_f", 3e-4), ("expec_f_gt", 1e-5)):
        d = np.abs(data[k].cpu().numpy().astype(np.float64) - g[k]).max() if len(g[k]) else 0.0
        assert d <= tol * max(1.0, np.abs(g[k]).max() if k == "expec_f_gt" else 1.0), (k, d)
    want = json.loads(str(g["losses"]))
    got = {k: float(v) for k, v in data["loss_scalars"].items()}
    for k in ("loss_c", "loss_f", "loss"):
        assert abs(got[k] - want[k]) <= 2e-4 * max(1.0, abs(want[k])), (k, got, want)


def test_train_mode_forward_from_images_runs(monkeypatch):
    """LoFTR.forward in .train() mode straight from images: the backbone goes through PyTorch (BatchNorm on batch statistics),
    CoarseMatching samples / pads with the ground truth, the fine stage runs on the padded list.  MIOpen's arithmetic is not
    the reference's bit for bit, so this checks structure, not the golden ids (test_training_step_chain_* pins those)."""
    import copy
    from loftr_amd import LoFTR
    from loftr_amd.training import compute_supervision_coarse
    dev = torch.device("cuda", 0)
    rc = MG.STEP_CASES["tstep_ds"]
    batch, geo = MG.step_batch(rc)
    cfg = MG.step_matcher_cfg(rc)
    torch.manual_seed(0)
    model = LoFTR(copy.deepcopy(cfg)).to(dev).train()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data = {"dataset_name": ["scannet"] * geo["N"], **{k: t(v) for k, v in batch.items()}}
    monkeypatch.setattr(torch, "randint", MG.det_randint)
    compute_supervision_coarse(data, CFG)
    model(data)
    n_train = int(geo["N"] * (geo["H"] // 8) * (geo["W"] // 8) * rc["percent"])
    assert data["b_ids"].shape[0] == n_train == data["expec_f"].shape[0]
    assert int(data["gt_mask"].sum()) >= rc["pad_min"]
    n = int((~data["gt_mask"]).sum())
    assert data["mkpts0_f"].shape == data["mkpts1_f"].shape == (n, 2) == tuple(data["mkpts1_c"].shape)
    assert torch.isfinite(data["expec_f"]).all() and torch.isfinite(data["mkpts1_f"]).all()
    assert model.backbone.bn1.num_batches_tracked.item() == 1          # BatchNorm really ran in training mode


@pytest.mark.parametrize("name", list(MG.GRAD_STEP_CASES))
def test_training_step_head_gradients_against_reference(name, monkeypatch):
    """The training step of test_training_step_chain_against_reference WITH the heads' backward (LoFTR.head_grads): after
    data['loss'].backward() the leaves data['_head_inputs'] hold d loss / d (inputs of CoarseMatching / FineMatching), compared
    with torch.autograd of the reference's heads inside the reference's own training step (tests/golden/tgrad_*.npz: dual-softmax and Sinkhorn)."""
    import copy
    import importlib.util
    import os
    from _cases import GOLDEN_DIR
    from loftr_amd import LoFTR
    from loftr_amd.training import LoFTRLoss, compute_supervision_coarse, compute_supervision_fine
    spec = importlib.util.spec_from_file_location("make_golden_e2e", os.path.join(GOLDEN_DIR, "make_golden_e2e.py"))
    E2E = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(E2E)
    dev = torch.device("cuda", 0)
    g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))
    rc = json.loads(str(g["recipe"]))
    batch, geo = MG.step_batch(rc)
    N = geo["N"]
    cfg = MG.step_matcher_cfg(rc)
    cpu = LoFTR(copy.deepcopy(cfg))
    sd = E2E.e2e_state_dict(cpu, cfg, 0.3, rc["coarse_gain"], rc["fine_gain"])
    cpu.load_state_dict(sd, strict=True)
    cpu.train()
    with torch.no_grad():
        fc, ff = cpu.backbone(torch.from_numpy(np.concatenate([batch["image0"], batch["image1"]], 0)))
    model = LoFTR(copy.deepcopy(cfg))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    model.head_grads = True
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data = {"dataset_name": ["scannet"] * N, **{k: t(v) for k, v in batch.items()}}
    monkeypatch.setattr(torch, "randint", MG.det_randint)
    compute_supervision_coarse(data, CFG)
    data.update({"bs": N, "hw0_i": data["image0"].shape[2:], "hw1_i": data["image1"].shape[2:]})
    with torch.no_grad():                                  # like LoFTR.forward: the heads re-enable grad themselves
        model.match_from_features(fc[:N].to(dev), fc[N:].to(dev), ff[:N].to(dev), ff[N:].to(dev), data)
    compute_supervision_fine(data, CFG)
    LoFTRLoss(MG.step_loss_cfg(rc)).train()(data)
    for k in ("b_ids", "i_ids", "j_ids"):
        assert np.array_equal(data[k].cpu().numpy(), g[k]), k
    assert np.abs(data["expec_f"].detach().cpu().numpy() - g["expec_f"]).max() <= 3e-4
    want = json.loads(str(g["losses"]))
    for k in ("loss_c", "loss_f", "loss"):
        assert abs(float(data["loss_scalars"][k]) - want[k]) <= 2e-4 * max(1.0, abs(want[k])), (k, data["loss_scalars"], want)
    assert data["loss"].requires_grad
    data["loss"].backward()
    torch.cuda.synchronize()
    leaves = data["_head_inputs"]
    rel = lambda a, b: np.abs(a.astype(np.float64) - b).max() / np.abs(b).max()
    # the head inputs come from the HIP transformers (~1e-5 from the reference's): the gradients inherit that, on top of float32 autograd
    WW = leaves["feat_f0_unfold"].shape[1]
    f0g, f1g = leaves["feat_f0_unfold"].grad.cpu().numpy(), leaves["feat_f1_unfold"].grad.cpu().numpy()
    assert rel(leaves["feat_c0"].grad.cpu().numpy(), g["grad_feat_c0"]) <= 2e-3
    assert rel(leaves["feat_c1"].grad.cpu().numpy(), g["grad_feat_c1"]) <= 2e-3
    off = f0g.copy(); off[:, WW // 2] = 0
    assert np.abs(off).max() == 0
    assert rel(f0g[:, WW // 2], g["grad_feat_f0_centre"]) <= 2e-3
    n_tail = g["grad_feat_f1_tail"].shape[0]
    assert rel(f1g[-n_tail:], g["grad_feat_f1_tail"]) <= 2e-3
    assert rel(np.sqrt((f1g.reshape(f1g.shape[0], -1).astype(np.float64) ** 2).sum(1)), g["grad_feat_f1_norm"]) <= 2e-3
    if "grad_bin_score" in g:                              # the Sinkhorn head's parameter
        got = float(model.coarse_matching.bin_score.grad)
        assert abs(got - float(g["grad_bin_score"])) <= 2e-3 * abs(float(g["grad_bin_score"])), (got, g["grad_bin_score"])


@pytest.mark.parametrize("name,backbone_on", [("tfull_ds", "cpu"), ("tfull_ot", "cpu"), ("tfull_ds", "hip"), ("tfull_ot", "hip"),
                                              ("tfull_ds", "hipglue"), ("tfull_ot", "hipglue")])
def test_training_step_full_backward_against_reference(name, backbone_on, monkeypatch):
    """The WHOLE training step's backward (round 4, LoFTR.full_grads): supervision -> matcher in .train() mode -> losses ->
    data['loss'].backward(), every node after the backbone an autograd node whose forward and backward are HIP kernels (position
    encoding, 8 + 2 encoder layers, CoarseMatching, FinePreprocess, FineMatching, both losses), the backbone the PyTorch mirror on
    the CPU in train mode (its autograd, the reference's own arithmetic).  Compared: the gradient of EVERY parameter (160 / 161
    tensors) with the reference's own training step under torch.autograd (tests/golden/tfull_*.npz, digests).
    backbone_on = "hip": the backbone runs on the GPU in train mode as well, every convolution the HIP autograd node (autograd.conv2d:
    forward, input gradient, weight gradient), BatchNorm / activations / adds / upsampling PyTorch autograd on the GPU -- the SAME float32
    arithmetic as the reference's for those nodes, so the convolution nodes are held to the tight bar against the reference's float32 step.
    backbone_on = "hipglue" (round 5, the default of a training step): BatchNorm with batch statistics, activations, residual adds and the
    bilinear upsampling run on csrc/train_glue.hip as well.  That arithmetic is independent of the reference's float32 kernels (float64 sums
    for the statistics instead of float32 Welford ...), and the backbone's gradients at this size are ill-conditioned (the reference's own
    float32 step is up to 3.4e-2 from its float64 step; profiles/r05_glue_vs_fp64.txt: PyTorch's float32 graph, HIP convolutions + PyTorch
    glue and HIP convolutions + HIP glue sit at medians 3.3e-3 / 3.7e-3 / 3.3e-3 and maxima 4.2e-2 / 2.9e-2 / 1.9e-2 from a float64 run):
    two independent float32 evaluations differ by more than either differs from the truth, so the backbone tensors of this variant are
    measured against the reference's FLOAT64 gradients (grad64/* digests of the golden), next to the reference's own float32 distance."""
    import copy
    import importlib.util
    import os
    from _cases import GOLDEN_DIR
    from loftr_amd import LoFTR
    from loftr_amd.training import LoFTRLoss, compute_supervision_coarse, compute_supervision_fine

    def load_mod(fname):
        spec = importlib.util.spec_from_file_location(fname, os.path.join(GOLDEN_DIR, fname + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    E2E, LG = load_mod("make_golden_e2e"), load_mod("make_golden_layer_grad")
    dev = torch.device("cuda", 0)
    g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))
    rc = json.loads(str(g["recipe"]))
    batch, geo = MG.step_batch(rc)
    N = geo["N"]
    cfg = MG.step_matcher_cfg(rc)
    cpu = LoFTR(copy.deepcopy(cfg))
    sd = E2E.e2e_state_dict(cpu, cfg, 0.3, rc["coarse_gain"], rc["fine_gain"])
    cpu.load_state_dict(sd, strict=True)
    cpu.train()
    model = LoFTR(copy.deepcopy(cfg))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    model.full_grads = True
    images = torch.from_numpy(np.concatenate([batch["image0"], batch["image1"]], 0))
    if backbone_on == "cpu":
        fc, ff = cpu.backbone(images)                                                                        # WITH its graph
        eps = float(os.environ.get("LOFTR_TEST_PERTURB_FEATURES", "0"))      # sensitivity experiment (tools/gpu/r5_bwd_sensitivity.sh): how far do the
        if eps > 0:                                                          # gradients move when the features move by eps (relative, Gaussian)?
            gp = torch.Generator().manual_seed(11)
            fc = fc * (1 + eps * torch.randn(fc.shape, generator=gp))
            ff = ff * (1 + eps * torch.randn(ff.shape, generator=gp))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data = {"dataset_name": ["scannet"] * N, **{k: t(v) for k, v in batch.items()}}
    monkeypatch.setattr(torch, "randint", MG.det_randint)
    from loftr_amd import backbone as BBmod
    monkeypatch.setattr(BBmod, "TRAIN_GLUE_HIP", backbone_on == "hipglue")
    if backbone_on in ("hip", "hipglue"):
        # the user's training step (training.trainval_inference = lightning_loftr.py:76-91): supervision, LoFTR.forward(data) itself --
        # backbone in train mode on the stacked image batch, every convolution the HIP node, its two halves straight into the matcher --
        # fine supervision, loss
        from loftr_amd import autograd as AG
        from loftr_amd.training import trainval_inference
        calls0 = AG._Conv2d.calls
        trainval_inference(model, LoFTRLoss(MG.step_loss_cfg(rc)).train(), data, CFG)
        assert AG._Conv2d.calls - calls0 == sum(isinstance(m, torch.nn.Conv2d) for m in model.backbone.modules()), "HIP convolution nodes not used"
    else:
        compute_supervision_coarse(data, CFG)
        data.update({"bs": N, "hw0_i": data["image0"].shape[2:], "hw1_i": data["image1"].shape[2:]})
        with torch.enable_grad():
            model.match_from_features(fc[:N].to(dev), fc[N:].to(dev), ff[:N].to(dev), ff[N:].to(dev), data)
        compute_supervision_fine(data, CFG)
        LoFTRLoss(MG.step_loss_cfg(rc)).train()(data)
    want = json.loads(str(g["losses"]))
    for k in ("loss_c", "loss_f", "loss"):
        assert abs(float(data["loss_scalars"][k]) - want[k]) <= 2e-4 * max(1.0, abs(want[k])), (k, data["loss_scalars"], want)
    data["loss"].backward()
    torch.cuda.synchronize()
    params = {"backbone." + n: p for n, p in (model if backbone_on != "cpu" else cpu).backbone.named_parameters()}
    params.update({n: p for n, p in model.named_parameters() if not n.startswith("backbone.")})
    names = sorted({k.split("/")[1] for k in g if k.startswith("grad/")})
    assert set(names) == set(params), set(names) ^ set(params)
    def dist(key, got):
        if key in g:                                             # vectors (and scalars): stored whole
            ref = g[key]
            return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)
        d = LG.digest(key, got)
        scale = max(float(g[key + "/absmax"]), 1e-12)
        return max(np.abs(d[key + "/sub"] - g[key + "/sub"]).max() / scale,
                   max(np.abs(d[key + "/" + s_] - g[key + "/" + s_]).max() / max(np.abs(g[key + "/" + s_]).max(), scale) for s_ in ("rowsum", "colsum")))
    worst, worst64 = {}, {}
    for n in names:
        key = "grad/" + n
        got = params[n].grad
        got = (got if got is not None else torch.zeros_like(params[n])).detach().cpu().numpy()
        got = got.reshape(got.shape[0], -1) if got.ndim > 2 else got
        if key in g:                                             # vectors (and scalars): stored whole
            ref = g[key]
            err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)
        else:
            d = LG.digest(key, got)
            scale = max(float(g[key + "/absmax"]), 1e-12)
            err = max(np.abs(d[key + "/sub"] - g[key + "/sub"]).max() / scale,
                      max(np.abs(d[key + "/" + s_] - g[key + "/" + s_]).max() / max(np.abs(g[key + "/" + s_]).max(), scale) for s_ in ("rowsum", "colsum")))
        worst[n] = float(err)
        worst64[n] = float(dist("grad64/" + n, got.astype(np.float64)))
    # Bar per tensor, relative to its largest entry: 3e-3 (round-4 verdict; was 6e-3), or 3 x the reference's OWN float32 rounding noise on
    # that tensor -- |its fp32 gradient - its fp64 gradient| of the same step, stored in the golden -- where that is larger (BatchNorm
    # shifts and the stride-2 block of layer2 are sums with heavy cancellation: the reference itself is off by up to 3.4e-2 there, and two
    # float32 evaluations can differ by twice their common noise).
    # Three named exceptions at 8e-3, measured and explained in profiles/r05_backward_sensitivity.txt (tools/gpu/r5_bwd_sensitivity.sh):
    #   * the fine-level q projections (behind the attention normaliser of 25-token windows): ill-conditioned -- perturbing the backbone
    #     features by a RELATIVE 1e-6 moves loftr_fine.layers.1.q_proj.weight's gradient from 3.7e-3 to 5.8e-3 of its scale (to 3.0e-3 at
    #     1e-5): any two float32 forwards differ by more than that, the reference's same-input noise (7e-4) does not measure it;
    #   * loftr_coarse.layers.7.mlp.0.weight with the HIP backbone (5.6e-3; 3.5e-5 = the reference's noise with identical features): the
    #     layer next to the loss sees gradients on few tokens, and a ReLU unit whose pre-activation changes sign under the 1e-5 forward
    #     difference switches its whole contribution on or off -- the same experiment moves mlp.0 gradients of other layers by 24 x their
    #     noise (5.6e-4); the match set, the sampled ids and both losses are identical to 7 digits in the two variants (full_backward_all.txt).
    #     Shown directly (tools/micro/mlp_flip_probe.py, profiles/r05_mlp_flip_probe.txt): exactly one of the layer's 196 608 ReLU units changes
    #     sign between the two variants (|pre-activation| 1.7e-6); mlp.0's gradient moves by 5.3e-3, mlp.2 / norm2 behind the ReLU by 3-6e-5.
    noise = json.loads(str(g["ref_noise"]))
    assert bool(g["ref64_same_matches"])
    loose = {"loftr_fine.layers.0.q_proj.weight": 8e-3, "loftr_fine.layers.1.q_proj.weight": 8e-3}
    if backbone_on != "cpu":
        loose["loftr_coarse.layers.7.mlp.0.weight"] = 8e-3
    # BatchNorm shifts of the backbone: the plain sum of the feature gradient over every pixel of the batch (cancellation): 4e-3
    # (measured 3.06e-3 on backbone.layer1.1.bn2.bias with the CPU mirror, where the reference's own noise is 7.4e-4)
    is_shift = lambda n: n.startswith("backbone.") and n.endswith(".bias")
    tol = {n: max(loose.get(n, 4e-3 if is_shift(n) else 3e-3), 3.0 * noise[n]) for n in worst}
    hip = {n: e for n, e in worst.items() if not n.startswith("backbone.")}
    bb = {n: e for n, e in worst.items() if n.startswith("backbone.")}
    rep = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "gpurun_out")
    os.makedirs(rep, exist_ok=True)
    with open(os.path.join(rep, "full_backward_margins.txt"), "a") as fh:
        fh.write(f"{name} (backbone on {backbone_on}): {len(hip)} matcher tensors, worst {max(hip.values()):.2e} ({max(hip, key=hip.get)}; reference fp32-vs-fp64 there "
                 f"{noise[max(hip, key=hip.get)]:.2e}); {len(bb)} backbone tensors, worst {max(bb.values()):.2e} ({max(bb, key=bb.get)}; reference "
                 f"{noise[max(bb, key=bb.get)]:.2e}); tensors above 2e-3: {sum(e > 2e-3 for e in worst.values())}; top: "
                 + ", ".join(f"{n} {e:.1e}" for n, e in sorted(worst.items(), key=lambda kv: -kv[1])[:4]) + "\n")
    with open(os.path.join(rep, "full_backward_all.txt"), "a") as fh:          # every tensor: distance, the reference's own fp32 noise, ratio
        ids = torch.stack([data["b_ids"], data["i_ids"], data["j_ids"]], 1).cpu().numpy().astype(np.int64)
        import hashlib
        fh.write(f"== {name} (backbone on {backbone_on}); training matches M = {len(ids)}, sha1 of (b, i, j) {hashlib.sha1(ids.tobytes()).hexdigest()[:12]}, "
                 f"loss_c {float(data['loss_scalars']['loss_c']):.7f} loss_f {float(data['loss_scalars']['loss_f']):.7f}\n" + "".join(f"{n:52s} {e:.2e}  ref noise {noise[n]:.2e}  x{e / max(noise[n], 1e-12):7.1f}\n"
                                                                         for n, e in sorted(worst.items(), key=lambda kv: -kv[1])))
    if backbone_on == "hipglue":
        # Backbone tensors against the reference's FLOAT64 gradients, as ratios to the reference's own float32 distance.  The tight statement
        # is the aggregate: median ratio <= 2 (measured 1.29 / 1.30).  Individual tensors cannot be held to a few x their noise: the
        # derivative of ReLU is discontinuous, and with the sparse gradients of a matching loss ONE unit whose pre-activation lies within
        # rounding noise of zero can carry percent of a layer's gradient.  Measured (tools/micro/relu_flip_probe.py,
        # profiles/r05_relu_flip_probe.txt): between the PyTorch-BatchNorm and the HIP-BatchNorm evaluation of tfull_ot (outputs equal to
        # 1e-6) exactly one of the 1 572 864 units behind layer1.1.bn1 changes sign (|pre-activation| 5e-7), and that layer's bias gradient
        # moves by 2.9e-2 while every layer upstream of it agrees to <= 7e-4; the tensors downstream of it (layer1.0, the stem) inherit 4e-3.
        # So: at most a quarter of the backbone tensors beyond 6 x their noise, none beyond 5e-2 of its scale.
        ratios = sorted(worst64[n] / max(noise[n], 1e-12) for n in bb)
        with open(os.path.join(rep, "full_backward_margins.txt"), "a") as fh:
            fh.write(f"   {name} (hipglue) backbone tensors vs the reference's float64 gradients: ratio to the reference's own float32 distance: median {ratios[len(ratios) // 2]:.2f}, "
                     f"max {ratios[-1]:.2f}, beyond 6x: {sum(r > 6 for r in ratios)} of {len(ratios)}; worst {max(bb, key=lambda n: worst64[n])} {max(worst64[n] for n in bb):.2e}\n")
        assert ratios[len(ratios) // 2] <= 2.0, ratios[len(ratios) // 2]
        assert sum(worst64[n] > max(3e-3, 6.0 * noise[n]) for n in bb) <= len(bb) // 4, sorted(((worst64[n], noise[n], n) for n in bb), reverse=True)[:16]
        bad = {n: (worst64[n], noise[n]) for n in bb if worst64[n] > 5e-2}
        bad.update({n: (e, tol[n]) for n, e in hip.items() if e > tol[n]})
    else:
        bad = {n: (e, tol[n]) for n, e in worst.items() if e > tol[n]}
    assert not bad, (sorted(bad.items(), key=lambda kv: -kv[1][0])[:8], len(bad), len(worst))


@pytest.mark.parametrize("match_type", ["dual_softmax", "sinkhorn"])
def test_train_mode_forward_from_images_with_head_grads(match_type, monkeypatch):
    """LoFTR.forward(data) in .train() mode with head_grads straight from images, then LoFTRLoss and loss.backward(): the four
    transformer outputs come back as leaves with finite, non-trivial gradients (the Sinkhorn head also fills bin_score.grad),
    and nothing upstream of them is touched (no parameter of the transformers / backbone receives a gradient)."""
    import copy
    from loftr_amd import LoFTR
    from loftr_amd.training import LoFTRLoss, compute_supervision_coarse, compute_supervision_fine
    dev = torch.device("cuda", 0)
    rc = dict(MG.STEP_CASES["tstep_ds"], match_type=match_type)
    batch, geo = MG.step_batch(rc)
    cfg = MG.step_matcher_cfg(rc)
    torch.manual_seed(0)
    model = LoFTR(copy.deepcopy(cfg)).to(dev).train()
    model.head_grads = True
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data = {"dataset_name": ["scannet"] * geo["N"], **{k: t(v) for k, v in batch.items()}}
    monkeypatch.setattr(torch, "randint", MG.det_randint)
    compute_supervision_coarse(data, CFG)
    model(data)
    compute_supervision_fine(data, CFG)
    LoFTRLoss(MG.step_loss_cfg(rc)).train()(data)
    assert data["loss"].requires_grad and torch.isfinite(data["loss"])
    data["loss"].backward()
    leaves = data["_head_inputs"]
    assert set(leaves) == {"feat_c0", "feat_c1", "feat_f0_unfold", "feat_f1_unfold"}
    for k, v in leaves.items():
        assert v.grad is not None and v.grad.shape == v.shape and torch.isfinite(v.grad).all(), k
    assert float(leaves["feat_c0"].grad.abs().max()) > 0 and float(leaves["feat_c1"].grad.abs().max()) > 0
    for name, prm in model.named_parameters():
        if name == "coarse_matching.bin_score":
            assert prm.grad is not None and torch.isfinite(prm.grad)
        else:
            assert prm.grad is None, name
