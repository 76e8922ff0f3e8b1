#!/usr/bin/env python
"""Round-4 verdict, weak #2: `loftr_coarse.layers.7.mlp.0.weight`'s gradient is 5.6e-3 of its scale from the reference's with the HIP backbone
and 3.5e-5 with the CPU-mirror backbone (identical features).  Hypothesis: ReLU units of that layer's MLP (transformer.py:55, hidden =
relu(mlp.0([x, message]))) whose pre-activation is within the 1e-5 forward difference of zero switch their gradient on or off.  Test: run the
tfull_ds step both ways, take the INPUTS of layers.7 in both, evaluate the layer's hidden pre-activation in float64 from each, and count
the units whose sign differs -- next to the gradient difference of mlp.0 and of the tensors behind the ReLU (mlp.2, norm2), which must not move."""
import sys, os, json, copy, importlib.util
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import GOLDEN_DIR
def load_mod(fname):
    spec = importlib.util.spec_from_file_location(fname, os.path.join(GOLDEN_DIR, fname + ".py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod); return mod
E2E, MG = load_mod("make_golden_e2e"), load_mod("make_golden_train")
from loftr_amd import LoFTR, backbone as BB
from loftr_amd.training import LoFTRLoss, trainval_inference, compute_supervision_coarse, compute_supervision_fine
import test_hip_training as T
name = "tfull_ds"
g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))
rc = json.loads(str(g["recipe"]))
batch, geo = MG.step_batch(rc)
N = geo["N"]
cfg = MG.step_matcher_cfg(rc)
dev = torch.device("cuda", 0)
BB.TRAIN_GLUE_HIP = False                                    # the 'hip' variant of the test: HIP convolutions, PyTorch glue
res = {}
for variant in ("cpu", "hip"):
    cpu = LoFTR(copy.deepcopy(cfg))
    sd = E2E.e2e_state_dict(cpu, cfg, 0.3, rc["coarse_gain"], rc["fine_gain"])
    cpu.load_state_dict(sd, strict=True); cpu.train()
    model = LoFTR(copy.deepcopy(cfg)); model.load_state_dict(sd, strict=True)
    model = model.to(dev).train(); model.full_grads = True
    calls = []
    layer = model.loftr_coarse.layers[7]
    layer.register_forward_pre_hook(lambda m, args: calls.append(tuple(None if a is None else a.detach().double() for a in args)))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data = {"dataset_name": ["scannet"] * N, **{k: t(v) for k, v in batch.items()}}
    real = torch.randint
    torch.randint = MG.det_randint
    try:
        if variant == "hip":
            trainval_inference(model, LoFTRLoss(MG.step_loss_cfg(rc)).train(), data, T.CFG)
        else:
            images = torch.from_numpy(np.concatenate([batch["image0"], batch["image1"]], 0))
            fc, ff = cpu.backbone(images)
            compute_supervision_coarse(data, T.CFG)
            data.update({"bs": N, "hw0_i": data["image0"].shape[2:], "hw1_i": data["image1"].shape[2:]})
            with torch.enable_grad():
                model.match_from_features(fc[:N].to(dev), fc[N:].to(dev), ff[:N].to(dev), ff[N:].to(dev), data)
            compute_supervision_fine(data, T.CFG)
            LoFTRLoss(MG.step_loss_cfg(rc)).train()(data)
    finally:
        torch.randint = real
    data["loss"].backward()
    w = {k: v.detach().double() for k, v in layer.state_dict().items()}
    pre = []
    for args in calls:                                       # transformer.py:35-58 / linear_attention.py:20-47 in float64 up to the ReLU's input
        x, source = args[0], args[1]
        nb, L, C = x.shape; S = source.shape[1]; H = 8; D = C // H
        q = (x @ w["q_proj.weight"].t()).view(nb, L, H, D); k = (source @ w["k_proj.weight"].t()).view(nb, S, H, D); v = (source @ w["v_proj.weight"].t()).view(nb, S, H, D)
        Q, K = F.elu(q) + 1, F.elu(k) + 1
        KV = torch.einsum("nshd,nshv->nhdv", K, v / S)
        Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + 1e-6)
        msg = (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S).reshape(nb, L, C)
        msg = F.layer_norm(msg @ w["merge.weight"].t(), (C,), w["norm1.weight"], w["norm1.bias"])
        pre.append(torch.cat([x, msg], dim=2) @ w["mlp.0.weight"].t())
    res[variant] = (calls, pre, {n: p.grad.detach().double() for n, p in layer.named_parameters()})
rel = lambda a, b: float((a - b).abs().max() / a.abs().max())
ca, cb = res["cpu"], res["hip"]
print(f"{name}: loftr_coarse.layers.7 (the last cross layer, called twice), CPU-mirror backbone vs HIP backbone")
for i in range(len(ca[1])):
    pa, pb = ca[1][i], cb[1][i]
    flips = (pa > 0) != (pb > 0)
    print(f"  call {i}: layer input |x_cpu - x_hip| {rel(ca[0][i][0], cb[0][i][0]):.1e} of its scale; hidden pre-activations {rel(pa, pb):.1e}; "
          f"ReLU units with different sign: {int(flips.sum())} of {pa.numel()}" + (f" (largest |pre-activation| among them {float(torch.maximum(pa.abs(), pb.abs())[flips].max()):.1e} of scale {float(pa.abs().max()):.1f})" if int(flips.sum()) else ""))
for n in ("mlp.0.weight", "mlp.2.weight", "norm2.weight", "norm2.bias", "merge.weight", "q_proj.weight"):
    print(f"  gradient of {n:14s}: CPU-backbone run vs HIP-backbone run {rel(ca[2][n], cb[2][n]):.1e}")
