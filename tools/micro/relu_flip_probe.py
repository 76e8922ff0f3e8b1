#!/usr/bin/env python
"""Why two float32 evaluations of the SAME training step differ by 1e-2 on a few early-layer gradient tensors (tests/test_hip_training.py, hipglue
variant): ReLU units whose pre-activation is within rounding noise of zero switch their whole gradient contribution on or off.  The tfull_ot step
is run twice -- BatchNorm on PyTorch and on csrc/train_glue.hip (outputs agree to ~1e-6) -- and for every BatchNorm -> ReLU of the backbone the
units whose mask differs are counted, with the share of that layer's bias gradient they carry."""
import sys, os, json, copy, importlib.util
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import GOLDEN_DIR
def load_mod(fname):
    spec = importlib.util.spec_from_file_location(fname, os.path.join(GOLDEN_DIR, fname + ".py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod); return mod
E2E, MG = load_mod("make_golden_e2e"), load_mod("make_golden_train")
from loftr_amd import LoFTR, backbone as BB
from loftr_amd.training import LoFTRLoss, trainval_inference
import test_hip_training as T
name = sys.argv[1] if len(sys.argv) > 1 else "tfull_ot"
g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))
rc = json.loads(str(g["recipe"]))
batch, geo = MG.step_batch(rc)
cfg = MG.step_matcher_cfg(rc)
dev = torch.device("cuda", 0)
res = {}
for parts in ("", "bn"):
    BB.TRAIN_GLUE_HIP = bool(parts); BB.GLUE_PARTS = set(parts.split(","))
    model = LoFTR(copy.deepcopy(cfg))
    model.load_state_dict(E2E.e2e_state_dict(model, cfg, 0.3, rc["coarse_gain"], rc["fine_gain"]), strict=True)
    model = model.to(dev).train(); model.full_grads = True
    for mod in model.backbone.modules():
        if isinstance(mod, torch.nn.ReLU):
            mod.inplace = False
    cap = {}
    def fhook(m, i, o, nm):
        cap["y/" + nm] = o.detach()
        o.register_hook(lambda gr, nm=nm: cap.__setitem__("dy/" + nm, gr.detach()))
    for nm, mod in model.backbone.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.register_forward_hook(lambda m, i, o, nm=nm: fhook(m, i, o, nm))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data = {"dataset_name": ["scannet"] * geo["N"], **{k: t(v) for k, v in batch.items()}}
    real = torch.randint
    torch.randint = MG.det_randint
    try:
        trainval_inference(model, LoFTRLoss(MG.step_loss_cfg(rc)).train(), data, T.CFG)
    finally:
        torch.randint = real
    data["loss"].backward()
    res[parts] = (cap, {n: p.grad.detach().double() for n, p in model.backbone.named_parameters()})
a, b = res[""], res["bn"]
print(f"{name}: BatchNorm -> ReLU layers of the backbone, PyTorch BatchNorm vs csrc/train_glue.hip")
for k in [k for k in a[0] if k.startswith("y/")]:
    nm = k[2:]
    if nm.endswith("bn2") or "downsample" in nm or nm.endswith("outconv2.1") and False:
        pass
    ya, yb = a[0][k], b[0][k]
    flips = (ya > 0) != (yb > 0)
    nflip = int(flips.sum())
    ga, gb = a[1][nm + ".bias"], b[1][nm + ".bias"]
    dgrad = float((ga - gb).abs().max() / ga.abs().max())
    line = f"  {nm:26s} |y_torch - y_hip| {float((ya - yb).abs().max() / ya.abs().max()):.1e}  units with different sign: {nflip:3d} of {ya.numel():9d}   bias-gradient difference {dgrad:.1e}"
    if nflip and ("dy/" + nm) in a[0]:
        line += f"  (largest |pre-activation| among them {float(torch.maximum(ya.abs(), yb.abs())[flips].max()):.1e})"
    print(line)
