#!/usr/bin/env python
"""Backbone in .train() mode: parameter gradients with the convolutions on the HIP autograd node against the same step on PyTorch's own
convolutions (float32 and float64), same weights, same output gradients.   python tools/micro/backbone_grad_ab.py [N=2] [H=480] [W=640]"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import LoFTR, get_cfg, backbone as BB  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = LoFTR(get_cfg()).backbone.to(dev).train()
img = torch.rand(2 * N, 1, H, W, device=dev)
gc = gf = None
res = {}
for tag, hip, dt in (("f64", False, torch.float64), ("f32", False, torch.float32), ("hip", True, torch.float32)):
    mm = copy.deepcopy(m).to(dt)
    BB.TRAIN_CONV_HIP = hip
    c, f = mm(img.to(dt))
    if gc is None:
        gc, gf = torch.randn_like(c) * 1e-3, torch.randn_like(f) * 1e-3
    ((c * gc.to(dt)).sum() + (f * gf.to(dt)).sum()).backward()
    res[tag] = ({n: p.grad.double() for n, p in mm.named_parameters()}, c.detach().double(), f.detach().double())
for i, nm in ((1, "coarse"), (2, "fine")):
    r = res["f64"][i]
    print(f"forward {nm}: f32 {float((res['f32'][i] - r).abs().max() / r.abs().max()):.2e}  hip {float((res['hip'][i] - r).abs().max() / r.abs().max()):.2e}")
for n in res["f64"][0]:
    r = res["f64"][0][n]
    s = float(r.abs().max()) + 1e-30
    e32, eh = float((res["f32"][0][n] - r).abs().max()) / s, float((res["hip"][0][n] - r).abs().max()) / s
    print(f"{n:40s} f32 {e32:.2e}  hip {eh:.2e}{'   <<<' if eh > 4 * e32 + 1e-5 else ''}")
