#!/usr/bin/env python
"""Backbone training step (forward + backward of a scalar loss) three ways: float64 PyTorch (truth), HIP convolutions + PyTorch glue,
HIP convolutions + HIP glue (csrc/train_glue.hip).  Per parameter: distance of the two float32 variants to the float64 gradient."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import LoFTR, get_cfg, backbone as BB
torch.manual_seed(0)
net = LoFTR(get_cfg(thr=0.0)).backbone.cuda().train()
sd = {k: v.clone() for k, v in net.state_dict().items()}
g = torch.Generator().manual_seed(1)
x = torch.rand(4, 1, 240, 320, generator=g).cuda()
gc = torch.randn(4, 256, 30, 40, generator=g).cuda() * 1e-3
gf = torch.randn(4, 128, 120, 160, generator=g).cuda() * 1e-3
if os.environ.get("SPARSE"):                       # a training step's upstream gradients: a few hundred matched cells / windows only
    mc = (torch.rand(4, 1, 30, 40, generator=g) < 0.05).cuda(); mf = (torch.rand(4, 1, 120, 160, generator=g) < 0.002).cuda()
    gc, gf = gc * mc, gf * mf
res = {}
for tag, glue, conv, dt in (("f64", False, False, torch.float64), ("torch glue", False, True, torch.float32), ("hip glue", True, True, torch.float32),
                            ("all torch f32", False, False, torch.float32)):
    BB.TRAIN_GLUE_HIP, BB.TRAIN_CONV_HIP = glue, conv
    if glue and os.environ.get("PARTS"):
        BB.GLUE_PARTS = set(os.environ["PARTS"].split(","))
    net.load_state_dict(sd)
    net.to(dt)
    net.zero_grad()
    fc, ff = net(x.to(dt))
    ((fc * gc.to(dt)).sum() + (ff * gf.to(dt)).sum()).backward()
    res[tag] = {n: p.grad.double().clone() for n, p in net.named_parameters()}
    res[tag]["__fc"], res[tag]["__ff"] = fc.detach().double(), ff.detach().double()
    net.float()
ref = res["f64"]
rows = []
for n in ref:
    sc = float(ref[n].abs().max())
    rows.append((n, *(float((res[t][n] - ref[n]).abs().max()) / sc for t in ("all torch f32", "torch glue", "hip glue"))))
rows.sort(key=lambda r: -r[3])
print(f"{'tensor':44s} {'all torch f32':>14s} {'HIP conv+torch glue':>20s} {'HIP conv+HIP glue':>18s}   (max |d| / max |ref|, vs float64)")
for r in rows[:14]:
    print(f"{r[0]:44s} {r[1]:14.2e} {r[2]:20.2e} {r[3]:18.2e}")
import statistics
for i, t in enumerate(("all torch f32", "HIP conv + torch glue", "HIP conv + HIP glue")):
    v = [r[1 + i] for r in rows]
    print(f"{t:24s} median {statistics.median(v):.2e}  max {max(v):.2e}")
