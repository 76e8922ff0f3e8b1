#!/usr/bin/env python
"""autograd.conv2d weight / input gradients against float64 at training sizes; data: 'randn' or 'relu' (x >= 0 with a positive mean, dy zero-mean:
the cancellation a real step has).   python tools/micro/conv_wgrad_probe.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import autograd  # noqa: E402

for (B, Cin, Cout, K, s, p, H, W) in [(4, 256, 256, 1, 1, 0, 60, 80), (4, 196, 256, 1, 1, 0, 120, 160), (1, 128, 256, 1, 1, 0, 160, 160), (1, 128, 256, 1, 1, 0, 256, 256), (1, 128, 128, 1, 1, 0, 256, 256)]:
    for data in ("randn",):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, Cin, H, W, generator=g)
        if data == "relu":
            x = x.clamp_min(0) + 0.3
        w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
        gy = None
        outs = {}
        for name, dt, fn in (("f64", torch.float64, None), ("f32", torch.float32, None), ("hip", torch.float32, autograd.conv2d)):
            xx = x.to("cuda", dt).requires_grad_(True)
            ww = w.to("cuda", dt).requires_grad_(True)
            y = fn(xx, ww, s, p) if fn else F.conv2d(xx, ww, None, s, p)
            if gy is None:
                gy = torch.randn(y.shape, generator=g) * 1e-3
            y.backward(gy.to("cuda", dt))
            outs[name] = (xx.grad.double().cpu(), ww.grad.double().cpu())
        msg = []
        for i, what in enumerate(("dx", "dw")):
            ref = outs["f64"][i]; sc = float(ref.abs().max())
            msg.append(f"{what}: f32 {float((outs['f32'][i] - ref).abs().max()) / sc:.2e} hip {float((outs['hip'][i] - ref).abs().max()) / sc:.2e}")
        print((B, Cin, Cout, K, s, H, W), data, " | ".join(msg), flush=True)
