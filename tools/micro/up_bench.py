#!/usr/bin/env python
"""The two FPN top-down steps of ResNetFPN_8_2: lateral 1x1 convolution + bilinear x2 upsample-add.
    python tools/micro/up_bench.py [B=16] [H=480] [W=640] [key=value ...]      (image size; the steps run at 1/4 and 1/2 of it)
    key=value: debug switches of the library"""
import os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import ops  # noqa: E402
a = [x for x in sys.argv[1:] if "=" not in x]
for kv in [x for x in sys.argv[1:] if "=" in x]:
    ops.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
B = int(a[0]) if len(a) > 0 else 16
H = int(a[1]) if len(a) > 1 else 480
W = int(a[2]) if len(a) > 2 else 640
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for tag, cin, cout, h, w in ((f"layer2_outconv 196>256 @1/4 {B}x{H // 4}x{W // 4}", 196, 256, H // 4, W // 4),
                             (f"layer1_outconv 128>196 @1/2 {B}x{H // 2}x{W // 2}", 128, 196, H // 2, W // 2)):
    conv = nn.Conv2d(cin, cout, 1, bias=False).to(dev)
    x = ops.sp_from_nhwc(torch.relu(torch.randn(B, h, w, cin, device=dev)))
    low = ops.sp_from_nhwc(torch.randn(B, h // 2, w // 2, cout, device=dev))
    f = lambda: ops.conv1x1_upsample_add(x, cin, conv, low)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = B * (h * w * ops.ceil32(cin) + h * w // 4 * ops.ceil32(cout) + h * w * ops.ceil32(cout)) * 4 / 1e9
    print(f"{tag:44s} {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s of the algorithmic {gb:.2f} GB", flush=True)
