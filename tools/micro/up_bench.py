#!/usr/bin/env python
"""The two FPN top-down steps of ResNetFPN_8_2 at the bench size (16 images): lateral 1x1 convolution + bilinear x2 upsample-add.
    python tools/micro/up_bench.py     (LOFTR_CONV_UP_SMALL=0/1 selects the workgroup shape)"""
import os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import ops  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for tag, cin, cout, h, w in (("layer2_outconv 196>256 @1/4", 196, 256, 120, 160), ("layer1_outconv 128>196 @1/2", 128, 196, 240, 320)):
    conv = nn.Conv2d(cin, cout, 1, bias=False).to(dev)
    x = ops.sp_from_nhwc(torch.relu(torch.randn(16, h, w, cin, device=dev)))
    low = ops.sp_from_nhwc(torch.randn(16, h // 2, w // 2, cout, device=dev))
    f = lambda: ops.conv1x1_upsample_add(x, cin, conv, low)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = 16 * (h * w * ops.ceil32(cin) + h * w // 4 * ops.ceil32(cout) + h * w * ops.ceil32(cout)) * 4 / 1e9
    print(f"{tag:32s} {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s of the algorithmic {gb:.2f} GB")
