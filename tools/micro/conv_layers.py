#!/usr/bin/env python
"""Per-layer timing of the backbone's convolutions (ResNetFPN_8_2, 16 images of 480x640 = the bench step) through the C-ABI:
one line per layer with its launch time, the executed fp16 MFMA rate (3 MFMAs per fp32 product, USEFUL flops only) and the
fraction of the 2.5 PF dense peak.     python tools/micro/conv_layers.py [B=16] [iters=10] [only=substring] [key=value ...]
(key=value: debug switches of the library, e.g. conv_rem=0)
LOFTR_HIP_LIB=<variant .so> selects another build of the library (A/B)."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import ops  # noqa: E402

# (tag, cin, cout, k, stride, h_in, w_in, residual, act)
def layers(h=480, w=640):
    h2, w2, h4, w4, h8, w8 = h // 2, w // 2, h // 4, w // 4, h // 8, w // 8
    return [
        ("layer1 3x3 128>128 @1/2", 128, 128, 3, 1, h2, w2, True, 1),
        ("layer1 3x3 128>128 @1/2 (conv1: no residual)", 128, 128, 3, 1, h2, w2, False, 1),
        ("layer2.0 3x3s2 128>196", 128, 196, 3, 2, h2, w2, False, 1),
        ("layer2.0 ds 1x1s2 128>196", 128, 196, 1, 2, h2, w2, False, 0),
        ("layer2 3x3 196>196 @1/4", 196, 196, 3, 1, h4, w4, True, 1),
        ("layer2 3x3 196>196 @1/4 (conv1: no residual)", 196, 196, 3, 1, h4, w4, False, 1),
        ("layer3.0 3x3s2 196>256", 196, 256, 3, 2, h4, w4, False, 1),
        ("layer3.0 ds 1x1s2 196>256", 196, 256, 1, 2, h4, w4, False, 0),
        ("layer3 3x3 256>256 @1/8", 256, 256, 3, 1, h8, w8, True, 1),
        ("layer3 3x3 256>256 @1/8 (conv1: no residual)", 256, 256, 3, 1, h8, w8, False, 1),
        ("layer3_outconv 1x1 256>256 @1/8", 256, 256, 1, 1, h8, w8, False, 0),
        ("layer2_outconv2.0 3x3 256>256 @1/4", 256, 256, 3, 1, h4, w4, False, 2),
        ("layer2_outconv2.3 3x3 256>196 @1/4", 256, 196, 3, 1, h4, w4, False, 0),
        ("layer1_outconv2.0 3x3 196>196 @1/2", 196, 196, 3, 1, h2, w2, False, 2),
        ("layer1_outconv2.3 3x3 196>128 @1/2", 196, 128, 3, 1, h2, w2, False, 0),
    ]


def main(B=16, iters=10, only="", *switches):
    for kv in switches:
        if kv.split("=")[0] == "conv_rem":                    # the remainder form needs the scratch buffer ops.conv_bn_act allocates when asked to
            ops.CONV_REM = bool(int(kv.split("=")[1]))
        else:
            ops.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    print(f"{'layer':48s} {'us':>9s} {'GFLOP':>8s} {'exec PF':>8s} {'frac':>6s}")
    total = 0.0
    # warm the GPU up first (clocks / power state).  NOTE (round 6): the rows WITH a residual read ~150 us higher here than the same launch inside
    # the model or in an interleaved series (tools/micro/conv_residual_exp.py: +90 us for the residual stream, not +190); cause not found
    warm = torch.randn(4096, 4096, device=dev)
    for _ in range(60):
        warm = (warm @ warm).clamp_(-1, 1)
    torch.cuda.synchronize()
    for tag, cin, cout, k, s, h, w, res, act in layers():
        if only and only not in tag:
            continue
        conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev)
        bn = nn.BatchNorm2d(cout).to(dev).eval() if act or res else None
        x = ops.sp_from_nhwc(torch.relu(torch.randn(B, h, w, cin, device=dev)))      # post-ReLU statistics (half zeros): power / clocks depend on the data
        ho, wo = h // s, w // s
        r = ops.sp_from_nhwc(torch.relu(torch.randn(B, ho, wo, cout, device=dev))) if res else None
        f = lambda: ops.conv_bn_act(x, cin, conv, bn, act=act, residual=r, want_sp=True)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        fl = 2.0 * B * ho * wo * cout * cin * k * k
        pf = 3 * fl / (us * 1e-6) / 1e15
        total += us
        print(f"{tag:48s} {us:9.1f} {fl / 1e9:8.1f} {pf:8.3f} {pf / 2.5:6.3f}", flush=True)
    print(f"{'sum (one launch each)':48s} {total:9.1f}")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if len(a) > 0 else 16, int(a[1]) if len(a) > 1 else 10, a[2] if len(a) > 2 else "", *a[3:])
