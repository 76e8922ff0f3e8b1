#!/usr/bin/env python
"""What does the residual branch of a BasicBlock convolution cost?  layer1 (128 > 128 at 1/2 resolution, 16 images), interleaved series:
no residual / residual = another tensor / residual = the input (aliased: no third stream) / residual = zeros.
    python tools/micro/conv_residual_exp.py [rounds=4] [iters=20]"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, h, w, c = 16, 240, 320, 128
conv = nn.Conv2d(c, c, 3, 1, 1, bias=False).to(dev)
bn = nn.BatchNorm2d(c).to(dev).eval()
x = ops.sp_from_nhwc(torch.relu(torch.randn(B, h, w, c, device=dev)))
r = ops.sp_from_nhwc(torch.relu(torch.randn(B, h, w, c, device=dev)))
rz = ops.sp_from_nhwc(torch.zeros(B, h, w, c, device=dev))


def t(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


cases = [("no residual", None), ("residual = other tensor", r), ("residual = the input", x), ("residual = zeros", rz)]
res = {k: [] for k, _ in cases}
for _ in range(rounds):
    for k, rr in cases:
        res[k].append(t(lambda: ops.conv_bn_act(x, c, conv, bn, act=1, residual=rr, want_sp=True)))
for k, _ in cases:
    print(f"{k:26s} " + " ".join(f"{v:7.1f}" for v in res[k]) + f"   median {sorted(res[k])[len(res[k]) // 2]:7.1f} us")

# ---- does the RELATIVE placement of the three streams (input, residual, output) matter?  the residual at several offsets inside a larger buffer
print(f"x at {x.data_ptr():#x}  r at {r.data_ptr():#x}  (r - x) mod 1 MiB = {(r.data_ptr() - x.data_ptr()) % (1 << 20)}")
y = ops.conv_bn_act(x, c, conv, bn, act=1, residual=r, want_sp=True)
yt = y[0] if isinstance(y, (tuple, list)) else y
print(f"output at {yt.data_ptr():#x}  (y - x) mod 1 MiB = {(yt.data_ptr() - x.data_ptr()) % (1 << 20)}")
del y, yt
big = torch.empty(r.numel() + (1 << 22), dtype=r.dtype, device=dev)
offs = [0, 64, 1024, 16384, 65536 + 1024, 262144 + 4096 + 64]
series = {o: [] for o in offs}
views = {}
for o in offs:
    views[o] = None
for _ in range(3):
    for o in offs:
        v = big[o:o + r.numel()].view(r.shape)
        v.copy_(r)
        series[o].append(t(lambda: ops.conv_bn_act(x, c, conv, bn, act=1, residual=v, want_sp=True)))
for o in offs:
    v = big[o:o + r.numel()]
    print(f"residual at x + {(v.data_ptr() - x.data_ptr()) % (1 << 21):8d} B (mod 2 MiB), offset {4 * o:8d} B in its buffer: " + " ".join(f"{q:7.1f}" for q in series[o]))
