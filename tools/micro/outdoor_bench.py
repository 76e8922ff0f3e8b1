#!/usr/bin/env python
"""BASELINE configs[3] (outdoor, 840 x 840, MegaDepth-style padding masks + scales): full forward timing, N pairs.

    python tools/micro/outdoor_bench.py [N] [reps] [sinkhorn] [mode=launches|persistent|auto] [overlap=0|1] [skip=0|1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import LoFTR                      # noqa: E402
from loftr_amd.config import get_cfg             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ot = "sinkhorn" in sys.argv[3:]
kv = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
torch.manual_seed(0)
cfg = get_cfg(thr=0.0, border_rm=2)
if ot:
    cfg["match_coarse"].update(match_type="sinkhorn", skh_prefilter=False, sparse_spvs=True)
model = LoFTR(cfg).eval().cuda()
if "mode" in kv:
    model.coarse_mode = kv["mode"]
if "skip" in kv:
    model.skip_padded_tiles = bool(int(kv["skip"]))
if "overlap" in kv:
    model.overlap_fine_branch = bool(int(kv["overlap"]))
g = torch.Generator().manual_seed(1234)
img0 = torch.rand(N, 1, 840, 840, generator=g)
img1 = (img0.roll((8, 16), (2, 3)) + 0.02 * torch.rand(N, 1, 840, 840, generator=g)).clamp(0, 1)
img0[:, :, 560:] = 0; img1[:, :, 560:] = 0                               # valid region 840 x 560, zero-padded bottom
mask = torch.zeros(N, 105, 105, dtype=torch.bool); mask[:, :70] = True
batch = lambda: {"image0": img0.cuda(), "image1": img1.cuda(), "mask0": mask.cuda(), "mask1": mask.cuda(),
                 "scale0": torch.full((N, 2), 1.9).cuda(), "scale1": torch.full((N, 2), 1.9).cuda()}
for _ in range(2):
    d = batch(); model(d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
d = batch()
e0.record()
for _ in range(reps):
    dd = dict(d); model(dd)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"outdoor 840x840 {'sinkhorn' if ot else 'dual-softmax'}, N={N}{''.join(' ' + k + '=' + v for k, v in kv.items())}: {ms:.2f} ms per forward = {N / ms * 1e3:.1f} pairs/s, M = {dd['mconf'].shape[0]}")
