#!/usr/bin/env python
"""Per-item trace of the persistent coarse transformer INSIDE a full LoFTR.forward of the bench's batch (the backbone has just streamed its
activations through the caches), next to the same call repeated on its own.

    python tools/micro/pct_in_model.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import LoFTR, get_cfg, ops                                   # noqa: E402
from loftr_amd.synth import make_images, make_weights, make_backbone_weights   # noqa: E402

cfg = get_cfg(thr=0.0)
model = LoFTR(cfg).eval()
sd = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(0, cfg).items()}
for k, v in make_backbone_weights(7, model.backbone, 0.3).items():
    sd["backbone." + k] = v
model.load_state_dict(sd, strict=True)
model = model.cuda()
model.overlap_fine_branch = False
model.coarse_mode = "persistent"
i0, i1 = make_images(1234, 8, 480, 640)
img0, img1 = torch.from_numpy(i0).cuda(), torch.from_numpy(i1).cuda()
N, L = 8, 4800
kinds = [0, 1] * 4
n_items = ops._lib.load().loftr_coarse_plan_bytes((ops.C.c_int * 8)(*kinds), 8, N, L, L) // 32 - 1
diag = torch.zeros(16 + 32 * n_items, dtype=torch.uint8, device="cuda")
orig = ops.transformer
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
saved = {}


def traced(feat0, feat1, *a, **kw):
    if feat0.shape[2] == 256:
        kw["diag"] = diag
        saved["args"] = (feat0.clone(), feat1.clone(), a, dict(kw))
        ev[0].record()
        r = orig(feat0, feat1, *a, **kw)
        ev[1].record()
        return r
    return orig(feat0, feat1, *a, **kw)


def report(tag):
    torch.cuda.synchronize()
    raw = diag[16:].view(torch.int64).cpu().numpy().reshape(n_items, 4)
    plan = ops.coarse_plan(kinds, N, L, L, img0.device, 0).cpu().numpy().view(np.uint32).reshape(-1, 8)[1:]
    typ = plan[:, 0] & 15
    t0 = raw[:, 0].min()
    pop, rdy, done = (raw[:, 0] - t0) * 0.01, (raw[:, 1] - t0) * 0.01, (raw[:, 2] - t0) * 0.01
    print(f"{tag}: events {ev[0].elapsed_time(ev[1]):.3f} ms, makespan {done.max():.1f} us")
    for t, name in ((0, "X"), (1, "K"), (2, "F")):
        m = typ == t
        dur, wait = (done - rdy)[m], (rdy - pop)[m]
        print(f"  {name}: run median {np.median(dur):7.2f} us p90 {np.percentile(dur, 90):7.2f}  sum(run)/256 {dur.sum() / 256:7.1f} us  sum(wait)/256 {wait.sum() / 256:6.1f} us")


ops.transformer = traced
with torch.no_grad():
    for it in range(3):
        diag.zero_()
        model({"image0": img0, "image1": img1})
        report(f"in model, forward {it}")
    f0, f1, a, kw = saved["args"]
    ops.transformer = orig
    for it in range(3):
        diag.zero_()
        g0, g1 = f0.clone(), f1.clone()
        torch.cuda.synchronize()
        ev[0].record()
        orig(g0, g1, *a, **kw)
        ev[1].record()
        report(f"same inputs, alone, run {it}")
