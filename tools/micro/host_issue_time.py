#!/usr/bin/env python
"""How long does the HOST need to issue a forward (Python + ctypes + launches), against the GPU time of the same forward?
A spin kernel queued first keeps the GPU behind the host, so the stamps before the one device sync (the match count) are pure
host time.    python tools/micro/host_issue_time.py [batch=1]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import LoFTR, ops                 # noqa: E402
from loftr_amd.config import get_cfg             # noqa: E402
from loftr_amd.synth import make_images          # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
cfg = get_cfg(thr=0.0)
cfg["coarse"]["temp_bug_fix"] = True
model = LoFTR(cfg).eval().cuda()
i0, i1 = make_images(1234, NB, 480, 640)
a, b = torch.from_numpy(i0).cuda(), torch.from_numpy(i1).cuda()
stamps = {}
orig = ops.coarse_match


def wrapped(*args, **kw):
    stamps["match_in"] = time.perf_counter()
    r = orig(*args, **kw)
    stamps["match_out"] = time.perf_counter()
    return r


ops.coarse_match = wrapped
for _ in range(3):
    model({"image0": a, "image1": b})
torch.cuda.synchronize()
rows = []
for it in range(10):
    torch.cuda.synchronize()
    if it % 2:
        torch.cuda._sleep(int(60e6))              # ~ 25 ms of GPU spin: the host runs ahead
    t0 = time.perf_counter()
    model({"image0": a, "image1": b})
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((it % 2, 1e3 * (stamps["match_in"] - t0), 1e3 * (stamps["match_out"] - stamps["match_in"]), 1e3 * (t1 - stamps["match_out"]), 1e3 * (t2 - t0)))
for spin in (0, 1):
    r = [x for x in rows if x[0] == spin]
    med = lambda k: sorted(x[k] for x in r)[len(r) // 2]
    print(f"batch {NB} {'host ahead (spin queued first)' if spin else 'normal':32s}: backbone + coarse transformer issued after {med(1):6.2f} ms, coarse matching incl. the sync {med(2):6.2f} ms, "
          f"fine stage issued in {med(3):5.2f} ms, GPU done after {med(4):6.2f} ms")
