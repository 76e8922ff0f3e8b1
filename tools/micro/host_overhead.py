#!/usr/bin/env python
"""Host (Python + launch) time of one forward against its GPU time: the margin by which the CPU runs ahead of the GPU."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import LoFTR                      # noqa: E402
from loftr_amd.config import get_cfg             # noqa: E402
from loftr_amd.synth import make_images          # noqa: E402

torch.manual_seed(0)
cfg = get_cfg(thr=0.0)
cfg["coarse"]["temp_bug_fix"] = True
model = LoFTR(cfg).eval().cuda()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
i0, i1 = make_images(1234, NB, 480, 640)
a, b = torch.from_numpy(i0).cuda(), torch.from_numpy(i1).cuda()
for _ in range(3):
    model({"image0": a, "image1": b})
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model({"image0": a, "image1": b})
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
print(f"batch {NB}: forward call returns after {1e3 * sorted(host)[5]:.2f} ms (includes the one device sync on the match count), GPU done after {1e3 * sorted(total)[5]:.2f} ms")
