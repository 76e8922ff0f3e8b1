#!/usr/bin/env python
"""Stand-alone timing of the Sinkhorn coarse matching (configs[4]): N pairs of random 4800 x 256 descriptors.

    python tools/micro/ot_bench.py [N] [reps] [h=60] [w=80]        (wrap in `rocprofv3 --kernel-trace --stats` for the kernel split)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import ops   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
h = int(sys.argv[3]) if len(sys.argv) > 3 else 60
w = int(sys.argv[4]) if len(sys.argv) > 4 else 80
g = torch.Generator(device="cpu").manual_seed(0)
f0 = torch.randn(N, h * w, 256, generator=g).cuda()
f1 = (0.5 * f0.roll(3, 1) + 0.9 * torch.randn(N, h * w, 256, generator=g).cuda()).contiguous()
kw = dict(thr=0.2, border_rm=2, scale=8.0, match_type="sinkhorn", bin_score=1.0, skh_iters=3, skh_prefilter=False)
for sparse in (True, False):
    for _ in range(2):
        r = ops.coarse_match(f0, f1, (h, w), (h, w), want_assign=sparse, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = ops.coarse_match(f0, f1, (h, w), (h, w), want_assign=sparse, **kw)
    e1.record()
    torch.cuda.synchronize()
    print(f"sinkhorn coarse matching, N={N}, {h} x {w}, conf_matrix_with_bin={'yes' if sparse else 'no'}: {e0.elapsed_time(e1) / reps:.3f} ms per call (M = {r['mconf'].shape[0]})")
