#!/usr/bin/env python
"""Stand-alone timing of the coarse-matching kernels (dual-softmax): N pairs of random 4800 x 256 descriptors.

    python tools/micro/score_bench.py [N] [reps]

Prints the library's per-kernel hipEvent timings; wrap in `rocprofv3 --pmc ... --` for counters."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import ops, _lib   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
h, w = 60, 80
g = torch.Generator(device="cpu").manual_seed(0)
f0 = torch.randn(N, h * w, 256, generator=g).cuda()
f1 = (0.5 * f0.roll(3, 1) + 0.9 * torch.randn(N, h * w, 256, generator=g).cuda()).contiguous()
lib = _lib.load()
ids = {lib.loftr_hip_timing_kernel_name(i).decode(): i for i in range(lib.loftr_hip_timing_kernel_count())}
for _ in range(2):
    r = ops.coarse_match(f0, f1, (h, w), (h, w), thr=0.0, border_rm=2, scale=8.0)
torch.cuda.synchronize()
lib.loftr_hip_timing_enable((1 << ids["score_sweep_kernel<0>"]) | (1 << ids["score_sweep_kernel<1>"]))
for _ in range(reps):
    r = ops.coarse_match(f0, f1, (h, w), (h, w), thr=0.0, border_rm=2, scale=8.0)
torch.cuda.synchronize()
for k in ("score_sweep_kernel<0>", "score_sweep_kernel<1>"):
    ms, n = C.c_double(0), C.c_longlong(0)
    lib.loftr_hip_timing_read(ids[k], C.byref(ms), C.byref(n), 1)
    print(f"{k}: {ms.value / max(n.value, 1) * 1e3:.1f} us per launch ({n.value} launches)")
print("M =", r["mconf"].shape[0], "conf max", float(r["conf_matrix"].max()))
sim = torch.einsum("nlc,nsc->nls", f0[:2].double(), f1[:2].double()) / (256 * 0.1)
ref = torch.softmax(sim, 1) * torch.softmax(sim, 2)
d = (r["conf_matrix"][:2].double() - ref).abs()
print("vs fp64 reference (2 pairs): max |d conf| %.3e, max rel at the row maxima %.3e" % (float(d.max()), float((d.amax(2) / ref.amax(2)).max())))
if "--diag" in sys.argv:
    c = r["conf_matrix"][0].double()
    R = c / ref[0].clamp_min(1e-300)
    a = R.median(dim=1).values          # per-row factor
    b = R.median(dim=0).values          # per-column factor
    bad_r = (a - 1).abs() > 1e-4
    bad_c = (b - 1).abs() > 1e-4
    print("rows off:", int(bad_r.sum()), "first", bad_r.nonzero()[:8].flatten().tolist(), "factors", a[bad_r][:6].tolist())
    print("cols off:", int(bad_c.sum()), "first", bad_c.nonzero()[:8].flatten().tolist(), "factors", b[bad_c][:6].tolist())
    print("row factor range", float(a.min()), float(a.max()), "col factor range", float(b.min()), float(b.max()))
