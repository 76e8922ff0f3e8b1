#!/usr/bin/env python
"""Stand-alone timing of the coarse LocalFeatureTransformer (8 layers, N pairs of 4800 x 256 tokens, seeded weights).

    python tools/micro/encoder_bench.py [N] [reps] [L]      (LOFTR_HIP_LIB=... selects a variant build)

Prints the library's per-kernel hipEvent timings (encoder kernels) and the whole-transformer time per call."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import LoFTR, get_cfg, _lib   # noqa: E402
from loftr_amd.synth import make_weights      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4800
cfg = get_cfg(thr=0.0)
model = LoFTR(cfg).eval()
model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(0, cfg).items()}, strict=False)
model = model.cuda()
g = torch.Generator(device="cpu").manual_seed(0)
both = torch.randn(2 * N, L, 256, generator=g).cuda()
lib = _lib.load()
ids = {lib.loftr_hip_timing_kernel_name(i).decode(): i for i in range(lib.loftr_hip_timing_kernel_count())}
names = [n for n in ("encoder_x_kernel", "proj_kv_kernel", "proj_kernel", "linear_kernel", "linear_ln_kernel") if n in ids]


def run():
    b = both.clone()
    return model.loftr_coarse(b[:N], b[N:], inplace=True)


with torch.no_grad():
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    mask = 0
    for n in names:
        mask |= 1 << ids[n]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    lib.loftr_hip_timing_enable(mask)
    tot = 0.0
    for _ in range(reps):
        b = both.clone()
        ev[0].record()
        model.loftr_coarse(b[:N], b[N:], inplace=True)
        ev[1].record()
        torch.cuda.synchronize()
        tot += ev[0].elapsed_time(ev[1])
    lib.loftr_hip_timing_enable(0)
out = [f"transformer {tot / reps:.3f} ms/call"]
for k in names:
    ms, n = C.c_double(0), C.c_longlong(0)
    lib.loftr_hip_timing_read(ids[k], C.byref(ms), C.byref(n), 1)
    if n.value:
        out.append(f"{k.replace('_kernel', '')} {ms.value / reps:.3f} ms/call ({ms.value / n.value * 1e3:.1f} us x {n.value // reps})")
print(os.environ.get("LOFTR_HIP_LIB", "product").split("libloftr_hip")[-1], " | ".join(out))
