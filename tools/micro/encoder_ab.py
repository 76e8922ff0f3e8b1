#!/usr/bin/env python
"""Coarse transformer output of this process's library settings -> gpurun_out/enc_<tag>.pt; compares bitwise with every other tag on disk.

    LOFTR_ENCODER_SCHEDULE=0 python tools/micro/encoder_ab.py callwise [N] [H W]; python tools/micro/encoder_ab.py scheduled [N] [H W]
"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import LoFTR, get_cfg          # noqa: E402
from loftr_amd.synth import make_weights      # noqa: E402

tag = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L0 = int(sys.argv[3]) if len(sys.argv) > 3 else 4800
L1 = int(sys.argv[4]) if len(sys.argv) > 4 else L0
masked = len(sys.argv) > 5 and sys.argv[5] == "mask"
cfg = get_cfg(thr=0.0)
model = LoFTR(cfg).eval()
model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(0, cfg).items()}, strict=False)
model = model.cuda()
g = torch.Generator(device="cpu").manual_seed(N * 7 + L0 + L1)
f0 = torch.randn(N, L0, 256, generator=g).cuda()
f1 = torch.randn(N, L1, 256, generator=g).cuda()
m0 = m1 = None
if masked:
    m0 = torch.ones(N, L0, dtype=torch.bool); m0[:, L0 - L0 // 5:] = False
    m1 = torch.ones(N, L1, dtype=torch.bool); m1[0, L1 - L1 // 3:] = False
    m0, m1 = m0.cuda(), m1.cuda()
with torch.no_grad():
    o0, o1 = model.loftr_coarse(f0, f1, m0, m1)
torch.cuda.synchronize()
out = os.environ.get("LOFTR_AB_DIR") or os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
key = f"{N}_{L0}_{L1}_{int(masked)}"
torch.save({"o0": o0.cpu(), "o1": o1.cpu()}, os.path.join(out, f"enc_{key}_{tag}.pt"))
for f in sorted(glob.glob(os.path.join(out, f"enc_{key}_*.pt"))):
    if f.endswith(f"_{tag}.pt"):
        continue
    o = torch.load(f)
    same = torch.equal(o["o0"], o0.cpu()) and torch.equal(o["o1"], o1.cpu())
    print(f"{key}: {tag} vs {os.path.basename(f)}: {'bit-identical' if same else 'DIFFERENT max ' + str(float((o['o0'] - o0.cpu()).abs().max()))}"
          f"  finite={bool(torch.isfinite(o0).all() and torch.isfinite(o1).all())}")
