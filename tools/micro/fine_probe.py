#!/usr/bin/env python
"""Where does a match of fine_pair_kernel spend its time?  Needs the probe build (python -m loftr_amd.build --variant fprobe
-DLOFTR_FINE_PROBE; run with LOFTR_HIP_LIB=loftr_amd/libloftr_hip_fprobe.so): every wave sums the 100 MHz wall clock over its
phases (four encoder calls).     python tools/micro/fine_probe.py [M=6200]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import LoFTR, get_cfg, _lib   # noqa: E402
from loftr_amd.synth import make_weights      # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 6200
cfg = get_cfg(thr=0.0)
w = make_weights(0, cfg)
model = LoFTR(cfg).eval()
model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in w.items()}, strict=False)
model = model.cuda()
g = torch.Generator(device="cpu").manual_seed(0)
f0 = torch.randn(M, 25, 128, generator=g).cuda()
f1 = (0.6 * f0 + 0.8 * torch.randn(M, 25, 128, generator=g).cuda()).contiguous()
lib = _lib.load()
lib.loftr_fine_probe_buffer.argtypes = [C.c_void_p]
buf = torch.zeros(((M + 3) // 4) * 4, 8, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(3):
        model.loftr_fine(f0.clone(), f1.clone(), inplace=True)
    assert lib.loftr_fine_probe_buffer(buf.data_ptr()) == 0
    torch.cuda.synchronize()
    a, b = f0.clone(), f1.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.loftr_fine(a, b, inplace=True)
    e1.record()
    torch.cuda.synchronize()
t = buf.cpu().numpy()[:M]
names = ["call head", "source side: K / V panels + K^T V (216 MFMAs)", "x side: Q, attention, merge (216 MFMAs)", "LayerNorm1 + fragments",
         "mlp (576 MFMAs)", "LayerNorm2 + residual + repack"]
tot = (t[:, 7] - t[:, 6]) / 100.0
print(f"fine transformer M={M}: {e0.elapsed_time(e1) * 1e3:.1f} us; per wave (= match): {tot.mean():.1f} us (min {tot.min():.1f}, max {tot.max():.1f}); "
      f"launch span {(t[:, 7].max() - t[:, 6].min()) / 100.0:.1f} us")
mf = [0, 216, 216, 0, 576, 0]
for i, n in enumerate(names):
    us = t[:, i].mean() / 100.0
    extra = f"   = {us / 4 / mf[i] * 1e3:.1f} ns per MFMA (32 clk = 13.3 ns at 2.4 GHz)" if mf[i] else ""
    print(f"  {n:52s} {us:7.2f} us over the 4 calls ({us / tot.mean() * 100:4.1f} %){extra}")
