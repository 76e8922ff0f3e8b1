#!/usr/bin/env python
"""loftr_linear_fwd (out = a w^T) against float64 over row counts and shapes the backward passes use."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import ops  # noqa: E402
g = torch.Generator().manual_seed(0)
for M in (4800, 9600, 14400, 19200, 38400, 50000, 76800):
    row = []
    for (N, K) in ((256, 256), (512, 512), (256, 512), (512, 256), (128, 128), (256, 128)):
        a = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        out = ops.linear(a, w)
        ref = a.double() @ w.double().T
        err = (out.double() - ref).abs().amax(1) / ref.abs().max()
        bad = (err > 1e-5).nonzero().flatten()
        row.append(f"({N},{K}) {float(err.max()):.1e}" + (f" bad rows {bad.numel()} [{int(bad[0])}..{int(bad[-1])}]" if bad.numel() else ""))
    print(f"M={M}: " + "  ".join(row), flush=True)
