#!/usr/bin/env python
"""head_feat_grads at full batch (304 workgroups: two per CU on some CUs) against float64: both products (plain and transposed A)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import ops  # noqa: E402
for N in (1, 8):
    g = torch.Generator().manual_seed(N)
    L = S = 4800
    dsim = (torch.randn(N, L, S, generator=g) * 1e-3).cuda()
    f0 = torch.randn(N, L, 256, generator=g).cuda(); f1 = torch.randn(N, S, 256, generator=g).cuda()
    g0, g1 = ops.head_feat_grads(dsim, f0, f1, 0.5)
    r0 = 0.5 * torch.bmm(dsim.double(), f1.double()); r1 = 0.5 * torch.bmm(dsim.double().transpose(1, 2), f0.double())
    e0 = (g0.double() - r0).abs().amax(dim=(1, 2)) / r0.abs().max(); e1 = (g1.double() - r1).abs().amax(dim=(1, 2)) / r1.abs().max()
    print(f"N={N}: g0 (plain) max err per pair {[f'{float(v):.1e}' for v in e0]}\n      g1 (transposed) {[f'{float(v):.1e}' for v in e1]}")
