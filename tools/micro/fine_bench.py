#!/usr/bin/env python
"""Fine-level LocalFeatureTransformer on M window pairs (25 x 128): fused kernel (fine_fused.hip) vs the per-layer kernels
(LOFTR_FUSED_FINE=0 in a second process would be the clean A/B; here both run in THIS process via the env switch read at the
first call, so run the script twice) and vs an fp64 torch restatement on a few matches.

    python tools/micro/fine_bench.py [M] [reps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import LoFTR, get_cfg, _lib   # noqa: E402
from loftr_amd.synth import make_weights      # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 7700
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = get_cfg(thr=0.0)
w = make_weights(0, cfg)
model = LoFTR(cfg).eval()
model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in w.items()}, strict=False)
model = model.cuda()
g = torch.Generator(device="cpu").manual_seed(0)
f0 = torch.randn(M, 25, 128, generator=g).cuda()
f1 = (0.6 * f0 + 0.8 * torch.randn(M, 25, 128, generator=g).cuda()).contiguous()
lib = _lib.load()
ids = {lib.loftr_hip_timing_kernel_name(i).decode(): i for i in range(lib.loftr_hip_timing_kernel_count())}
names = [n for n in ("fine_pair_kernel", "proj_kernel", "attn_small_kernel", "linear_kernel", "linear_ln_kernel") if n in ids]
with torch.no_grad():
    for _ in range(2):
        o0, o1 = model.loftr_fine(f0, f1)
    torch.cuda.synchronize()
    mask = 0
    for n in names:
        mask |= 1 << ids[n]
    lib.loftr_hip_timing_enable(mask)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    tot = 0.0
    for _ in range(reps):
        a, b = f0.clone(), f1.clone()
        ev[0].record()
        model.loftr_fine(a, b, inplace=True)
        ev[1].record()
        torch.cuda.synchronize()
        tot += ev[0].elapsed_time(ev[1])
    lib.loftr_hip_timing_enable(0)
out = [f"fine transformer M={M}: {tot / reps:.3f} ms/call"]
for k in names:
    ms, n = C.c_double(0), C.c_longlong(0)
    lib.loftr_hip_timing_read(ids[k], C.byref(ms), C.byref(n), 1)
    if n.value:
        out.append(f"{k.replace('_kernel', '')} {ms.value / reps:.3f} ms ({n.value // reps} launches)")
print(" | ".join(out))

# fp64 restatement of transformer.py:35-58 / linear_attention.py:20-47 on the first matches
def enc(x, s, p):
    W = lambda n: torch.from_numpy(np.asarray(w[p + n])).double().cuda()
    q, k, v = x @ W("q_proj.weight").T, s @ W("k_proj.weight").T, s @ W("v_proj.weight").T
    B, L, _ = q.shape
    Q = torch.nn.functional.elu(q.view(B, L, 8, 16)) + 1
    K = torch.nn.functional.elu(k.view(B, -1, 8, 16)) + 1
    V = v.view(B, -1, 8, 16) / s.shape[1]
    KV = torch.einsum("nshd,nshv->nhdv", K, V)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + 1e-6)
    msg = (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * s.shape[1]).reshape(B, L, 128)
    msg = torch.nn.functional.layer_norm(msg @ W("merge.weight").T, (128,), W("norm1.weight"), W("norm1.bias"))
    h = torch.relu(torch.cat([x, msg], 2) @ W("mlp.0.weight").T) @ W("mlp.2.weight").T
    return x + torch.nn.functional.layer_norm(h, (128,), W("norm2.weight"), W("norm2.bias"))

for lo in sorted({0, max(0, min(M, 1024 + 64) - 64), max(0, M - 64)}):
    n = min(M - lo, 64)
    x0, x1 = f0[lo:lo + n].double(), f1[lo:lo + n].double()
    x0, x1 = enc(x0, x0, "loftr_fine.layers.0."), enc(x1, x1, "loftr_fine.layers.0.")
    x0 = enc(x0, x1, "loftr_fine.layers.1.")
    x1 = enc(x1, x0, "loftr_fine.layers.1.")
    d0 = (o0[lo:lo + n].double() - x0).abs().amax((1, 2))
    d1 = (o1[lo:lo + n].double() - x1).abs().amax((1, 2))
    print("vs fp64 restatement, matches %d..%d: max |d| f0 %.3e  f1 %.3e   (|f| max %.2f)  bad matches %s" % (
        lo, lo + n - 1, float(d0.max()), float(d1.max()), float(x0.abs().max()), [lo + int(i) for i in ((d0 > 1e-3) | (d1 > 1e-3)).nonzero().flatten()[:16]]))
