#!/usr/bin/env python
"""Duration of the heads' backward at the bench's sizes (N pairs of 4800 x 4800 cells, M fine windows).

    python tools/micro/grad_bench.py [N=8] [M=7700] [reps=5] [with_bmm=0]
"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from loftr_amd import autograd, ops  # noqa: E402


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def main(N=8, M=7700, reps=5, with_bmm=0):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    h, w = 60, 80
    L = h * w
    f0 = torch.randn(N, L, 256, device=dev)
    f1 = torch.randn(N, L, 256, device=dev)
    G = torch.zeros(N, L, L, device=dev)
    idx = torch.randint(L, (N, 900), device=dev)
    for n in range(N):
        G[n, idx[n], torch.randint(L, (900,), device=dev)] = 1.0
    t_dsim = timed(lambda: ops.dual_softmax_bwd(f0, f1, G, (h, w), (h, w), 0.1), reps)
    dsim = ops.dual_softmax_bwd(f0, f1, G, (h, w), (h, w), 0.1)
    # what round 3 ran (rocBLAS fp32): only on request, so that the profile of this script shows the product's kernels only
    t_gemm = timed(lambda: (torch.bmm(dsim, f1), torch.bmm(dsim.transpose(1, 2), f0)), reps) if with_bmm else float('nan')
    t_own = timed(lambda: ops.head_feat_grads(dsim, f0, f1, 1.0 / 25.6), reps)                    # csrc/head_grads.hip
    Gd = torch.randn(N, L, L, device=dev)
    t_dense = timed(lambda: ops.dual_softmax_bwd(f0, f1, Gd, (h, w), (h, w), 0.1), reps)
    kw = dict(thr=0.2, border_rm=2, scale=8.0, match_type="dual_softmax", temperature=0.1, want_conf=True)
    t_fwd = timed(lambda: ops.coarse_match(f0, f1, (h, w), (h, w), **kw), reps)
    Ga = torch.zeros(N, L + 1, L + 1, device=dev)
    Ga[:, :L, :L] = G
    Ga[:, :L, L] = 1e-3
    t_ot = timed(lambda: ops.sinkhorn_bwd(f0, f1, Ga, (h, w), (h, w), 1.0, 3), reps)
    okw = dict(thr=0.2, border_rm=2, scale=8.0, match_type="sinkhorn", bin_score=1.0, skh_iters=3, skh_prefilter=False, want_assign=True)
    t_otf = timed(lambda: ops.coarse_match(f0, f1, (h, w), (h, w), **okw), reps)
    a, b = torch.randn(M, 25, 128, device=dev), torch.randn(M, 25, 128, device=dev)
    ge = torch.randn(M, 3, device=dev)
    z2, zb = torch.zeros(M, 2, device=dev), torch.zeros(M, dtype=torch.long, device=dev)
    t_ffwd = timed(lambda: ops.fine_match(a, b, z2, zb, 2.0), reps)
    t_fbwd = timed(lambda: ops.fine_match_bwd(a, b, ge), reps)
    print(f"N={N} L=S={L}: coarse match forward {t_fwd:.3f} ms | dual-softmax backward: dsim {t_dsim:.3f} ms (sparse G) / {t_dense:.3f} ms (dense G), "
          f"+ the two feature-gradient GEMMs {t_own:.3f} ms (csrc/head_grads.hip; torch.bmm fp32: {t_gemm:.3f} ms) | Sinkhorn (3 iterations): forward {t_otf:.3f} ms, backward {t_ot:.3f} ms (+ the same two GEMMs) | M={M}: fine match forward {t_ffwd:.3f} ms, backward {t_fbwd:.3f} ms")


if __name__ == "__main__":
    main(*[int(x) for x in sys.argv[1:]])
