#!/usr/bin/env python
"""Which split-K partial of loftr_conv_wgrad is wrong?  1 x 1 problem: ws holds [ns][Cout][Cin] after the call."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import _lib, ops  # noqa: E402
lib = _lib.load()
chunk = int(os.environ.get("LOFTR_WGRAD_CHUNK", "0"))
assert chunk, "set LOFTR_WGRAD_CHUNK"
for (B, H, W, Cin, Cout) in [(4, 60, 80, 256, 256), (1, 256, 256, 128, 256)]:
    T = B * H * W
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    dy = (torch.randn(B, H, W, Cout, generator=g) * 1e-3).cuda()
    if os.environ.get("CONST") == "1":                       # every product the same: a mix-up of data is invisible, broken arithmetic is not
        x.fill_(1.0); dy.fill_(1e-3)
    if os.environ.get("CONST") == "2":                       # x constant per pixel-chunk: partial n = known multiple
        x.fill_(1.0); dy.copy_((torch.arange(T, device="cuda") // chunk + 1).float().view(B, H, W, 1).expand(B, H, W, Cout) * 1e-3)
    nbytes = lib.loftr_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, 1, 1)
    for fill in (0.0, 7.0):
        ws = torch.full((nbytes // 4 + 16,), fill, dtype=torch.float32, device="cuda")
        taps = torch.empty(1, Cout, Cin, device="cuda")
        rc = lib.loftr_conv_wgrad(C.c_void_p(dy.data_ptr()), C.c_void_p(x.data_ptr()), B, H, W, Cin, Cout, 1, 1, 1, 0, C.c_void_p(taps.data_ptr()),
                                  C.c_void_p(ws.data_ptr()), ws.numel() * 4, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        ns = (T + chunk - 1) // chunk
        part = ws[:ns * Cout * Cin].view(ns, Cout, Cin).double()
        X, D = x.view(T, Cin).double(), dy.view(T, Cout).double()
        ref = torch.stack([D[n * chunk:(n + 1) * chunk].t() @ X[n * chunk:(n + 1) * chunk] for n in range(ns)])
        err = (part - ref).abs().amax(dim=(1, 2)) / ref.abs().amax()
        bad = (err > 1e-5).nonzero().flatten().tolist()
        tot = (taps[0].double() - ref.sum(0)).abs().max() / ref.sum(0).abs().max()
        sumerr = (taps[0].double() - part.sum(0)).abs().max() / ref.sum(0).abs().max()
        print(f"T={T} Cin={Cin} Cout={Cout} ns={ns} fill={fill} rc={rc}: bad partials {len(bad)} {bad[:12]} max partial err {float(err.max()):.2e}; dW err {float(tot):.2e}; dW vs sum of partials {float(sumerr):.2e}; ws MB {nbytes / 1e6:.1f}")
        if bad:
            n = bad[0]
            e = (part[n] - ref[n]).abs()
            rows = (e.amax(1) > 1e-5 * ref.abs().amax()).nonzero().flatten().tolist()
            cols = (e.amax(0) > 1e-5 * ref.abs().amax()).nonzero().flatten().tolist()
            print(f"   partial {n}: bad rows {len(rows)} [{rows[:4]}..{rows[-4:]}], bad cols {len(cols)} [{cols[:4]}..{cols[-4:]}]; equals fill: {bool((part[n][rows[0]][cols[0]] == fill))}")
