#!/usr/bin/env python
"""Where does a tile of conv3x3_duo_kernel spend its time?  Needs the probe build (python -m loftr_amd.build --variant probe
-DLOFTR_CONV_PROBE; run with LOFTR_HIP_LIB=loftr_amd/libloftr_hip_probe.so): every workgroup stamps the 100 MHz wall clock at
entry, before its k-loop, after it and after its epilogue, plus HW_ID / XCC_ID.    python tools/micro/conv_probe.py [layer substring]"""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import ops, _lib  # noqa: E402
from tools.micro.conv_layers import layers  # noqa: E402


def main(only="layer1 3x3", B=16):
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    lib.loftr_conv_probe_buffer.argtypes = [C.c_void_p]
    buf = torch.zeros(1 << 16, 6, dtype=torch.int64, device=dev)
    assert lib.loftr_conv_probe_buffer(buf.data_ptr()) == 0
    for tag, cin, cout, k, s, h, w, res, act in layers():
        if only not in tag or k != 3 or s != 1:
            continue
        conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev)
        bn = nn.BatchNorm2d(cout).to(dev).eval()
        x = ops.sp_from_nhwc(torch.relu(torch.randn(B, h, w, cin, device=dev)))
        r = ops.sp_from_nhwc(torch.relu(torch.randn(B, h, w, cout, device=dev))) if res else None
        for _ in range(3):
            ops.conv_bn_act(x, cin, conv, bn, act=act, residual=r, want_sp=True)
        buf.zero_()
        torch.cuda.synchronize()
        for _ in range(8):            # back to back: the stamps left in the buffer are the LAST launch's (steady clocks, not a launch after an idle GPU)
            ops.conv_bn_act(x, cin, conv, bn, act=act, residual=r, want_sp=True)
        torch.cuda.synchronize()
        t = buf.cpu().numpy()
        t = t[t[:, 0] != 0]
        t0 = t[:, 0].min()
        st, lo, hi, en = [(t[:, i] - t0) / 100.0 for i in range(4)]        # us
        hw, xcc = t[:, 4], t[:, 5]
        cu = ((xcc & 0xf) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) << 1 | ((hw >> 8) & 0xf)
        print(f"{tag}: {len(t)} workgroups, launch span {en.max():.1f} us; distinct CU ids {len(np.unique(cu))}")
        print(f"  per workgroup [us]: prologue {np.mean(lo - st):.2f}  k-loop {np.mean(hi - lo):.2f}  epilogue {np.mean(en - hi):.2f}  "
              f"total {np.mean(en - st):.2f}   (k-loop min {np.min(hi - lo):.2f} max {np.max(hi - lo):.2f}; epilogue min {np.min(en - hi):.2f} max {np.max(en - hi):.2f})")
        kl = np.sort(hi - lo)
        print(f"  k-loop percentiles [us]: 5 % {kl[len(kl) // 20]:.2f}  25 % {kl[len(kl) // 4]:.2f}  50 % {kl[len(kl) // 2]:.2f}  75 % {kl[3 * len(kl) // 4]:.2f}  95 % {kl[19 * len(kl) // 20]:.2f}")
        print(f"  wave-slot parity of wave 0: {np.bincount((hw & 1).astype(int))}")
        # phase of the two workgroups that share a CU: for every CU, sort its workgroups by start and look at how much of each
        # epilogue overlaps a k-loop of ANOTHER workgroup on the same CU
        ov = []
        for c in np.unique(cu):
            m = np.where(cu == c)[0]
            for a in m:
                e0, e1 = hi[a], en[a]
                o = 0.0
                for b_ in m:
                    if b_ != a:
                        o += max(0.0, min(e1, hi[b_]) - max(e0, lo[b_]))
                ov.append(o / max(e1 - e0, 1e-9))
        print(f"  fraction of an epilogue that runs under another workgroup's k-loop on the same CU: mean {np.mean(ov):.2f}")
        first = np.argsort(st)[:8]
        for a in first:
            print(f"    wg start {st[a]:7.2f} loop {lo[a]:7.2f}..{hi[a]:7.2f} end {en[a]:7.2f} cu {cu[a]:#x} slot {hw[a] & 15} simd {(hw[a] >> 4) & 3}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "layer1 3x3", int(sys.argv[2]) if len(sys.argv) > 2 else 16)
