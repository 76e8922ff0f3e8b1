#!/usr/bin/env python
"""head_grad_kernel co-residency fault: which step of the staging path produces the bad operand?
Needs a library built with -DHG_DIAG (loftr_hg_diag); constant operands so that any zero operand element is a fault.
   LOFTR_HIP_LIB=loftr_amd/libloftr_hip_hgdiag.so LOFTR_WGRAD_CHUNK=128 python tools/micro/hg_diag.py [reps]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import _lib  # noqa: E402
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
has_diag = hasattr(raw, "loftr_hg_diag")
chunk = int(os.environ.get("LOFTR_WGRAD_CHUNK", "0"))
assert chunk, "set LOFTR_WGRAD_CHUNK"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
KIND = ["zero ra", "zero rb", "LDS readback A", "LDS readback B", "zero hi in A frag", "zero hi in B frag", "B frag re-read differs"]
tot_bad = 0
for (B, H, W, Cin, Cout) in [(4, 60, 80, 256, 256), (1, 256, 256, 128, 256)]:
    T = B * H * W
    x = torch.full((B, H, W, Cin), 1.0, device="cuda")
    dy = torch.full((B, H, W, Cout), 1e-3, device="cuda")
    nbytes = lib.loftr_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, 1, 1)
    ns = (T + chunk - 1) // chunk
    for rep in range(reps):
        ws = torch.full((nbytes // 4 + 16,), 7.0, dtype=torch.float32, device="cuda")
        taps = torch.empty(1, Cout, Cin, device="cuda")
        if has_diag:
            raw.loftr_hg_diag(None, 1)
        rc = lib.loftr_conv_wgrad(C.c_void_p(dy.data_ptr()), C.c_void_p(x.data_ptr()), B, H, W, Cin, Cout, 1, 1, 1, 0, C.c_void_p(taps.data_ptr()),
                                  C.c_void_p(ws.data_ptr()), ws.numel() * 4, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        part = ws[:ns * Cout * Cin].view(ns, Cout, Cin)
        exp = torch.full((ns,), float(chunk), device="cuda")
        if T % chunk:
            exp[-1] = T % chunk
        err = (part / 1e-3 - exp.view(ns, 1, 1)).abs().amax(dim=(1, 2))
        bad = (err > 1e-2).nonzero().flatten().tolist()
        tot_bad += len(bad)
        line = f"T={T} Cin={Cin} ns={ns} wgs={ns * ((Cout + 127) // 128)} rep={rep} rc={rc}: bad partials {len(bad)} {bad[:6]} max missing k-terms {float(err.max()):.2f}"
        if has_diag:
            d = (C.c_uint * (64 + 1024))()
            raw.loftr_hg_diag(d, 0)
            line += " | diag " + ", ".join(f"{KIND[k]}={d[k]} (blk {d[8 + 4 * k]} kt {d[9 + 4 * k]} tid {d[10 + 4 * k]})" for k in range(7) if d[k])
        print(line)
        if has_diag and d[7]:
            u = list(d[64:])
            print(f"   {d[7]} mismatching dwords; first ones (block kt tid wave lane | e hl q | lds_off want got | hw_id):")
            for k in range(min(d[7], 24)):
                r = u[8 * k: 8 * k + 8]
                print(f"     blk {r[0]:4d} kt {r[1]} tid {r[2]:3d} w{r[2] >> 6} l{r[2] & 63:2d} | e{r[3] >> 3} {'lo' if r[3] & 4 else 'hi'} q{r[3] & 3} | off {r[4]:5d} want {r[5]:08x} got {r[6]:08x} | {r[7]:08x}")
print("TOTAL bad partials", tot_bad)
