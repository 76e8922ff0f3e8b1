#!/usr/bin/env python
"""Kill test for a Winograd F(2x2, 3x3) backbone BEFORE any kernel is written (round-4 verdict, next #3): does its rounding noise fit
the image-level margin guard?  CPU only.  The PyTorch mirror of the backbone is run three ways on an image-level golden --
   direct : F.conv2d fp32 (what the reference does),
   wino   : every stride-1 3x3 convolution as F(2x2, 3x3) in float32 arithmetic the way a kernel would do it (input transform B^T d B
            in fp32, filter transform G g G^T in fp64 rounded once to fp32, 16 transformed-domain products accumulated in fp32 over the
            channels, output transform A^T m A in fp32),
   wino22 : the same with the transformed operands rounded to 22 mantissa bits (the split-fp16 product of csrc/gemm.h),
-- and the numpy oracle of the matching path finishes the forward.  Reported: distance of each result to the reference's FLOAT64 forward
(stored in the golden) next to the reference's own fp32 distance, max and RMS over the common matches: the quantities of the margin guard
(tests/test_e2e_golden.py).   python tools/micro/winograd_numerics.py [e2e_synth ...]"""
import copy, json, os, sys
import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("make_golden_e2e", os.path.join(ROOT, "tests", "golden", "make_golden_e2e.py"))
E2E = importlib.util.module_from_spec(spec); spec.loader.exec_module(E2E)
from oracle import loftr_oracle as O

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def round_bits(x, bits):
    """round an fp32 tensor to `bits` mantissa bits (round to nearest)"""
    m, e = torch.frexp(x.double())
    return (torch.round(m * 2.0 ** bits) / 2.0 ** bits * torch.pow(torch.tensor(2.0, dtype=torch.float64), e.double())).float()


def winograd_conv(x, w, bits=None):
    """x [B,C,H,W] fp32, w [O,C,3,3] fp32, padding 1, stride 1 -> [B,O,H,W] fp32"""
    B_, C, H, W = x.shape
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [B, C, nH, nW, 4, 4]
    bt = BT.float()
    v = torch.einsum("ij,bcyxjk->bcyxik", bt, t)                 # fp32 adds / subtracts only
    v = torch.einsum("bcyxik,lk->bcyxil", v, bt)
    u = (G @ w.double() @ G.t()).float()                         # [O, C, 4, 4], rounded once
    if bits:
        v, u = round_bits(v, bits), round_bits(u, bits)
    m = torch.einsum("bcyxil,ocil->boyxil", v, u)                # fp32 accumulation over c
    at = AT.float()
    y = torch.einsum("pi,boyxil->boyxpl", at, m)
    y = torch.einsum("boyxpl,ql->boyxpq", y, at)                 # [B, O, nH, nW, 2, 2]
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(B_, -1, Hp, Wp)
    return y[:, :, :H, :W].contiguous()


def run(name, mode):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz")))
    rc = json.loads(str(g["recipe"]))
    img0, img1 = E2E.images_from_golden(g)
    ex = E2E.extras(rc, img0, img1)
    from loftr_amd import LoFTR
    cfg = E2E.e2e_cfg(0.0, rc)
    model = LoFTR(copy.deepcopy(cfg)).eval()
    model.load_state_dict(E2E.e2e_state_dict(model, cfg, rc["bn_strength"], rc.get("coarse_gain", 1.0)), strict=True)
    if mode != "direct":
        bits = 22 if mode == "wino22" else None
        for mod in model.backbone.modules():
            if isinstance(mod, torch.nn.Conv2d) and mod.kernel_size == (3, 3) and mod.stride == (1, 1):
                mod.forward = (lambda x, mod=mod: winograd_conv(x, mod.weight, bits))
    with torch.no_grad():
        if img0.shape == img1.shape:
            fc, ff = model.backbone(torch.from_numpy(np.concatenate([img0, img1], 0)))
            n = img0.shape[0]
            fc0, fc1, ff0, ff1 = fc[:n], fc[n:], ff[:n], ff[n:]
        else:
            (fc0, ff0), (fc1, ff1) = model.backbone(torch.from_numpy(img0)), model.backbone(torch.from_numpy(img1))
    w = {k: v.numpy() for k, v in model.state_dict().items() if not k.startswith("backbone.")}
    out = O.loftr_hot_path(fc0.numpy(), fc1.numpy(), ff0.numpy(), ff1.numpy(), w, model.config, img0.shape[2:], img1.shape[2:],
                           **({k: v for k, v in ex.items()} if ex else {}))
    return out, g, (fc0.numpy(), ff0.numpy())


def dist(out, g, tag="ref64"):
    r = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith(tag + "/")}
    ko = {k: n for n, k in enumerate(zip(out["i_ids"].tolist(), out["j_ids"].tolist()))}
    com = [(ko[k], n) for n, k in enumerate(zip(r["i_ids"].tolist(), r["j_ids"].tolist())) if k in ko]
    io, ir = np.array([c[0] for c in com]), np.array([c[1] for c in com])
    dp = out["mkpts1_f"][io].astype(np.float64) - r["mkpts1_f"][ir]
    dc = out["mconf"][io].astype(np.float64) - r["mconf"][ir]
    return len(com), np.abs(dp).max(), np.sqrt((dp ** 2).mean()), np.abs(dc).max(), np.sqrt((dc ** 2).mean())


if __name__ == "__main__":
    torch.set_num_threads(16)
    for name in (sys.argv[1:] or ["e2e_synth"]):
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz")))
        ref32 = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith("thr0/")}
        n, pm, pr, cm, cr = dist(ref32, g)
        print(f"{name}: reference fp32 vs its fp64: common {n}, px max {pm:.2e} rms {pr:.2e}, conf max {cm:.2e} rms {cr:.2e}")
        feats = {}
        for mode in ("direct", "wino", "wino22"):
            out, g, f = run(name, mode)
            feats[mode] = f
            n, pm2, pr2, cm2, cr2 = dist(out, g)
            extra = ""
            if mode != "direct":
                extra = (f" | features vs direct: coarse {np.abs(f[0] - feats['direct'][0]).max() / np.abs(feats['direct'][0]).max():.1e}, "
                         f"fine {np.abs(f[1] - feats['direct'][1]).max() / np.abs(feats['direct'][1]).max():.1e} (rel. to max)")
            print(f"  {mode:7s}: common {n}, px max {pm2:.2e} ({pm2 / pm:.2f}x ref) rms {pr2:.2e} ({pr2 / pr:.2f}x), conf max {cm2:.2e} rms {cr2:.2e} ({cr2 / cr:.2f}x){extra}", flush=True)
