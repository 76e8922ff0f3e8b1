#!/usr/bin/env python
"""bench.py -- image-pairs/sec of the LoFTR forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): per GPU a batch of B=8 synthetic 640x480 grayscale pairs,
indoor dual-softmax config, seeded random-init weights (no checkpoints exist on the box).  With
random weights the stock thr=0.2 yields zero matches, which would skip the whole fine stage, so
the bench runs thr=0.0 (~0.7-1k mutual-NN matches per pair, the realistic load) and says so in
`config`.  One *step* = one full `LoFTR.forward(batch)`: PyTorch-ROCm backbone (fp32) + the HIP
matching path (fp32 MFMA) incl. materialising data['conf_matrix'], + (N>1) the RCCL all-gather
of the per-pair match counts.  Inputs are resident in HBM before the timed region.

Multi-GPU: pairs are independent -> each rank owns its own pairs; the only collective is the all-gather of
the per-pair match counts (RCCL through the library's C-ABI, loftr_rccl_allgather_counts).  `--scaling weak`
(default): B pairs per GPU (BASELINE configs[1] x N); `--scaling strong`: a fixed total of `--total-batch` 64 pairs
(BASELINE configs[2]) split over the ranks.  `python bench.py --gpus N` without a launcher spawns the N ranks
itself (torch.distributed.run on 127.0.0.1); under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.

The JSON line also carries
  roofline          -- the kernel group that dominates the hand-written matching path: the linear-attention encoder kernels
                       (encoder_x / proj_kv / fine_pair / linear) against the dense fp16 MFMA roof (executed MFMA rate,
                       hipEvents around every launch in serial instrumented steps of this run; `mfma_busy` =
                       SQ_VALU_MFMA_BUSY_CYCLES utilisation and `traffic` = HBM bytes from the rocprofv3 --pmc passes of
                       THIS build: profiles/pmc_traffic.json carries a hash of csrc/, null when it does not match);
                       `dominant_share_of_step` = its share of the matching path, `share_of_serial_step` = of the step;
  roofline_score_volume -- the north_star's score-volume kernel (score_conf_kernel, dual-softmax pass B) against the HBM
                       roof: algorithmic bytes per launch / hipEvent-measured average launch
                       duration (events recorded by the library on the launch stream INSIDE the timed region);
  roofline_backbone -- the convolution kernels of the ResNet-FPN (the largest share of the whole step), same form;
  kernels           -- the same for every instrumented kernel (serial instrumented steps), backbone included;
  cpu_baseline      -- kind "reference": zju3dv/LoFTR's own LoFTR.forward on the host cores of this box in this run (imported
                       from /root/reference or from the bytecode bundle oracle/stage_ref.py made of it), pair 0 of the GPU
                       batch, the GPU model's weights, + `parity_vs_reference` of the GPU result for that pair; kind "port"
                       (torch-CPU backbone + numpy oracle) with the reason when the reference cannot be imported.  Thread
                       count by probe, 1 warm-up + median of 3 (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from loftr_amd import LoFTR, get_cfg, _lib          # noqa: E402
from loftr_amd.distributed import all_gather_match_counts, RcclCounts   # noqa: E402
from loftr_amd.synth import make_images, make_weights, make_backbone_weights   # noqa: E402

H_IMG, W_IMG = 480, 640
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak (for context only)
SPLIT_FACTOR = 3               # every fp32 product = 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi), csrc/gemm.h
GEMM_KERNELS = ("proj_kernel", "proj_kv_kernel", "linear_kernel", "linear_ln_kernel", "score_sweep_kernel<0>", "score_sweep_kernel<1>",
                "conv_kernel", "conv3x3_duo_kernel<Cfg<4,2,4,4,1>>", "conv3x3_duo_kernel<Cfg<7,2,4,8,2>>", "encoder_x_kernel", "fine_pair_kernel")

K_IDS = {}


def kernel_ids(lib):
    if not K_IDS:
        for i in range(lib.loftr_hip_timing_kernel_count()):
            K_IDS[lib.loftr_hip_timing_kernel_name(i).decode()] = i
    return K_IDS


def read_timing(lib, kid, reset=True):
    ms, n = C.c_double(0), C.c_longlong(0)
    _lib.check(lib.loftr_hip_timing_read(kid, C.byref(ms), C.byref(n), int(reset)))
    return ms.value, n.value


def algorithmic_work(B, L, S, M, C_=256, Cf=128, WW=25, fused=True, fused_fine=True, persistent=False):
    """Algorithmic (flops, bytes) per STEP of every instrumented kernel for a batch of B pairs with
    M matches in total.  fp32 everywhere: 4 bytes per element.  fused: the x side of a coarse
    layer runs as encoder_x_kernel (csrc/encoder_fused.hip) instead of proj / linear_ln / linear / linear_ln; fused_fine: the whole
    fine-level transformer runs as fine_pair_kernel (csrc/fine_fused.hip) instead of proj / attn_small / linear_ln / linear / linear_ln."""
    w = {}
    rows_c = 2 * B * L                      # both images, L == S
    n_self = n_cross = 4

    def enc(rows, c, layers, fused_kv=False):
        # per encoder-layer pass over `rows` tokens of width c
        proj = (2 * rows * c * 3 * c, 4 * (rows * c + 3 * c * c + 3 * rows * c))
        if fused_kv:      # coarse level: proj_kernel computes q only; k, v (+ the KV reduction) live in proj_kv_kernel
            proj = (2 * rows * c * c, 4 * (rows * c + c * c + rows * c))
        mlp0 = (2 * rows * (2 * c) * (2 * c), 4 * (2 * rows * c + 4 * c * c + 2 * rows * c))
        ln = (2 * rows * c * c + 2 * rows * 2 * c * c,
              4 * (rows * c + c * c + rows * c) + 4 * (2 * rows * c + 2 * c * c + 2 * rows * c))
        return {k: (v[0] * layers, v[1] * layers) for k, v in dict(proj_kernel=proj, linear_kernel=mlp0,
                                                                     linear_ln_kernel=ln).items()}
    coarse = enc(rows_c, C_, n_self + n_cross, fused_kv=True)
    fine = enc(2 * M * WW, Cf, 2)
    if fused:
        nlc = n_self + n_cross
        # q + merge (2 C^2), mlp.0 (4 C^2), mlp.2 (2 C^2) per token; x (SP + fp32) in, out (fp32 + SP) in place, 2 MB of weights per call
        w["encoder_x_kernel"] = (nlc * 2 * rows_c * 8 * C_ * C_, nlc * 4 * (4 * rows_c * C_ + 8 * C_ * C_))
        coarse = {k: (0, 0) for k in coarse}
    w["attn_small_kernel"] = (2 * 2 * (2 * 2 * M * WW * Cf * 16), 2 * 4 * 4 * 2 * M * WW * Cf)
    if fused_fine:
        # 4 encoder calls per match (2 layers x 2 windows): q, k, v, merge (4 Cf^2), mlp.0 (4 Cf^2), mlp.2 (2 Cf^2) per token + the
        # attention itself; both windows in and out as fp32, once
        fl = sum(v[0] for v in fine.values()) + w["attn_small_kernel"][0]
        w["fine_pair_kernel"] = (fl, 4 * (2 * 2 * M * WW * Cf) + 2 * 4 * 10 * Cf * Cf)
        fine = {k: (0, 0) for k in fine}
        w["attn_small_kernel"] = (0, 0)
    for k in coarse:
        w[k] = (coarse[k][0] + fine[k][0], coarse[k][1] + fine[k][1])
    # fine preprocess linears ride on linear_kernel: down_proj, ctx, window merge (x2 sides)
    fp_f = 2 * (2 * M * C_ * Cf + 2 * M * Cf * Cf + 2 * M * WW * Cf * Cf)
    fp_b = 2 * 4 * (M * C_ + Cf * C_ + M * Cf + M * Cf + Cf * Cf + M * Cf + 2 * M * WW * Cf + Cf * Cf)
    w["linear_kernel"] = (w["linear_kernel"][0] + fp_f, w["linear_kernel"][1] + fp_b)
    nl = n_self + n_cross
    w["proj_kv_kernel"] = (nl * (2 * rows_c * C_ * 2 * C_ + 2 * rows_c * C_ * 32), nl * 4 * (rows_c * C_ + 2 * C_ * C_))
    if persistent and fused:
        # round 6: the coarse transformer as ONE persistent launch (csrc/encoder_fused.hip: coarse_persistent_kernel) -- its K items are
        # proj_kv_kernel's work (k / v projection + the K^T V reduction), timed in the encoder_x slot together with the X items
        w["encoder_x_kernel"] = (w["encoder_x_kernel"][0] + w["proj_kv_kernel"][0], w["encoder_x_kernel"][1] + w["proj_kv_kernel"][1])
        w["proj_kv_kernel"] = (0, 0)
    w["score_sweep_kernel<0>"] = (2 * B * L * S * C_, 4 * B * (L + S) * C_)
    w["score_sweep_kernel<1>"] = (2 * B * L * S * C_, 4 * B * ((L + S) * C_ + L * S))
    w["gather_windows_kernel"] = (0, 2 * 4 * 2 * M * WW * Cf)
    # backbone convolutions on the same GEMM core (conv.hip): ResNetFPN_8_2 over 2B images of 480x640
    # 3x3 stride-1 layers run the patch-in-LDS kernels (the 224-column one when ceil32(Cout) == 224)
    acc = {"conv_kernel": [0, 0], "conv3x3_duo_kernel<Cfg<4,2,4,4,1>>": [0, 0], "conv3x3_duo_kernel<Cfg<7,2,4,8,2>>": [0, 0]}
    for (cin, cout, k, stride, hin, win) in backbone_convs(H_IMG, W_IMG):
        ho, wo = hin // stride, win // stride
        patch = k == 3 and stride == 1
        a = acc[("conv3x3_duo_kernel<Cfg<7,2,4,8,2>>" if (cout + 31) // 32 == 7 else "conv3x3_duo_kernel<Cfg<4,2,4,4,1>>") if patch else "conv_kernel"]
        a[0] += 2 * 2 * B * ho * wo * cout * cin * k * k
        a[1] += 4 * 2 * B * (hin * win * cin + ho * wo * cout) + 4 * cout * cin * k * k
    for name, (fl, by) in acc.items():
        w[name] = (fl, by)
    return w


def backbone_convs(h, w):
    """(cin, cout, k, stride, h_in, w_in) of every convolution conv.hip runs (resnet_fpn.py:43-118 minus the stem)."""
    h2, w2, h4, w4, h8, w8 = h // 2, w // 2, h // 4, w // 4, h // 8, w // 8
    L = []
    L += [(128, 128, 3, 1, h2, w2)] * 4                                                   # layer1
    L += [(128, 196, 3, 2, h2, w2), (128, 196, 1, 2, h2, w2)] + [(196, 196, 3, 1, h4, w4)] * 3   # layer2
    L += [(196, 256, 3, 2, h4, w4), (196, 256, 1, 2, h4, w4)] + [(256, 256, 3, 1, h8, w8)] * 3   # layer3
    L += [(256, 256, 1, 1, h8, w8), (196, 256, 1, 1, h4, w4), (256, 256, 3, 1, h4, w4), (256, 196, 3, 1, h4, w4)]
    L += [(128, 196, 1, 1, h2, w2), (196, 196, 3, 1, h2, w2), (196, 128, 3, 1, h2, w2)]
    return L


def roofline_entry(name, total_ms, launches, flops, nbytes, steps):
    """Roofline of one kernel.  `flops` / `nbytes` are ALGORITHMIC per step.  For the GEMM
    kernels the matrix-core figure is the EXECUTED fp16 MFMA rate (3 MFMAs per fp32 product) against the
    dense fp16 peak; the algorithmic fp32-equivalent rate and its ratio to the fp32-MFMA peak (what a
    plain fp32 implementation is bounded by) are reported next to it."""
    if launches == 0 or total_ms <= 0:
        return None
    per_launch_ms = total_ms / launches
    t = total_ms / steps * 1e-3                      # seconds per step spent in this kernel
    gbs, tf = nbytes / t / 1e9, flops / t / 1e12
    exec_tf = tf * (SPLIT_FACTOR if name in GEMM_KERNELS else 1)
    peak_tf = MFMA_F16_PEAK_TF if name in GEMM_KERNELS else MFMA_F32_PEAK_TF
    f_h, f_m = gbs / HBM_PEAK_GBS, exec_tf / peak_tf
    bound = "hbm" if f_h >= f_m else "mfma"
    e = {"kernel": name, "bound": bound, "achieved": round(gbs if bound == "hbm" else exec_tf, 2),
         "peak": HBM_PEAK_GBS if bound == "hbm" else peak_tf,
         "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(max(f_h, f_m), 4), "traffic": None,
         "avg_launch_us": round(per_launch_ms * 1e3, 2), "launches_per_step": round(launches / steps, 2),
         "ms_per_step": round(total_ms / steps, 4), "alg_GB_s": round(gbs, 1), "hbm_frac": round(f_h, 4),
         "alg_TFLOP_s": round(tf, 2), "mfma_frac": round(f_m, 4)}
    if name in GEMM_KERNELS:
        e["executed_fp16_TFLOP_s"] = round(exec_tf, 1)
        e["x_fp32_mfma_peak"] = round(tf / MFMA_F32_PEAK_TF, 3)
    return e


def source_hash():
    """sha256 over the kernel sources the library is built from (csrc/*.hip, *.h + the C header), 16 hex digits.
    tools/rocpd_pmc.py stamps the same hash into profiles/pmc_traffic.json when the PMC passes are collected."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "loftr_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "loftr_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def pmc_kernel_code_hash(table_keys, lib_path=None):
    """Identity of the MACHINE CODE of the kernels a PMC table describes: sha256 over the gfx950 function bodies (tools/kernel_code_hash.py, pure
    Python over the library's offload bundles) of every device function whose name contains the base name of a table entry
    ("conv3x3_duo_kernel<Cfg<4,2,4,4,1>>" -> "conv3x3_duo", "encoder_x_kernel" -> "encoder_x": the pooled variants are included)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_code_hash", os.path.join(ROOT, "tools", "kernel_code_hash.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    subs = sorted({k.split("<")[0][:-len("_kernel")] if k.split("<")[0].endswith("_kernel") else k.split("<")[0] for k in table_keys if not k.startswith("_")})
    return m.kernel_code_hash(lib_path or _lib.LIB_PATH, subs)[0]


_PMC = None
PMC_IDENTITY = None          # "source" | "kernel_code" | None: which identity tied profiles/pmc_traffic.json to this build


def pmc_table():
    """profiles/pmc_traffic.json (rocprofv3 --pmc passes: FETCH_SIZE x2 + WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES) if it was collected on THIS
    build of the kernels it describes, else {}.  "This build": the hash of every kernel source matches, or -- round 5: a training-only
    translation unit must not invalidate the forward kernels' counters -- the machine code of the kernels in the table is byte-identical
    (pmc_kernel_code_hash)."""
    global _PMC, PMC_IDENTITY
    if _PMC is None:
        _PMC = {}
        p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            t = json.load(open(p))
            meta = t.get("_meta", {})
            if meta.get("source_hash") == source_hash():
                _PMC, PMC_IDENTITY = t, "source"
            elif meta.get("kernel_code_hash") and meta["kernel_code_hash"] == pmc_kernel_code_hash(t.keys()):
                _PMC, PMC_IDENTITY = t, "kernel_code"
        except Exception:
            pass
    return _PMC


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the PMC passes of this build, or None."""
    return pmc_table().get(kernel, {}).get("hbm_bytes_per_launch")


def pmc_mfma_busy(kernel):
    return pmc_table().get(kernel, {}).get("mfma_busy")


def reference_cpu_forward(model, img0, img1, gpu_data):
    """The reference's OWN `LoFTR.forward` (zju3dv/LoFTR src/loftr/loftr.py:29-75, its ResNet-FPN included) on the host cores of
    THIS box, in THIS run, on pair 0 of the GPU batch with the GPU model's weights (north_star: "the reference's CPU forward()
    timed on the host cores of the same box (core count stated) in the same run").  The reference is imported through
    oracle/ref_shim.py: from /root/reference where that exists, otherwise from the bytecode bundle oracle/stage_ref.py compiled
    from it (oracle/_ref/, travels with the snapshot).  Also returns the live parity of the GPU result for that pair against it."""
    import copy
    from oracle import ref_shim
    mode = ref_shim.reference_mode()
    if mode is None:
        return None, "no reference checkout and no staged bundle (oracle/_ref/loftr_reference.bundle) on this machine"
    RefLoFTR, _ = ref_shim.import_reference()
    ref = RefLoFTR(copy.deepcopy(model.config)).eval()
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    x0, x1 = img0[:1].cpu(), img1[:1].cpu()
    cores = os.cpu_count()
    probe = {}
    with torch.no_grad():
        for nt in [t for t in (16, 32, 64, 128) if t <= cores] or [cores]:       # a 256-thread pool is 3-5x slower than the best
            torch.set_num_threads(nt)
            ref.backbone(torch.cat([x0, x1], 0))
            t0 = time.perf_counter()
            ref.backbone(torch.cat([x0, x1], 0))
            probe[nt] = time.perf_counter() - t0
        threads = min(probe, key=probe.get)
        torch.set_num_threads(threads)
        bb = {}
        h0 = ref.backbone.register_forward_pre_hook(lambda m, a: bb.__setitem__("t0", time.perf_counter()))
        h1 = ref.backbone.register_forward_hook(lambda m, a, o: bb.__setitem__("dt", time.perf_counter() - bb["t0"]))
        runs = []
        for it in range(4):                                                       # 1 warm-up + 3 timed
            data = {"image0": x0.clone(), "image1": x1.clone()}
            t0 = time.perf_counter()
            ref(data)
            runs.append((time.perf_counter() - t0, bb["dt"]))
        h0.remove(); h1.remove()
    runs = sorted(runs[1:])
    total, bbt = runs[1]
    # live parity of the GPU forward (last timed step) against this reference forward: pair 0 (the timed sample) and the LAST pair of the
    # batch (one more reference forward; a batch-position mistake on the GPU side would show there and not on pair 0)
    g = gpu_data

    def pair_parity(b, rd):
        sel = (g["b_ids"] == b).nonzero().squeeze(1)
        msel = (g["m_bids"] == b).nonzero().squeeze(1)
        gi, gj = g["i_ids"][sel].cpu().numpy(), g["j_ids"][sel].cpu().numpy()
        ri, rj = rd["i_ids"].numpy(), rd["j_ids"].numpy()
        rk = {k: n for n, k in enumerate(zip(ri.tolist(), rj.tolist()))}
        com = [(n, rk[k]) for n, k in enumerate(zip(gi.tolist(), gj.tolist())) if k in rk]
        ia, ib = [c[0] for c in com], [c[1] for c in com]
        par = {"pair": int(b), "matches_gpu": int(len(gi)), "matches_reference": int(len(ri)), "common": len(com)}
        if com:
            gm = {k: g[k][msel].cpu().numpy() for k in ("mconf", "mkpts0_f", "mkpts1_f")}
            par.update(d_mconf=float(np.abs(gm["mconf"][ia] - rd["mconf"].numpy()[ib]).max()),
                       d_mkpts0_f_px=float(np.abs(gm["mkpts0_f"][ia] - rd["mkpts0_f"].numpy()[ib]).max()),
                       d_mkpts1_f_px=float(np.abs(gm["mkpts1_f"][ia] - rd["mkpts1_f"].numpy()[ib]).max()))
        if "conf_matrix" in g and g["conf_matrix"] is not None:
            par["d_conf_matrix"] = float((g["conf_matrix"][b].cpu() - rd["conf_matrix"][0]).abs().max())
        return par
    ri = data["i_ids"].numpy()
    pars = [pair_parity(0, data)]
    last = int(img0.shape[0]) - 1
    if last > 0:
        dl = {"image0": img0[last:last + 1].cpu().clone(), "image1": img1[last:last + 1].cpu().clone()}
        with torch.no_grad():
            ref(dl)
        pars.append(pair_parity(last, dl))
    par = {"pairs": pars,
           "note": "GPU forward (HIP backbone + HIP matching path) vs the reference's fp32 CPU forward on the same pair and weights in this run, for the "
                   "first and the last pair of the batch; match sets can differ by near-tie flips at thr 0 with random weights (the goldens pin this "
                   "against the reference's fp64 run: profiles/r06_parity_margins.txt)"}
    out = {"value": round(1.0 / total, 4), "unit": "image-pairs/s", "cores": threads, "kind": "reference", "reference_import": mode,
           "reference_forward_s": round(total, 3), "reference_backbone_s": round(bbt, 3), "reference_hot_path_s": round(total - bbt, 3),
           "sample": f"pair 0 of the GPU batch (640x480), zju3dv/LoFTR LoFTR.forward (eval, no_grad, fp32, torch CPU), the GPU model's weights: "
                     f"1 warm-up + median of 3 forwards; {threads} torch threads chosen by a backbone probe "
                     f"{ {k: round(v, 2) for k, v in probe.items()} } s, host has {cores} logical cores; M={len(ri)}",
           "host_cores": cores, "runs_s": [round(r[0], 3) for r in runs], "parity_vs_reference": par}
    return out, None


def cpu_baseline(model, img0, img1, gpu_data=None):
    """`cpu_baseline` of the JSON line.  kind "reference" when the reference can be imported on this box (see
    reference_cpu_forward); otherwise kind "port" -- torch-CPU backbone (the reference's backbone is this very PyTorch module)
    + oracle/loftr_oracle.py (numpy restatement of the matching path) -- with the reason.  BASELINE.md §3 protocol either way:
    eval / no_grad / fp32, thread count picked by a short probe, 1 warm-up + 3 timed forwards, median."""
    why = None
    if gpu_data is not None:
        try:
            ref, why = reference_cpu_forward(model, img0, img1, gpu_data)
            if ref is not None:
                return ref
        except Exception as e:                                # noqa: BLE001  (fall back to the port, say why)
            why = f"reference leg failed: {e!r}"
        finally:
            model.to(img0.device)
    from oracle import loftr_oracle as O
    cores = os.cpu_count()
    cpu_model = model.backbone.to("cpu").float()
    w = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if not k.startswith("backbone.")}
    x = torch.cat([img0[:1].cpu(), img1[:1].cpu()], 0)

    def backbone():
        with torch.no_grad():
            fc, ff = cpu_model(x)
        return fc.numpy(), ff.numpy()

    probe = {}
    for nt in [t for t in (16, 32, 64, 128) if t <= cores] or [cores]:
        torch.set_num_threads(nt)
        backbone()                                            # warm the thread pool / oneDNN primitives at this size
        t0 = time.perf_counter()
        backbone()
        probe[nt] = time.perf_counter() - t0
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)

    def forward():
        t0 = time.perf_counter()
        fc, ff = backbone()
        t1 = time.perf_counter()
        out = O.loftr_hot_path(fc[:1], fc[1:], ff[:1], ff[1:], w, model.config, (H_IMG, W_IMG), (H_IMG, W_IMG))
        t2 = time.perf_counter()
        return t2 - t0, t1 - t0, t2 - t1, len(out["mconf"])

    forward()                                                 # warm-up
    runs = sorted(forward() for _ in range(3))
    total, bb, hot, m = runs[1]                               # median by total time
    model.backbone.to(img0.device)
    return {"value": round(1.0 / total, 4), "unit": "image-pairs/s", "cores": threads, "kind": "port",
            "kind_reason": why or "reference leg not requested",
            "sample": f"pair 0 of the GPU batch (640x480), 1 warm-up + median of 3 forwards: torch-CPU backbone {bb:.2f}s + "
                      f"numpy oracle matching path {hot:.2f}s (M={m}); {threads} torch threads chosen by probe "
                      f"{ {k: round(v, 2) for k, v in probe.items()} } s/backbone, host has {cores} logical cores",
            "host_cores": cores, "backbone_s": round(bb, 3), "hot_path_s": round(hot, 3),
            "runs_s": [round(r[0], 3) for r in runs]}


BACKBONE_KERNELS = ("conv3x3_duo_kernel<Cfg<4,2,4,4,1>>", "conv3x3_duo_kernel<Cfg<7,2,4,8,2>>", "conv_kernel", "conv3x3s2_kernel")
ENCODER_KERNELS = ("proj_kv_kernel", "proj_kernel", "linear_kernel", "linear_ln_kernel", "encoder_x_kernel", "fine_pair_kernel")
# The north_star kernels carry hipEvents inside the timed region: the score-volume kernel (one launch per step) and, since round 6, the
# encoder group (8 launches per step with the persistent coarse transformer; 34 as launches); everything else is measured in the
# serial instrumented steps just before it (round-2 verdict: 83 event pairs per step in the timed region).
NORTH_STAR_TIMED = ("score_sweep_kernel<1>",)
# The library times kernel FAMILIES (one hipEvent slot per family, csrc/common.h: LoftrTimedKernel); a slot is named after the kernel the
# default path launches, and these are the names `rocprofv3 --kernel-trace --stats` prints for everything pooled in it.
POOLED_FROM = {
    "score_sweep_kernel<0>": ["sweep::score_sweep_kernel<0, false, true>", "sweep::score_sweep_kernel<0, false, false>", "sweep::score_sweep_kernel<0, true, false>",
                              "(C != 256: score_stats_kernel)"],
    "score_sweep_kernel<1>": ["sweep::score_sweep_kernel<1, false, false, false>", "sweep::score_sweep_kernel<1, false, false, true>",
                              "sweep::score_sweep_kernel<1, true, false, true>", "(C != 256: score_conf_kernel)"],
    "score_sweep_kernel<2>": ["sweep::score_sweep_kernel<2, false>", "sweep::score_sweep_kernel<2, true>", "(C != 256: score_store_kernel)"],
    "conv3x3_duo_kernel<Cfg<4,2,4,4,1>>": ["conv3x3_duo_kernel<c3d::Cfg<4, 2, 4, 4, 1> >", "conv3x3_duo_kernel<c3d::Cfg<6, 2, 4, 8, 2> > (Coutp = 192; not on the LoFTR path)",
                                           "(Cout not a multiple of 128 / 192 / 224: conv3x3_kernel)"],
    "conv3x3_duo_kernel<Cfg<7,2,4,8,2>>": ["conv3x3_duo_kernel<c3d::Cfg<7, 2, 4, 8, 2> >"],
    "encoder_x_kernel": ["efx::coarse_persistent_kernel", "efx::encoder_x_kernel", "efx::encoder_x2_kernel"],
    "fine_pair_kernel": ["ffx::fine_pair_kernel"],
    "conv_kernel": ["conv_kernel<GemmCfg<...>, false> (strided 3x3, 1x1)", "conv_kernel<GemmCfg<...>, true> (FPN top-down: 1x1 + bilinear x2 + add)"],
    "proj_kernel": ["proj_kernel"], "linear_ln_kernel": ["linear_ln_kernel"],
}


def run_with_retry(argv):
    """OPT-IN (LOFTR_BENCH_RETRY=1; off by default since round 4 -- a retry can turn an intermittent kernel fault into a passing
    headline number): run the measurement in a CHILD process and repeat it ONCE, in a fresh process / HIP context, if the child is
    killed by a signal (a GPU memory fault makes the ROCm runtime abort(): SIGABRT) -- seen twice in round 3 on single boxes with commands
    that passed on every other box, no reproducer (docs/HISTORY.md).  An ordinary non-zero exit (Python exception) is NOT retried.
    The JSON line records the attempt when it is not the first."""
    env = dict(os.environ, LOFTR_BENCH_CHILD="1")
    for attempt in (1, 2):
        env["LOFTR_BENCH_ATTEMPT"] = str(attempt)
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=subprocess.PIPE)
        if p.returncode >= 0 or attempt == 2:
            sys.stdout.write(p.stdout.decode(errors="replace"))
            sys.stdout.flush()
            return p.returncode if p.returncode >= 0 else 128 - p.returncode
        print(f"[bench] attempt {attempt} was killed by signal {-p.returncode}; one more try in a fresh process", file=sys.stderr)


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_spawn(argv, n):
    """`python bench.py --gpus N` without a launcher: start the N ranks exactly as the driver's launcher would."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def group_roofline(names, timing, work, steps):
    """One MFMA-roof entry over a set of GEMM kernels (the encoder group, the backbone): `achieved` / `frac` = summed ALGORITHMIC fp32
    flops / summed launch time against the dense fp16 MFMA peak (the task's definition; round-5 verdict, weak #3), `executed_*` = the
    fp16 MFMAs the matrix pipe actually runs (3 per fp32 product, csrc/gemm.h) -- its ceiling for algorithmic work is 1 / 3."""
    ms = sum(timing[n][0] for n in names if n in timing)
    launches = sum(timing[n][1] for n in names if n in timing)
    if ms <= 0 or not launches:
        return None
    fl = sum(work[n][0] for n in names if n in timing and n in work)
    by = sum(work[n][1] for n in names if n in timing and n in work)
    t = ms / steps * 1e-3
    alg_tf = fl / t / 1e12
    exec_tf = alg_tf * SPLIT_FACTOR
    traffic = [pmc_traffic(n) for n in names if n in timing]
    busy = [(pmc_mfma_busy(n), timing[n][0]) for n in names if n in timing]
    e = {"kernels": [n for n in names if n in timing], "bound": "mfma", "achieved": round(alg_tf, 2), "peak": MFMA_F16_PEAK_TF,
         "unit": "TFLOP/s", "frac": round(alg_tf / MFMA_F16_PEAK_TF, 4),
         "executed_TFLOP_s": round(exec_tf, 1), "executed_frac": round(exec_tf / MFMA_F16_PEAK_TF, 4),
         "ms_per_step": round(ms / steps, 4),
         "launches_per_step": round(launches / steps, 2), "alg_TFLOP_s": round(alg_tf, 2),
         "alg_GB_s": round(by / t / 1e9, 1), "hbm_frac": round(by / t / 1e9 / HBM_PEAK_GBS, 4),
         "traffic": None, "mfma_busy": None,
         "note": "achieved / frac = ALGORITHMIC fp32 flops of these kernels / their launch time against the 2.5 PF dense fp16 peak; executed_* = "
                 "3 fp16 MFMAs per fp32 product (csrc/gemm.h: fp32-accurate products on the fp16 matrix cores; ceiling for algorithmic work = 1/3); "
                 "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), time-weighted over the kernels"}
    if all(x is not None for x in traffic) and traffic:
        # bytes per STEP: per-launch PMC bytes x launches per step of each kernel
        e["traffic"] = round(sum(pmc_traffic(n) * timing[n][1] / steps for n in names if n in timing))
        e["traffic_note"] = "HBM bytes per step over these kernels (PMC per-launch bytes x launches per step)"
    if all(b is not None for b, _ in busy) and busy:
        e["mfma_busy"] = round(sum(b * w for b, w in busy) / sum(w for _, w in busy), 4)
    return e


def other_configs(lib, ids, dev, sd, backbone, overlap=True, steps=5, warmup=2):
    """BASELINE configs[3] (MegaDepth outdoor_ds: 840 x 840 pairs zero-padded from 840 x 560, coarse padding masks, scale0 / scale1,
    L = S = 11 025, /root/reference configs/loftr/outdoor/loftr_ds.py:1-5) and configs[4] (indoor_ot: Sinkhorn matching,
    configs/loftr/indoor/loftr_ot.py:1-3) through the SAME library, after the headline's timed region and outside it:
    a few full forwards each (same seeded weights, thr 0.0), plus two instrumented ones for the score-volume kernels."""
    out = {}

    def run(tag, cfg, batch_fn, n_pairs, L, extra, impl=None, strict=True, n_steps=None):
        model = LoFTR(cfg).eval()
        model.load_state_dict(dict(sd), strict=strict)
        model = model.to(dev)
        model.backbone_impl = impl or backbone
        model.overlap_fine_branch = overlap
        n_steps = n_steps or steps
        for _ in range(warmup):
            d = batch_fn(); model(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_steps):
            d = batch_fn(); model(d)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n_steps
        M = int(d["mconf"].shape[0])
        names = [n for n in ("score_sweep_kernel<1>", "score_sweep_kernel<0>", "score_sweep_kernel<2>", "encoder_x_kernel", "proj_kv_kernel", "fine_pair_kernel") if n in ids]
        mask = 0
        for n in names:
            mask |= 1 << ids[n]; read_timing(lib, ids[n])
        lib.loftr_hip_timing_enable(mask)
        for _ in range(2):
            d = batch_fn(); model(d)
        torch.cuda.synchronize()
        lib.loftr_hip_timing_enable(0)
        ent = {"pairs_per_s": round(n_pairs / ms * 1e3, 2), "ms_per_step": round(ms, 3), "pairs_per_step": n_pairs, "matches_per_pair": round(M / n_pairs, 1),
               "L": L, "steps": n_steps, "backbone_impl": model.backbone_impl}
        ent.update(extra)
        kt, kl = {}, {}
        for n in names:
            t, c = read_timing(lib, ids[n])
            if c:
                kt[n] = round(t / c * 1e3, 1)
                kl[n] = c // 2
        ent["kernel_avg_us"] = kt
        ent["kernel_launches_per_step"] = kl
        if "score_sweep_kernel<1>" in kt:                       # dual-softmax pass B against the HBM roof at THIS L
            by = 4 * n_pairs * (2 * L * 256 + L * L)
            gbs = by / (kt["score_sweep_kernel<1>"] * 1e-6) / 1e9
            ent["score_conf_roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes": by}
        out[tag] = ent
        del model
        torch.cuda.empty_cache()

    # ---- configs[0] on the GPU: ONE 640 x 480 pair per forward (notebooks/demo_single_pair.ipynb cells 3-4; demo/demo_loftr.py:150-158
    # is a batch-1 stream) -- the latency every interactive caller of the reference sees.  The coarse transformer has 38 token tiles per
    # sequence here (76 workgroups in a self call, 38 in a cross call) for 256 CUs: ops.COARSE_MODE "auto" runs it as launches
    # (a chain of 12 dependent calls; the persistent form wins from ~150 tiles per call on).
    cfg = get_cfg(thr=0.0)
    cfg["coarse"]["temp_bug_fix"] = True
    s0, s1 = make_images(1234, 1, H_IMG, W_IMG)
    s0, s1 = torch.from_numpy(s0).to(dev), torch.from_numpy(s1).to(dev)
    Lc = (H_IMG // 8) * (W_IMG // 8)
    tiles = (Lc + 127) // 128
    run("single_pair_640", cfg, lambda: {"image0": s0, "image1": s1}, 1, Lc,
        {"workload": "BASELINE configs[0] on the GPU: one 640x480 pair per forward (batch 1), dual-softmax, thr 0.0; ms_per_step = latency of a pair",
         "encoder_occupancy": {"token_tiles_per_sequence": tiles, "workgroups_per_self_call": 2 * tiles, "workgroups_per_cross_call": tiles,
                               "cus": 256, "cu_occupancy_self": round(2 * tiles / 256, 3), "cu_occupancy_cross": round(tiles / 256, 3)}},
        n_steps=4 * steps)
    # ---- the north_star's literal split: ResNet-FPN in PyTorch-ROCm (MIOpen fp32, channels-last) + the HIP matching path
    if backbone == "hip":
        a0, a1 = make_images(1234, 8, H_IMG, W_IMG)
        a0, a1 = torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)
        run("torch_backbone", cfg, lambda: {"image0": a0, "image1": a1}, 8, Lc,
            {"workload": "the headline step (8 pairs 640x480, dual-softmax) with backbone_impl='torch': the backbone stays in PyTorch-ROCm as north_star "
                         "words it; the headline runs this library's HIP convolutions instead"}, impl="torch")
        del a0, a1
    # ---- configs[3]: outdoor
    N = 2
    cfg = get_cfg(thr=0.0, border_rm=2)
    cfg["coarse"]["temp_bug_fix"] = False                    # configs/loftr/outdoor/buggy_pos_enc/loftr_ds.py:3-4 (outdoor_ds.ckpt)
    o0, o1 = make_images(1234, N, 840, 840)               # the images of tests/golden/e2e_outdoor_840.npz (pinned there against the reference from images)
    i0, i1 = torch.from_numpy(o0), torch.from_numpy(o1)
    i0[:, :, 560:] = 0; i1[:, :, 560:] = 0
    mask = torch.zeros(N, 105, 105, dtype=torch.bool); mask[:, :70] = True
    fixed = {"image0": i0.to(dev), "image1": i1.to(dev), "mask0": mask.to(dev), "mask1": mask.to(dev),
             "scale0": torch.full((N, 2), 1.9).to(dev), "scale1": torch.full((N, 2), 1.9).to(dev)}
    run("outdoor_840_masked", cfg, lambda: dict(fixed), N, 105 * 105,
        {"workload": "BASELINE configs[3]: 2 pairs 840x840 (valid 840x560, zero-padded), mask0/1 [2,105,105], scale 1.9, dual-softmax, thr 0.0",
         "parity": "the same images, masks, scales and configuration are pinned against the reference's forward from images (with the goldens' seeded backbone weights, "
                   "not this run's): tests/golden/e2e_outdoor_840.npz, tests/test_e2e_golden.py"})
    del fixed
    # ---- configs[4]: indoor_ot
    B = 8
    cfg = get_cfg(thr=0.0)
    cfg["coarse"]["temp_bug_fix"] = True
    cfg["match_coarse"].update(match_type="sinkhorn", skh_prefilter=False, sparse_spvs=True)
    a0, a1 = make_images(1234, B, H_IMG, W_IMG)
    a0, a1 = torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)
    run("indoor_ot", cfg, lambda: {"image0": a0, "image1": a1}, B, (H_IMG // 8) * (W_IMG // 8),
        {"workload": "BASELINE configs[4]: 8 pairs 640x480, match_type sinkhorn (3 iterations, conf_matrix_with_bin), thr 0.0"},
        strict=False)                                          # (bin_score keeps its constructor value 1.0: not in the seeded state dict)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU per step (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch pairs per GPU (configs[1] x N); strong: --total-batch pairs split over the ranks (configs[2])")
    ap.add_argument("--total-batch", type=int, default=64, help="pairs per step over ALL GPUs with --scaling strong")
    ap.add_argument("--thr", type=float, default=0.0)
    ap.add_argument("--no-conf", action="store_true", help="elide data['conf_matrix'] (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the untimed outdoor / Sinkhorn forwards reported as other_configs")
    ap.add_argument("--no-overlap", action="store_true", help="run the FPN fine branch on the main stream (no second HIP stream)")
    ap.add_argument("--match-type", default="dual_softmax", choices=["dual_softmax", "sinkhorn"],
                    help="sinkhorn = BASELINE configs[4] (indoor_ot); not the headline")
    ap.add_argument("--backbone", default="hip", choices=["hip", "torch"],
                    help="hip: implicit-GEMM convolutions of this library (default; image-level parity with the reference held at "
                         "1e-4 / 1e-3 px, profiles/r05_parity_margins.txt); torch: PyTorch-ROCm / MIOpen fp32")
    ap.add_argument("--backbone-halves", type=int, default=None, help="1 / 0: image0 / image1 batches through the backbone on two side streams (default: the model's)")
    ap.add_argument("--coarse-mode", default=None, choices=["auto", "launches", "persistent"],
                    help="coarse transformer as per-call launches or as the persistent work-queue kernel (default: the model's rule -- launches while a "
                         "second stream shares the GPU, persistent otherwise)")
    ap.add_argument("--debug-switch", action="append", default=[], metavar="KEY=VALUE",
                    help="loftr_hip_debug_set(KEY, VALUE) before the run (A/B switches of the library, include/loftr_hip.h); recorded in the JSON")
    ap.add_argument("--collective", default="auto", choices=["auto", "cabi", "torch"],
                    help="count all-gather transport: the library's C-ABI RCCL call, or torch.distributed (also RCCL); auto = cabi, "
                         "falling back to torch if the communicator cannot be created (recorded in the JSON)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_spawn(sys.argv[1:], args.gpus))
    if ("WORLD_SIZE" not in os.environ and os.environ.get("LOFTR_BENCH_CHILD") != "1" and os.environ.get("LOFTR_BENCH_RETRY") == "1"
            and os.environ.get("LOFTR_BENCH_FORCE_DIST") != "1"):
        sys.exit(run_with_retry(sys.argv[1:]))              # opt-in only: a faulting kernel must fail the default run
    # stdout carries exactly ONE line, the JSON: libraries that print banners to fd 1 (RCCL prints its version block at
    # communicator creation) are pointed at stderr for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rccl, transport = None, "none"
    # LOFTR_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, C-ABI communicator, collectives, timing
    # gathers) even with ONE rank -- the only way to exercise it on a 1-GPU box (tools/gpu/dist1.sh)
    multi = world > 1 or os.environ.get("LOFTR_BENCH_FORCE_DIST") == "1"
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        transport = "torch.distributed (nccl = RCCL)"
        if args.collective in ("auto", "cabi"):
            # The communicator is created and proven (one collective, checked) on a watchdog thread: in-round boxes have one
            # GPU, so ranks > 1 of this path first run under the driver -- a hang there must cost a fallback, not the run.
            import threading
            box = {}

            def _setup():
                try:
                    with torch.cuda.device(dev):
                        rc = RcclCounts(dev)
                        probe = torch.full((2,), rank + 1, dtype=torch.int32, device=dev)
                        got = rc.all_gather(probe)
                        torch.cuda.current_stream(dev).synchronize()
                        want = [r + 1 for r in range(world) for _ in range(2)]
                        if got.tolist() != want:
                            raise RuntimeError(f"probe all-gather returned {got.tolist()}, expected {want}")
                    box["rccl"] = rc
                except Exception as e:                       # noqa: BLE001
                    box["err"] = e

            th = threading.Thread(target=_setup, daemon=True)
            th.start()
            th.join(float(os.environ.get("LOFTR_BENCH_RCCL_TIMEOUT", 120)))
            if "rccl" in box:
                rccl = box["rccl"]
                transport = "C-ABI loftr_rccl_allgather_counts (RCCL over xGMI)"
            else:
                err = box.get("err", "timed out")
                if args.collective == "cabi":
                    raise RuntimeError(f"C-ABI RCCL communicator unavailable: {err}")
                print(f"[bench] rank {rank}: C-ABI RCCL communicator unavailable ({err}); using torch.distributed", file=sys.stderr)
            # every rank must take the same path: agree on it
            ok = torch.tensor([1 if rccl is not None else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and rccl is not None:
                rccl.close()
                rccl, transport = None, "torch.distributed (nccl = RCCL)"

    # MIOpen picks its fastest fp32 channels-last kernels only through the find step (47.8 vs 73 ms for
    # the 16-image backbone batch, tools/micro/backbone_variants.py); the search runs during warm-up.
    torch.backends.cudnn.benchmark = True
    lib = _lib.load()
    _lib.check(lib.loftr_hip_device_check(), "device check")
    ids = kernel_ids(lib)
    for kv in args.debug_switch:
        _lib.check(lib.loftr_hip_debug_set(kv.split("=")[0].encode(), int(kv.split("=")[1])), "--debug-switch " + kv)

    torch.manual_seed(0)                                   # backbone init
    cfg = get_cfg(thr=args.thr)
    cfg["coarse"]["temp_bug_fix"] = True                   # indoor_ds_new / notebook setting
    if args.match_type == "sinkhorn":                      # configs/loftr/indoor/loftr_ot.py + default.py:29-36
        cfg["match_coarse"].update(match_type="sinkhorn", skh_prefilter=False, sparse_spvs=True)
    model = LoFTR(cfg).eval()
    # the WHOLE state dict, loaded strict=True like a reference checkpoint (README.md:57-60): matcher weights seed 0 and the seeded backbone
    # (filters + non-trivial BatchNorm statistics) of the image-level goldens e2e_synth / e2e_batch (tests/golden/make_golden_e2e.py:
    # BACKBONE_SEED 7, bn_strength 0.3) -- pair 0 of this batch with these weights is pinned against the reference there
    sd = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(0, cfg).items()}
    for k, v in make_backbone_weights(7, model.backbone, 0.3).items():
        sd["backbone." + k] = v
    model.load_state_dict(dict(sd), strict=True)
    model = model.to(dev)
    model.coarse_matching.materialize_conf = not args.no_conf
    model.backbone_impl = args.backbone
    model.overlap_fine_branch = not args.no_overlap
    if args.coarse_mode:
        model.coarse_mode = args.coarse_mode
    if args.backbone_halves is not None:
        model.backbone_halves = bool(args.backbone_halves)
    if args.scaling == "strong":
        assert args.total_batch % world == 0, "--total-batch must be divisible by the number of ranks"
        B = args.total_batch // world
    else:
        B = args.batch
    i0, i1 = make_images(1234 + rank, B, H_IMG, W_IMG)
    img0, img1 = torch.from_numpy(i0).to(dev), torch.from_numpy(i1).to(dev)
    last = {}

    def step():
        data = {"image0": img0, "image1": img1}
        model(data)
        if multi:                                           # RCCL all-gather of the per-pair match counts
            data["match_counts_global"] = all_gather_match_counts(data["_match_counts"][1:], world * B, rccl=rccl)
        last.clear()
        last.update(M=int(data["mconf"].shape[0]), data=data)

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if multi:                                                # the gathered counts must add up to the ranks' match totals
        mt = torch.tensor([last["M"]], dtype=torch.int64, device=dev)
        dist.all_reduce(mt)
        assert int(last["data"]["match_counts_global"].sum().item()) == int(mt.item()), "count all-gather mismatch"

    # ---- extra untimed steps with every kernel instrumented, streams serial: per-kernel breakdown
    lib.loftr_hip_timing_enable((1 << len(ids)) - 1)
    for kid in ids.values():
        read_timing(lib, kid)
    NB = 3                                                   # instrumented steps for the breakdown
    bb_steps, hot_steps = [], []
    model.overlap_fine_branch = False                        # serial streams here: clean per-kernel / per-stage times
    for it in range(NB + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        with torch.no_grad():
            data = {"image0": img0, "image1": img1}
            ev[0].record()
            feats = model.run_backbone(data)
            ev[1].record()
            model.match_from_features(*feats, data)
            ev[2].record()
        torch.cuda.synchronize()
        if it == 0:                                          # the first instrumented step creates the library's event pairs (tens of ms of
            for kid in ids.values():                         # hipEventCreate on some boxes: round-4 runs showed 33 ms "backbone" means): not measured
                read_timing(lib, kid)
            continue
        bb_steps.append(ev[0].elapsed_time(ev[1]))
        hot_steps.append(ev[1].elapsed_time(ev[2]))
    backbone_ms, hot_ms = float(np.median(bb_steps)), float(np.median(hot_steps))
    M = int(data["mconf"].shape[0])
    L = (H_IMG // 8) * (W_IMG // 8)
    raw = {name: read_timing(lib, kid) for name, kid in ids.items()}
    persistent = raw.get("encoder_x_kernel", (0, 0))[1] > 0 and raw.get("proj_kv_kernel", (0, 0))[1] == 0
    work = algorithmic_work(B, L, L, M, fused=raw.get("encoder_x_kernel", (0, 0))[1] > 0,
                            fused_fine=raw.get("fine_pair_kernel", (0, 0))[1] > 0, persistent=persistent)
    kernels = []
    for name, (ms, n) in raw.items():
        if name in work and n:
            e = roofline_entry(name, ms, n, work[name][0], work[name][1], NB)
            if e:
                e["traffic"] = pmc_traffic(name)
                e["mfma_busy"] = pmc_mfma_busy(name)
                if name in POOLED_FROM:
                    e["pooled_from"] = POOLED_FROM[name]
                kernels.append(e)
    model.overlap_fine_branch = not args.no_overlap
    kernels.sort(key=lambda k: -k["ms_per_step"])

    # ---- timed region: the north_star kernels (score volume AND the encoder group) carry events -- `roofline` is measured HERE
    timed_ids = [n for n in NORTH_STAR_TIMED + ENCODER_KERNELS if n in ids]
    if args.match_type != "dual_softmax":
        timed_ids = [n for n in timed_ids if not n.startswith("score_")]
    mask = 0
    for n in timed_ids:
        mask |= 1 << ids[n]
        read_timing(lib, ids[n])
    lib.loftr_hip_timing_enable(mask)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed_local = time.perf_counter() - t0
    lib.loftr_hip_timing_enable(0)
    timing = {}
    for n in timed_ids:
        ms, cnt = read_timing(lib, ids[n])
        if cnt:
            timing[n] = (ms, cnt)
    # the timed region's match count can differ from the instrumented steps' only if the inputs did (they do not); same work table
    roof = None
    if "score_sweep_kernel<1>" in timing:
        ms, cnt = timing["score_sweep_kernel<1>"]
        roof = roofline_entry("score_sweep_kernel<1>", ms, cnt, work["score_sweep_kernel<1>"][0], work["score_sweep_kernel<1>"][1], args.steps)
        roof["bound"], roof["achieved"], roof["peak"], roof["unit"], roof["frac"] = "hbm", roof["alg_GB_s"], HBM_PEAK_GBS, "GB/s", roof["hbm_frac"]
        roof["traffic"] = pmc_traffic("score_sweep_kernel<1>")
        roof["note"] = ("north_star score-volume kernel (dual-softmax pass B: recompute the score tile on MFMA, write conf_matrix "
                        "once): algorithmic bytes = descriptors + conf_matrix (DESIGN.md §4) / in-region hipEvent launch time")
    # Encoder group (the headline `roofline`): hipEvents around every launch of these kernels INSIDE the timed region (round-5 verdict,
    # weak #3: until round 5 the headline figure came from serialised instrumented steps while the timed step ran two streams).  The
    # serial instrumented steps before the timed region give `alone` (the kernels with the GPU to themselves).
    serial = {k["kernel"]: (k["ms_per_step"] * NB, int(round(k["launches_per_step"] * NB))) for k in kernels}
    roof_alone = group_roofline([n for n in ENCODER_KERNELS if n in serial], serial, work, NB)
    roof_bb = group_roofline([n for n in BACKBONE_KERNELS if n in serial], serial, work, NB)
    roof_enc = group_roofline([n for n in ENCODER_KERNELS if n in timing], timing, work, args.steps)
    serial_step_ms = backbone_ms + hot_ms
    if roof_bb:
        roof_bb["measured"] = "hipEvents around every launch of these kernels in 3 instrumented steps of this run on one stream (kernels alone on the GPU)"
        roof_bb["share_of_serial_step"] = round(roof_bb["ms_per_step"] / serial_step_ms, 4)
    if roof_enc:
        roof_enc["measured"] = (f"hipEvents (recorded by the library on the launch stream) around every launch of these kernels in the {args.steps} steps of the "
                                "TIMED region" + (", in which the FPN fine branch runs on a second HIP stream next to the coarse transformer" if model.overlap_fine_branch else ""))
        roof_enc["kernel"] = ("coarse_persistent_kernel (the whole coarse LocalFeatureTransformer: q / k / v projections, K^T V, merge, MLP, LayerNorms; one launch) "
                              "+ fine_pair_kernel + the fine-level linear kernels: the linear-attention encoder group" if persistent else
                              "encoder_x_kernel (+ proj_kv / fine_pair / linear: the linear-attention encoder group)")
        roof_enc["coarse_transformer"] = "one persistent launch (work queue, per-pair / per-tile dependencies)" if persistent else "per-call launches"
        roof_enc["pooled_from"] = sorted({x for n in ENCODER_KERNELS for x in POOLED_FROM.get(n, [n])})
        roof_enc["share_of_step"] = round(roof_enc["ms_per_step"] / (elapsed_local / args.steps * 1e3), 4)
        if roof_alone:
            roof_enc["alone"] = {k: roof_alone[k] for k in ("achieved", "frac", "executed_TFLOP_s", "executed_frac", "ms_per_step", "launches_per_step")}
            roof_enc["alone"]["measured"] = "the same kernels in 3 instrumented steps on one stream before the timed region"
            roof_enc["alone_frac"] = roof_alone["frac"]
            roof_enc["dominant_share_of_matching_path"] = round(roof_alone["ms_per_step"] / hot_ms, 4)
    if roof is not None:
        roof["share_of_serial_step"] = round(roof["ms_per_step"] / serial_step_ms, 4)
    elapsed, per_rank_ms = elapsed_local, [round(elapsed_local / args.steps * 1e3, 3)]
    if multi:
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [round(float(x.item()) / args.steps * 1e3, 3) for x in allt]
        elapsed = max(float(x.item()) for x in allt)
        mt = torch.tensor([last["M"]], dtype=torch.int64, device=dev)
        dist.all_reduce(mt)
        m_total = int(mt.item())
    else:
        m_total = last["M"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        bb_desc = ("HIP implicit-GEMM / patch convolutions incl. the 7x7 stem" if args.backbone == "hip" else "PyTorch-ROCm / MIOpen fp32")
        out = {
            "metric": "image-pairs/sec @640x480 indoor-ds", "value": round(value, 3), "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 data everywhere; every GEMM / convolution evaluates each fp32 product as 3 fp16 MFMAs on a "
                          "(hi, lo) fp16 split with fp32 accumulation (fp32-class accuracy, csrc/gemm.h); --backbone torch = MIOpen fp32",
            "config": {"workload": f"batch={B} 640x480 synthetic grayscale pairs per GPU, indoor_ds dual-softmax "
                                   f"({'BASELINE configs[1]' if args.scaling == 'weak' else 'BASELINE configs[2]: fixed total ' + str(world * B)}), "
                                   f"full LoFTR.forward = ResNet-FPN backbone ({bb_desc}) + HIP matching path",
                       "weights": "seeded random init (no checkpoint on the box)", "thr": args.thr,
                       "thr_note": "stock thr 0.2 gives 0 matches with random weights; thr 0.0 keeps the fine stage loaded",
                       "conf_matrix_materialised": not args.no_conf, "match_type": args.match_type, "matches_per_pair": round(m_total / (world * B), 1),
                       "global_batch": world * B, "parallelism": f"dp{world} (pairs sharded; RCCL all-gather of match counts)",
                       "parity": "image-level goldens of the reference forward, both backbones (incl. a 3-pair batch and the 840 x 840 masked outdoor batch): profiles/r06_parity_margins.txt"},
            "per_rank_ms_per_step": per_rank_ms,
            "collective": {"transport": transport, "ranks_in_communicator": (rccl.ranks_seen if rccl is not None else world) if multi else 1},
            "pmc_identity": PMC_IDENTITY if pmc_table() else None,      # what ties profiles/pmc_traffic.json to this build: "source" (every kernel source unchanged) / "kernel_code" (machine code of the profiled kernels byte-identical) / None (no traffic figures)
            "stage_ms": {"backbone": round(backbone_ms, 3), "backbone_impl": args.backbone, "hot_path_hip": round(hot_ms, 3),
                         "note": "median of 3 instrumented steps (after one unmeasured instrumented step) run WITHOUT the two-stream overlap (serial sum > ms_per_step when "
                                 "the timed region overlaps the FPN fine branch with the coarse stage); `kernels` likewise"},
            "fine_branch_overlapped_in_timed_region": bool(model.overlap_fine_branch), "coarse_mode": "persistent" if persistent else "launches", "backbone_halves_on_two_streams": bool((B >= 8 if model.backbone_halves is None else model.backbone_halves) and model.overlap_fine_branch and args.backbone == "hip"),
            **({"debug_switches": args.debug_switch} if args.debug_switch else {}),
            **({"attempt": int(os.environ["LOFTR_BENCH_ATTEMPT"]), "attempt_note": "the first attempt was killed by a signal (run_with_retry)"}
               if os.environ.get("LOFTR_BENCH_ATTEMPT", "1") != "1" else {}),
            "hot_path_pairs_per_s": round(B / (hot_ms * 1e-3), 2),
            # headline roofline = the kernel group that dominates the hand-written matching path (linear-attention encoder, MFMA-bound:
            # the north_star's "MFMA utilisation on the linear-attention kernel"); the north_star's HBM figure on the score-volume
            # kernel is `roofline_score_volume`; `roofline_backbone` = the convolution kernels (largest share of the whole step)
            "roofline": roof_enc if roof_enc else roof, "roofline_score_volume": roof, "roofline_backbone": roof_bb, "kernels": kernels,
            "pmc_source": ("profiles/pmc_traffic.json (rocprofv3 --pmc passes of this build, source hash " + source_hash() + ")")
                          if pmc_table() else "no PMC passes for this build of the kernels: traffic / mfma_busy are null",
        }
        if world == 1 and not args.no_other_configs and args.match_type == "dual_softmax":
            try:
                out["other_configs"] = other_configs(lib, ids, dev, sd, args.backbone, overlap=not args.no_overlap)
            except Exception as e:                          # noqa: BLE001  (the headline line must not depend on the extras)
                out["other_configs"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, img0, img1, last.get("data"))
            out["cpu_baseline_kind"] = out["cpu_baseline"].get("kind")       # top level: "reference" (the reference's own forward) or "port" -- never to be mixed
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if multi:
        dist.barrier()
        if rccl is not None:
            rccl.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
