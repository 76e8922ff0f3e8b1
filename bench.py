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

Multi-GPU: pairs are independent -> each rank owns its own B pairs (weak scaling); the only
collective is the count all-gather.

The JSON line also carries
  roofline      -- the dominant hot-path kernel against its bound: algorithmic bytes/flops per
                   launch / hipEvent-measured average launch duration (events
                   recorded by the library on the launch stream inside the timed region);
  kernels       -- the same for every instrumented kernel (measured in one extra untimed step);
  cpu_baseline  -- the CPU port (torch-CPU backbone + numpy oracle of the hot path) timed on this
                   box's host cores on ONE pair of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from loftr_amd import LoFTR, get_cfg, _lib          # noqa: E402
from loftr_amd.distributed import all_gather_match_counts   # noqa: E402
from loftr_amd.synth import make_images, make_weights   # noqa: E402

H_IMG, W_IMG = 480, 640
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak (for context only)
SPLIT_FACTOR = 3               # every fp32 product = 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi), csrc/gemm.h
GEMM_KERNELS = ("proj_kernel", "proj_kv_kernel", "linear_kernel", "linear_ln_kernel", "score_stats_kernel", "score_conf_kernel",
                "conv_kernel", "conv3x3_kernel", "conv3x3_wide_kernel")

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


def algorithmic_work(B, L, S, M, C_=256, Cf=128, WW=25):
    """Algorithmic (flops, bytes) per STEP of every instrumented kernel for a batch of B pairs with
    M matches in total.  fp32 everywhere: 4 bytes per element."""
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
    for k in coarse:
        w[k] = (coarse[k][0] + fine[k][0], coarse[k][1] + fine[k][1])
    # fine preprocess linears ride on linear_kernel: down_proj, ctx, window merge (x2 sides)
    fp_f = 2 * (2 * M * C_ * Cf + 2 * M * Cf * Cf + 2 * M * WW * Cf * Cf)
    fp_b = 2 * 4 * (M * C_ + Cf * C_ + M * Cf + M * Cf + Cf * Cf + M * Cf + 2 * M * WW * Cf + Cf * Cf)
    w["linear_kernel"] = (w["linear_kernel"][0] + fp_f, w["linear_kernel"][1] + fp_b)
    nl = n_self + n_cross
    w["proj_kv_kernel"] = (nl * (2 * rows_c * C_ * 2 * C_ + 2 * rows_c * C_ * 32), nl * 4 * (rows_c * C_ + 2 * C_ * C_))
    w["attn_small_kernel"] = (2 * 2 * (2 * 2 * M * WW * Cf * 16), 2 * 4 * 4 * 2 * M * WW * Cf)
    w["score_stats_kernel"] = (2 * B * L * S * C_, 4 * B * (L + S) * C_)
    w["score_conf_kernel"] = (2 * B * L * S * C_, 4 * B * ((L + S) * C_ + L * S))
    w["gather_windows_kernel"] = (0, 2 * 4 * 2 * M * WW * Cf)
    # backbone convolutions on the same GEMM core (conv.hip): ResNetFPN_8_2 over 2B images of 480x640
    # 3x3 stride-1 layers run the patch-in-LDS kernels (the 224-column one when ceil32(Cout) == 224)
    acc = {"conv_kernel": [0, 0], "conv3x3_kernel": [0, 0], "conv3x3_wide_kernel": [0, 0]}
    for (cin, cout, k, stride, hin, win) in backbone_convs(H_IMG, W_IMG):
        ho, wo = hin // stride, win // stride
        patch = k == 3 and stride == 1
        a = acc[("conv3x3_wide_kernel" if (cout + 31) // 32 == 7 else "conv3x3_kernel") if patch else "conv_kernel"]
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


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 --pmc pass (profiles/pmc_traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(kernel, {}).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def cpu_baseline(model, img0, img1):
    """CPU port of the same forward on ONE pair: torch-CPU backbone (the reference's backbone is the
    same PyTorch module) + oracle/loftr_oracle.py (numpy restatement of the matching path)."""
    from oracle import loftr_oracle as O
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    cpu_model = model.backbone.to("cpu").float()
    w = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if not k.startswith("backbone.")}
    x = torch.cat([img0[:1].cpu(), img1[:1].cpu()], 0)
    t0 = time.perf_counter()
    with torch.no_grad():
        fc, ff = cpu_model(x)
    t1 = time.perf_counter()
    fc, ff = fc.numpy(), ff.numpy()
    out = O.loftr_hot_path(fc[:1], fc[1:], ff[:1], ff[1:], w, model.config, (H_IMG, W_IMG), (H_IMG, W_IMG))
    t2 = time.perf_counter()
    model.backbone.to(img0.device)
    total = t2 - t0
    return {"value": round(1.0 / total, 4), "unit": "image-pairs/s", "cores": cores, "kind": "port",
            "sample": f"1 of the batch's 640x480 pairs, 1 run: torch-CPU backbone {t1 - t0:.2f}s + numpy oracle "
                      f"matching path {t2 - t1:.2f}s (M={len(out['mconf'])})",
            "backbone_s": round(t1 - t0, 3), "hot_path_s": round(t2 - t1, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU per step")
    ap.add_argument("--thr", type=float, default=0.0)
    ap.add_argument("--no-conf", action="store_true", help="elide data['conf_matrix'] (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-kernel", default="auto")
    ap.add_argument("--no-overlap", action="store_true", help="run the FPN fine branch on the main stream (no second HIP stream)")
    ap.add_argument("--match-type", default="dual_softmax", choices=["dual_softmax", "sinkhorn"],
                    help="sinkhorn = BASELINE configs[4] (indoor_ot); not the headline")
    ap.add_argument("--backbone", default="hip", choices=["hip", "torch"],
                    help="hip: implicit-GEMM convolutions of this library (default); torch: MIOpen fp32")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    # MIOpen picks its fastest fp32 channels-last kernels only through the find step (47.8 vs 73 ms for
    # the 16-image backbone batch, tools/micro/backbone_variants.py); the search runs during warm-up.
    torch.backends.cudnn.benchmark = True
    lib = _lib.load()
    _lib.check(lib.loftr_hip_device_check(), "device check")
    ids = kernel_ids(lib)

    torch.manual_seed(0)                                   # backbone init
    cfg = get_cfg(thr=args.thr)
    cfg["coarse"]["temp_bug_fix"] = True                   # indoor_ds_new / notebook setting
    if args.match_type == "sinkhorn":                      # configs/loftr/indoor/loftr_ot.py + default.py:29-36
        cfg["match_coarse"].update(match_type="sinkhorn", skh_prefilter=False, sparse_spvs=True)
    model = LoFTR(cfg).eval()
    sd = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(0, cfg).items()}
    model.load_state_dict(sd, strict=False)
    model = model.to(dev)
    model.coarse_matching.materialize_conf = not args.no_conf
    model.backbone_impl = args.backbone
    model.overlap_fine_branch = not args.no_overlap
    B = args.batch
    i0, i1 = make_images(1234 + rank, B, H_IMG, W_IMG)
    img0, img1 = torch.from_numpy(i0).to(dev), torch.from_numpy(i1).to(dev)
    last = {}

    def step():
        data = {"image0": img0, "image1": img1}
        model(data)
        if world > 1:                                       # RCCL all-gather of the per-pair match counts
            data["match_counts_global"] = all_gather_match_counts(data["_match_counts"][1:], world * B)
        last.clear()
        last.update(M=int(data["mconf"].shape[0]), data=data)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()

    # ---- one extra untimed step with every kernel instrumented: breakdown + pick the dominant one
    lib.loftr_hip_timing_enable((1 << len(ids)) - 1)
    for kid in ids.values():
        read_timing(lib, kid)
    NB = 3                                                   # instrumented steps for the breakdown
    backbone_ms = hot_ms = 0.0
    model.overlap_fine_branch = False                        # serial streams here: clean per-kernel / per-stage times
    for _ in range(NB):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        with torch.no_grad():
            data = {"image0": img0, "image1": img1}
            ev[0].record()
            feats = model.run_backbone(data)
            ev[1].record()
            model.match_from_features(*feats, data)
            ev[2].record()
        torch.cuda.synchronize()
        backbone_ms += ev[0].elapsed_time(ev[1]) / NB
        hot_ms += ev[1].elapsed_time(ev[2]) / NB
    M = int(data["mconf"].shape[0])
    L = (H_IMG // 8) * (W_IMG // 8)
    work = algorithmic_work(B, L, L, M)
    kernels = []
    for name, kid in ids.items():
        ms, n = read_timing(lib, kid)
        if name in work and n:
            kernels.append(roofline_entry(name, ms, n, work[name][0], work[name][1], NB))
    model.overlap_fine_branch = not args.no_overlap
    kernels = [k for k in kernels if k]
    kernels.sort(key=lambda k: -k["ms_per_step"])
    dom = args.roofline_kernel if args.roofline_kernel != "auto" else (kernels[0]["kernel"] if kernels else None)

    # ---- timed region: only the dominant kernel carries events
    lib.loftr_hip_timing_enable(1 << ids[dom] if dom else 0)
    if dom:
        read_timing(lib, ids[dom])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.loftr_hip_timing_enable(0)
    roof = None
    if dom:
        ms, n = read_timing(lib, ids[dom])
        roof = roofline_entry(dom, ms, n, work[dom][0], work[dom][1], args.steps)
        if roof:
            roof["traffic"] = pmc_traffic(dom)
            if not args.no_overlap and args.backbone == "hip":
                solo = next((k for k in kernels if k["kernel"] == dom), None)
                roof["note"] = ("timed region: the fine-branch launches of this kernel run on the side stream CONCURRENTLY with the "
                                "coarse stage, so their durations (and this average) include time-slicing with other kernels; "
                                "running alone (instrumented steps, `kernels`): avg %.1f us, frac %.3f"
                                % ((solo["avg_launch_us"], solo["frac"]) if solo else (float("nan"), float("nan"))))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mt = torch.tensor([last["M"]], dtype=torch.int64, device=dev)
        dist.all_reduce(mt)
        m_total = int(mt.item())
    else:
        m_total = last["M"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        out = {
            "metric": "image-pairs/sec @640x480 indoor-ds", "value": round(value, 3), "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 data everywhere; every GEMM / convolution evaluates each fp32 product as 3 fp16 MFMAs on a "
                          "(hi, lo) fp16 split with fp32 accumulation (fp32-class accuracy, csrc/gemm.h); --backbone torch = MIOpen fp32",
            "config": {"workload": f"batch={B} 640x480 synthetic grayscale pairs per GPU, indoor_ds dual-softmax "
                                   f"(BASELINE configs[1]), full LoFTR.forward = ResNet-FPN backbone ({'HIP implicit-GEMM / patch convolutions incl. the 7x7 stem' if args.backbone == 'hip' else 'PyTorch-ROCm / MIOpen fp32'}) + HIP matching path",
                       "weights": "seeded random init (no checkpoint on the box)", "thr": args.thr,
                       "thr_note": "stock thr 0.2 gives 0 matches with random weights; thr 0.0 keeps the fine stage loaded",
                       "conf_matrix_materialised": not args.no_conf, "match_type": args.match_type, "matches_per_pair": round(m_total / (world * B), 1),
                       "global_batch": world * B, "parallelism": f"dp{world} (pairs sharded; RCCL all-gather of match counts)"},
            "stage_ms": {"backbone": round(backbone_ms, 3), "backbone_impl": args.backbone, "hot_path_hip": round(hot_ms, 3),
                         "note": "mean of 3 instrumented steps run WITHOUT the two-stream overlap (serial sum > ms_per_step when "
                                 "the timed region overlaps the FPN fine branch with the coarse stage); `kernels` likewise"},
            "fine_branch_overlapped_in_timed_region": not args.no_overlap,
            "hot_path_pairs_per_s": round(B / (hot_ms * 1e-3), 2),
            "roofline": roof, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, img0, img1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
