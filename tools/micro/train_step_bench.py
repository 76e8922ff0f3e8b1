#!/usr/bin/env python
"""Forward + backward of the matcher (everything after the backbone) in .train() mode with LoFTR.full_grads at 640 x 480: the backbone
features are leaves, the loss is a random linear functional of conf_matrix and expec_f (no supervision data needed for a timing).
    python tools/micro/train_step_bench.py [N=2] [reps=3] [images]
"images": the step starts from the images -- the backbone in train mode with its convolutions on the HIP autograd node
(LOFTR_TRAIN_CONV=0: on PyTorch / MIOpen), BatchNorm with batch statistics -- so every parameter of the model gets a gradient."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import LoFTR, get_cfg  # noqa: E402

def main(N=2, reps=3, images=False):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = get_cfg(thr=0.0); cfg["coarse"]["temp_bug_fix"] = True
    m = LoFTR(cfg).to(dev).train()
    m.full_grads = True
    fc = torch.randn(2 * N, 256, 60, 80, device=dev); ff = torch.randn(2 * N, 128, 240, 320, device=dev)
    img = torch.rand(2 * N, 1, 480, 640, device=dev)
    def step():
        if images:
            with torch.enable_grad():
                a, b = m.backbone(img)
        else:
            a, b = fc.clone().requires_grad_(True), ff.clone().requires_grad_(True)
        data = {"bs": N, "hw0_i": (480, 640), "hw1_i": (480, 640)}
        m.coarse_matching.train(False)                      # eval-style selection (no ground truth to pad with), conf_matrix with its graph
        with torch.enable_grad():
            m.match_from_features(a[:N], a[N:], b[:N], b[N:], data)
            loss = (data["conf_matrix"] * 1e-3).sum() + data["expec_f"].sum()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, data["mconf"].shape[0]
    step()
    tf = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tb, M = step()
        torch.cuda.synchronize(); tf.append((time.perf_counter() - t0 - tb, tb))
    f, b = sorted(tf)[len(tf) // 2]
    from loftr_amd import backbone as BB
    what = ("backbone (convolutions: " + ("HIP nodes" if BB.TRAIN_CONV_HIP else "PyTorch / MIOpen") + ") + matcher") if images else "matcher"
    print(f"N={N} pairs 640x480, M={M}: {what} forward (autograd nodes, unfused transformer schedule) {f*1e3:.1f} ms, backward {b*1e3:.1f} ms "
          f"({sum(p.numel() for p in m.parameters() if p.grad is not None) / 1e6:.1f} M parameters with gradients)")

if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 2, int(a[1]) if len(a) > 1 else 3, len(a) > 2 and a[2] == "images")
