#!/usr/bin/env python
"""Does the ResNet-FPN's tail / round quantisation go away when the 16-image batch runs as two 8-image halves on two HIP streams?

    python tools/micro/backbone_halves.py [reps]

One launch per layer over 16 images (the product's schedule) against the two image sets (image0 batch, image1 batch: exactly the halves
LoFTR.forward splits the backbone output into, loftr.py:44-49) on two streams, whose launches fill each other's last partly filled
round of workgroups.  Results are bit-identical by construction (same kernels per image)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loftr_amd import LoFTR, get_cfg                                  # noqa: E402
from loftr_amd.synth import make_images, make_weights, make_backbone_weights   # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = get_cfg(thr=0.0)
model = LoFTR(cfg).eval()
sd = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(0, cfg).items()}
for k, v in make_backbone_weights(7, model.backbone, 0.3).items():
    sd["backbone." + k] = v
model.load_state_dict(sd, strict=True)
model = model.cuda()
bb = model.backbone
i0, i1 = make_images(1234, 8, 480, 640)
x = torch.cat([torch.from_numpy(i0), torch.from_numpy(i1)]).cuda().contiguous(memory_format=torch.channels_last)
s = [torch.cuda.Stream() for _ in range(4)]
main = torch.cuda.current_stream()


def whole():
    return bb.forward_hip(x)


def split(n):
    outs = []
    step = x.shape[0] // n
    for k in range(n):
        s[k].wait_stream(main)
        with torch.cuda.stream(s[k]):
            outs.append(bb.forward_hip(x[k * step:(k + 1) * step]))
    for k in range(n):
        main.wait_stream(s[k])
    return outs


def timed(fn):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    a = whole()
    b = split(2)
    torch.cuda.synchronize()
    same = all(torch.equal(a[j], torch.cat([b[0][j], b[1][j]])) for j in range(2))
print(f"two halves vs one batch: {'bit-identical' if same else 'DIFFERENT'}")
for name, fn in (("one batch of 16, one stream", whole), ("2 x 8 on two streams", lambda: split(2)), ("4 x 4 on four streams", lambda: split(4)),
                 ("one batch of 16, one stream", whole)):
    print(f"{name:32s} {timed(fn):7.3f} ms per backbone pass")
