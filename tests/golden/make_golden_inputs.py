"""Golden vectors for the input wire format, produced by the REAL reference helpers
(src/utils/dataset.py: get_resized_wh, get_divisible_wh, pad_bottom_right) and torch's F.interpolate exactly as
src/datasets/megadepth.py:116-121 calls it.  Authoring container only:

    python tests/golden/make_golden_inputs.py        ->  tests/golden/inputs.npz
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_shim import import_reference_dataset_utils   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ds = import_reference_dataset_utils()
    out = {}
    # size arithmetic (dataset.py:61-75) on a grid of original sizes, MegaDepth test settings (840, df 8) and others
    sizes = [(1600, 1200), (1200, 1600), (1599, 1067), (640, 480), (1024, 1024), (333, 777), (3000, 2000), (845, 843)]
    rows = []
    for (w, h) in sizes:
        for resize, df in ((840, 8), (640, 8), (832, 16), (None, 8), (1200, None)):
            w1, h1 = ds.get_resized_wh(w, h, resize)
            w2, h2 = ds.get_divisible_wh(w1, h1, df)
            rows.append([w, h, -1 if resize is None else resize, -1 if df is None else df, w1, h1, w2, h2])
    out["sizes"] = np.array(rows, np.int64)
    # padding + normalisation + coarse masks (dataset.py:78-89,111-118; megadepth.py:116-121)
    rng = np.random.default_rng(5)
    for name, pad, hws, cs in (("md", 96, [(96, 64), (72, 96), (96, 96), (8, 8)], 0.125), ("odd", 80, [(77, 50), (80, 13)], 0.125),
                               ("q", 64, [(64, 40), (24, 64)], 0.25)):
        imgs, masks = [], []
        for i, (h, w) in enumerate(hws):
            im = rng.integers(0, 256, (h, w), dtype=np.uint8)
            padded, mask = ds.pad_bottom_right(im, pad, ret_mask=True)
            imgs.append((torch.from_numpy(padded).float()[None] / 255).numpy())
            masks.append(mask)
            out[f"{name}_src{i}"] = im
        m = torch.from_numpy(np.stack(masks))
        mc = F.interpolate(m[None].float(), scale_factor=cs, mode="nearest", recompute_scale_factor=False)[0].bool()
        out[f"{name}_image"] = np.stack(imgs)
        out[f"{name}_mask"] = m.numpy()
        out[f"{name}_mask_c"] = mc.numpy()
        out[f"{name}_meta"] = np.array([pad, len(hws), int(round(1 / cs))], np.int64)
    np.savez_compressed(os.path.join(HERE, "inputs.npz"), **out)
    print("wrote inputs.npz", {k: v.shape for k, v in out.items() if k.endswith(("image", "mask_c"))})


if __name__ == "__main__":
    main()
