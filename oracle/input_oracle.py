"""CPU restatement of the reference's input wire format: what the data loaders hand to
`LoFTR.forward` once a grayscale image has been decoded and resized.

TEST INFRASTRUCTURE ONLY -- imported by tests/ (and nothing else).  Product: csrc/input.hip + loftr_amd/inputs.py.

Pinned against the real `src/utils/dataset.py` helpers and torch's F.interpolate by tests/golden/inputs.npz
(generator: tests/golden/make_golden_inputs.py).  NOT restated: `cv2.imread` / `cv2.imdecode` / `cv2.resize`
(dataset.py:40-56,108,146) -- OpenCV (opencv-python 4.4.0.46) is absent from this image and its fixed-point
bilinear resize cannot be pinned without it; the wire format below starts from the resized uint8 image.
"""
import numpy as np


def get_resized_wh(w, h, resize=None):
    """dataset.py:61-67: scale the longer edge to `resize`."""
    if resize is not None:
        scale = resize / max(h, w)
        return int(round(w * scale)), int(round(h * scale))
    return w, h


def get_divisible_wh(w, h, df=None):
    """dataset.py:70-75: round both down to a multiple of df."""
    if df is not None:
        return int(w // df * df), int(h // df * df)
    return w, h


def pack_gray(images, pad_hw, coarse_scale=0.125):
    """images: list of uint8 [h_i, w_i] -> (image float32 [N,1,PH,PW], mask bool [N,PH,PW], mask_c bool [N,PH*s,PW*s]).

    dataset.py:78-89 `pad_bottom_right` (zero pad at the bottom / right, mask True on the image), :117 and :149
    `torch.from_numpy(image).float()[None] / 255`, megadepth.py:116-121 `F.interpolate(mask.float(),
    scale_factor=coarse_scale, mode='nearest', recompute_scale_factor=False).bool()` = mask[floor(y / s), floor(x / s)]
    on an output of floor(P * s) cells."""
    PH, PW = pad_hw
    N = len(images)
    img = np.zeros((N, 1, PH, PW), np.float32)
    mask = np.zeros((N, PH, PW), bool)
    for n, im in enumerate(images):
        h, w = im.shape
        img[n, 0, :h, :w] = im.astype(np.float32) / np.float32(255)
        mask[n, :h, :w] = True
    ch, cw = int(np.floor(PH * coarse_scale)), int(np.floor(PW * coarse_scale))
    inv = 1.0 / coarse_scale
    ys = np.minimum(np.floor(np.arange(ch) * np.float32(inv)).astype(np.int64), PH - 1)
    xs = np.minimum(np.floor(np.arange(cw) * np.float32(inv)).astype(np.int64), PW - 1)
    mask_c = mask[:, ys][:, :, xs]
    return img, mask, mask_c


def scale_of(w, h, w_new, h_new):
    """dataset.py:109: scale = [w / w_new, h / h_new] (float32)."""
    return np.array([w / w_new, h / h_new], np.float32)
