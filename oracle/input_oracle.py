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


# ---- cv2.resize(image, (w_new, h_new)) for uint8 grayscale, default INTER_LINEAR (dataset.py:108,146) -------------
# PARITY UNPINNED: OpenCV is not installed in this image, so neither the library nor golden vectors made with it are
# available.  The function below restates the published algorithm of OpenCV 4.x `resize.cpp` for 8UC1 /
# INTER_LINEAR (fixed point, INTER_RESIZE_COEF_BITS = 11): it pins the device kernel `loftr_resize_linear_u8` to THIS
# restatement only.  When cv2 becomes available: compare against it and move the resize into the pinned wire format.
def resize_linear_u8(img, dsize):
    """img uint8 [h, w] -> uint8 [h_new, w_new], dsize = (w_new, h_new) as in cv2.resize."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    sh, sw = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])

    def axis(n_dst, n_src, zero_at_border):
        scale = 1.0 / (float(n_dst) / float(n_src))                       # scale_x = 1. / inv_scale_x (doubles)
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if zero_at_border:                                                # columns: coefficient forced to 0 outside
            lo, hi = s < 0, s >= n_src - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, n_src - 1, s))
        c0 = np.rint((np.float32(1.0) - f) * np.float32(2048)).astype(np.int64)      # saturate_cast<short>(cvRound)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, np.clip(c0, -32768, 32767), np.clip(c1, -32768, 32767)

    sx, a0, a1 = axis(dw, sw, True)
    sy, b0, b1 = axis(dh, sh, False)
    x1 = np.minimum(sx + 1, sw - 1)
    y0 = np.clip(sy, 0, sh - 1)                                            # rows: indices clamped, coefficients kept
    y1 = np.clip(sy + 1, 0, sh - 1)
    src = img.astype(np.int64)
    H0 = src[y0][:, sx] * a0[None, :] + src[y0][:, x1] * a1[None, :]       # horizontal pass, scale 2^11
    H1 = src[y1][:, sx] * a0[None, :] + src[y1][:, x1] * a1[None, :]
    out = (((b0[:, None] * (H0 >> 4)) >> 16) + ((b1[:, None] * (H1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)
