"""Input wire format of the matching path: the reference's loader-side preparation
(src/utils/dataset.py:61-89,92-150, src/datasets/megadepth.py:116-121) from the resized uint8 image on.

    w1, h1 = get_divisible_wh(*get_resized_wh(w, h, 840), 8)          # same size arithmetic as the reference
    img_u8 = cv2.resize(cv2.imread(path, 0), (w1, h1))                  # caller: OpenCV decode + resize
    batch = pack_pairs(list0_u8, list1_u8, pad_to=840, orig_sizes0=..., orig_sizes1=...)
    matcher(batch)                                                      # image0/1, mask0/1 (coarse), scale0/1

The host uploads 1 byte per pixel; padding, `/ 255`, the masks and their 1/8 versions are produced on the
device by `loftr_pack_gray_u8` (csrc/input.hip).  No CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import _ptr, _stream, check, _on_device


def get_resized_wh(w, h, resize=None):
    """dataset.py:61-67."""
    if resize is not None:
        scale = resize / max(h, w)
        return int(round(w * scale)), int(round(h * scale))
    return w, h


def get_divisible_wh(w, h, df=None):
    """dataset.py:70-75."""
    if df is not None:
        return int(w // df * df), int(h // df * df)
    return w, h


@_on_device
def resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize) (default INTER_LINEAR) for a uint8 grayscale DEVICE tensor [h, w] -> [h_new, w_new];
    dsize = (w_new, h_new).  Parity unpinned (see include/loftr_hip.h): OpenCV's algorithm restated, not verified against
    the library."""
    if not (torch.is_tensor(img) and img.is_cuda and img.dtype == torch.uint8 and img.dim() == 2):
        raise _lib.LoftrHipError("resize_linear_u8: expected a uint8 GPU tensor [h, w]")
    img = img.contiguous()
    dw, dh = int(dsize[0]), int(dsize[1])
    out = torch.empty(dh, dw, dtype=torch.uint8, device=img.device)
    check(_lib.load().loftr_resize_linear_u8(_ptr(img), img.shape[0], img.shape[1], img.shape[1], _ptr(out), dh, dw, dw, _stream()),
          "loftr_resize_linear_u8")
    return out


class _PinnedRing:
    """A few persistent page-locked staging buffers reused round-robin.  A slot is handed out again only after the
    event recorded behind its last upload has completed, so the asynchronous H2D copy never races the next fill;
    allocating / pinning host memory per batch costs milliseconds, this costs a memcpy."""

    def __init__(self, slots=4):
        self.slots = [dict(buf=None, ev=None) for _ in range(slots)]
        self.next = 0

    def acquire(self, nbytes):
        slot = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        if slot["ev"] is not None:
            slot["ev"].synchronize()
        if slot["buf"] is None or slot["buf"].numel() < nbytes:
            slot["buf"] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
        return slot

    def release(self, slot):
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record(torch.cuda.current_stream())


_STAGING = _PinnedRing()


def pack_gray(images, pad_hw=None, coarse_div=8, device="cuda", want_mask=True):
    """images: list of uint8 arrays / tensors [h_i, w_i] (already resized) -> (image f32 [N,1,PH,PW] on the device,
    mask bool [N,PH,PW] or None, mask_c bool [N,PH//d,PW//d] or None).  pad_hw None: all images share one size and
    nothing is padded (ScanNet: dataset.py:146-150, no masks)."""
    if len(images) == 0:
        raise ValueError("pack_gray: empty batch")
    arrs = [np.ascontiguousarray(im.cpu().numpy() if torch.is_tensor(im) else im) for im in images]
    for a in arrs:
        if a.dtype != np.uint8 or a.ndim != 2:
            raise _lib.LoftrHipError("pack_gray: expected uint8 [h, w] grayscale images")
    hmax, wmax = max(a.shape[0] for a in arrs), max(a.shape[1] for a in arrs)
    if pad_hw is None:
        if any(a.shape != arrs[0].shape for a in arrs):
            raise _lib.LoftrHipError("pack_gray: images of different sizes need pad_hw (the reference pads to a square, dataset.py:111-113)")
        PH, PW, masks = hmax, wmax, False
    else:
        PH, PW = (pad_hw, pad_hw) if isinstance(pad_hw, int) else pad_hw
        masks = want_mask
        if PH < hmax or PW < wmax:
            raise AssertionError(f"{(PH, PW)} < {(hmax, wmax)}")           # pad_bottom_right's assert (dataset.py:79)
    N = len(arrs)
    if not torch.cuda.is_available():
        raise _lib.LoftrHipError("pack_gray needs a GPU (no CPU fallback)")
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    with torch.cuda.device(dev):                     # copy, event and launch all on the target GPU's current stream
        return _pack_gray_on(dev, arrs, N, hmax, wmax, PH, PW, masks, coarse_div)


def _pack_gray_on(dev, arrs, N, hmax, wmax, PH, PW, masks, coarse_div):
    # staging: a slot of the persistent pinned ring (one row pitch for the batch, the (h, w) table behind the pixels),
    # filled with plain memcpys and uploaded by ONE asynchronous H2D copy of ~1 B / pixel
    pitch = (wmax + 3) // 4 * 4
    npix = N * hmax * pitch
    slot = _STAGING.acquire(npix + 8 * N)
    host = slot["buf"].numpy()
    pix = host[:npix].reshape(N, hmax, pitch)
    hw_host = host[npix:npix + 8 * N].view(np.int32).reshape(N, 2)
    for n, a in enumerate(arrs):
        pix[n, :a.shape[0], :a.shape[1]] = a
        hw_host[n] = a.shape
    both = slot["buf"][:npix + 8 * N].to(dev, non_blocking=True)
    _STAGING.release(slot)
    src = both[:npix]
    hw_d = both[npix:].view(torch.int32)
    image = torch.empty(N, 1, PH, PW, dtype=torch.float32, device=dev)
    mask = torch.empty(N, PH, PW, dtype=torch.uint8, device=dev) if masks else None
    mask_c = torch.empty(N, PH // coarse_div, PW // coarse_div, dtype=torch.uint8, device=dev) if masks and coarse_div else None
    check(_lib.load().loftr_pack_gray_u8(_ptr(src), hmax * pitch, pitch, _ptr(hw_d), N, PH, PW, _ptr(image), _ptr(mask), _ptr(mask_c),
                                         int(coarse_div or 0), _stream()), "loftr_pack_gray_u8")
    return image, (mask.bool() if mask is not None else None), (mask_c.bool() if mask_c is not None else None)


def pack_pairs(images0, images1, pad_to=None, orig_sizes0=None, orig_sizes1=None, coarse_div=8, device="cuda"):
    """Batch dict for LoFTR.forward from two lists of resized uint8 images: image0/1 (+ mask0/1 = COARSE masks as in
    megadepth.py:116-122 and scale0/1 = [w / w_new, h / h_new], dataset.py:109, when padding is requested)."""
    im0, _, mc0 = pack_gray(images0, pad_to, coarse_div, device)
    im1, _, mc1 = pack_gray(images1, pad_to, coarse_div, device)
    batch = {"image0": im0, "image1": im1}
    if pad_to is not None:
        batch.update({"mask0": mc0, "mask1": mc1})
    for key, imgs, orig in (("scale0", images0, orig_sizes0), ("scale1", images1, orig_sizes1)):
        if orig is not None:                                   # orig: [(w, h)] of the images before resizing
            sc = [[w / im.shape[1], h / im.shape[0]] for (w, h), im in zip(orig, imgs)]
            batch[key] = torch.tensor(sc, dtype=torch.float32, device=device)
    return batch
