"""Input wire format on the GPU: loftr_pack_gray_u8 through the C-ABI, bit-exact against
the reference-generated goldens and the numpy oracle; full-size properties; the batch feeds LoFTR.forward."""
import numpy as np
import pytest
import torch

from oracle import input_oracle as io_
from test_input_oracle import GOLD, CASES, case_images

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_pack_gray_bit_exact_vs_reference_golden(name):
    from loftr_amd import inputs
    npz = np.load(GOLD)
    imgs, pad, div = case_images(npz, name)
    image, mask, mask_c = inputs.pack_gray(imgs, pad, coarse_div=div)
    assert image.is_cuda and image.dtype == torch.float32 and mask.dtype == torch.bool
    assert np.array_equal(image.cpu().numpy(), npz[f"{name}_image"])
    assert np.array_equal(mask.cpu().numpy(), npz[f"{name}_mask"])
    assert np.array_equal(mask_c.cpu().numpy(), npz[f"{name}_mask_c"])


def test_pack_gray_full_size_properties_and_edges():
    from loftr_amd import inputs
    from loftr_amd._lib import LoftrHipError
    rng = np.random.default_rng(3)
    hws = [(840, 560), (632, 840), (840, 840), (8, 8), (837, 843 - 8)]            # MegaDepth-style, incl. odd widths
    imgs = [rng.integers(0, 256, hw, dtype=np.uint8) for hw in hws]
    image, mask, mask_c = inputs.pack_gray(imgs, 840)
    ref_i, ref_m, ref_c = io_.pack_gray(imgs, (840, 840))
    assert np.array_equal(image.cpu().numpy(), ref_i) and np.array_equal(mask.cpu().numpy(), ref_m)
    assert np.array_equal(mask_c.cpu().numpy(), ref_c) and mask_c.shape == (5, 105, 105)
    # size-independent properties: exact round trip of the bytes, zero padding, mask areas
    back = torch.round(image[:, 0] * 255).to(torch.uint8).cpu().numpy()
    for n, (im, (h, w)) in enumerate(zip(imgs, hws)):
        assert np.array_equal(back[n, :h, :w], im) and back[n, h:].sum() == 0 and back[n, :, w:].sum() == 0
        assert int(mask[n].sum()) == h * w and int(mask_c[n].sum()) == -(-h // 8) * -(-w // 8)
    # no padding requested: one common size, no masks (ScanNet loader)
    same = [rng.integers(0, 256, (480, 640), dtype=np.uint8) for _ in range(3)]
    image, mask, mask_c = inputs.pack_gray(same)
    assert image.shape == (3, 1, 480, 640) and mask is None and mask_c is None
    assert np.array_equal(image.cpu().numpy()[:, 0], np.stack(same).astype(np.float32) / np.float32(255))
    with pytest.raises(LoftrHipError):
        inputs.pack_gray([same[0], imgs[3]])                                   # ragged sizes need pad_hw
    with pytest.raises(AssertionError):
        inputs.pack_gray(imgs, 800)                                            # pad smaller than an image (dataset.py:79)
    with pytest.raises(LoftrHipError):
        inputs.pack_gray([same[0].astype(np.float32)])


def test_packed_batch_feeds_the_matcher():
    """pack_pairs -> LoFTR.forward: masks at coarse resolution and scales in the keys the reference's loader uses."""
    import copy
    from loftr_amd import LoFTR, default_cfg, inputs
    rng = np.random.default_rng(4)
    i0 = [rng.integers(0, 256, (240, 160), dtype=np.uint8), rng.integers(0, 256, (200, 240), dtype=np.uint8)]
    i1 = [rng.integers(0, 256, (240, 240), dtype=np.uint8), rng.integers(0, 256, (160, 240), dtype=np.uint8)]
    batch = inputs.pack_pairs(i0, i1, pad_to=240, orig_sizes0=[(640, 960), (960, 800)], orig_sizes1=[(480, 480), (1200, 800)])
    assert batch["image0"].shape == (2, 1, 240, 240) and batch["mask0"].shape == (2, 30, 30) and batch["mask0"].dtype == torch.bool
    assert torch.allclose(batch["scale0"].cpu(), torch.tensor([[4.0, 4.0], [4.0, 4.0]]))
    assert torch.allclose(batch["scale1"].cpu(), torch.tensor([[2.0, 2.0], [5.0, 5.0]]))
    cfg = copy.deepcopy(default_cfg)
    cfg["match_coarse"]["thr"] = 0.0
    torch.manual_seed(0)
    m = LoFTR(config=cfg).eval().to("cuda:0")
    m(batch)
    M = batch["mkpts0_f"].shape[0]
    assert M > 0
    # matches only inside the valid rectangles (mask0/mask1), in original-image pixels (x scale)
    b = batch["m_bids"].cpu().numpy()
    k0 = batch["mkpts0_f"].cpu().numpy()
    assert (k0[b == 0][:, 0] < 160 * 4.0).all() and (k0[b == 1][:, 1] < 200 * 4.0).all()


def test_resize_linear_u8_matches_restatement():
    """Device bilinear resize vs the numpy restatement of OpenCV's algorithm, bit-exact (PARITY UNPINNED against cv2)."""
    from loftr_amd import inputs
    rng = np.random.default_rng(8)
    for (h, w), dsize in (((480, 640), (640, 480)), ((1200, 1600), (840, 632)), ((1067, 1599), (840, 560)), ((480, 640), (832, 624)),
                          ((33, 57), (200, 101)), ((64, 64), (1, 1)), ((5, 7), (7, 5))):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        got = inputs.resize_linear_u8(torch.from_numpy(img).cuda(), dsize).cpu().numpy()
        assert got.shape == (dsize[1], dsize[0]) and np.array_equal(got, io_.resize_linear_u8(img, dsize)), ((h, w), dsize)
