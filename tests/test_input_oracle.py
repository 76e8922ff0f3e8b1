"""Input wire format: oracle and host-side size arithmetic against goldens produced by the
reference's own src/utils/dataset.py helpers and torch's F.interpolate (tests/golden/make_golden_inputs.py)."""
import os

import numpy as np
import pytest

from oracle import input_oracle as io_

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs.npz")
CASES = ("md", "odd", "q")


def case_images(npz, name):
    pad, n, div = (int(v) for v in npz[f"{name}_meta"])
    return [npz[f"{name}_src{i}"] for i in range(n)], pad, div


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_size_arithmetic_matches_reference(impl):
    if impl == "oracle":
        m = io_
    else:
        from loftr_amd import inputs as m
    rows = np.load(GOLD)["sizes"]
    assert len(rows) == 40
    for w, h, resize, df, w1, h1, w2, h2 in rows:
        r = None if resize < 0 else int(resize)
        d = None if df < 0 else int(df)
        assert m.get_resized_wh(int(w), int(h), r) == (w1, h1)
        assert m.get_divisible_wh(int(w1), int(h1), d) == (w2, h2)


@pytest.mark.parametrize("name", CASES)
def test_oracle_pack_matches_reference(name):
    npz = np.load(GOLD)
    imgs, pad, div = case_images(npz, name)
    image, mask, mask_c = io_.pack_gray(imgs, (pad, pad), 1.0 / div)
    assert image.dtype == np.float32 and np.array_equal(image, npz[f"{name}_image"])          # bit-exact
    assert np.array_equal(mask, npz[f"{name}_mask"]) and np.array_equal(mask_c, npz[f"{name}_mask_c"])


def test_resize_restatement_properties():
    """The OpenCV-bilinear restatement (PARITY UNPINNED: no cv2 here) at least satisfies what the algorithm implies:
    identity at equal size, constants preserved, exact 2x2 box average at half size (coefficients 1024 / 1024)."""
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(io_.resize_linear_u8(a, (53, 37)), a)
    c = np.full((40, 60), 137, np.uint8)
    assert (io_.resize_linear_u8(c, (33, 21)) == 137).all() and (io_.resize_linear_u8(c, (121, 77)) == 137).all()
    b = rng.integers(0, 256, (40, 60), dtype=np.uint8).astype(np.int64)
    box = (b[0::2, 0::2] + b[0::2, 1::2] + b[1::2, 0::2] + b[1::2, 1::2] + 2) // 4
    assert np.array_equal(io_.resize_linear_u8(b.astype(np.uint8), (30, 20)), box.astype(np.uint8))
    up = io_.resize_linear_u8(b.astype(np.uint8), (120, 80))
    assert up.shape == (80, 120) and up.min() >= b.min() and up.max() <= b.max()
