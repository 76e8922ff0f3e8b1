"""TEST INFRASTRUCTURE (not the product): numpy restatement, in float64, of the training-mode glue of the reference's backbone that
csrc/train_glue.hip implements on the device.  Pinned against the PyTorch ops the reference's modules call (tests/test_glue_oracle.py, CPU):

  * bn_train_fwd / bn_train_bwd -- nn.BatchNorm2d in .train() mode (src/loftr/backbone/resnet_fpn.py:25-26,36,68 ...: every BatchNorm of the
    ResNet-FPN; F.batch_norm(training=True)): batch mean and BIASED variance over (N, H, W) normalise, the UNBIASED variance feeds the running
    estimate (torch/nn/modules/batchnorm.py: momentum update);
  * act_fwd / act_bwd -- nn.ReLU / nn.LeakyReLU and BasicBlock's relu(x + y) (resnet_fpn.py:33-40, :66-77);
  * upsample2x_fwd / upsample2x_bwd -- F.interpolate(x, scale_factor=2., mode='bilinear', align_corners=True) (resnet_fpn.py:110,115):
    src = dst * (in - 1) / (out - 1), the four-tap interpolation, and its adjoint.
"""
import numpy as np


def bn_train_fwd(x, gamma=None, beta=None, eps=1e-5):
    """x [N,C,H,W] -> (y, mean [C], invstd [C], unbiased variance [C])."""
    x = np.asarray(x, np.float64)
    m = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.mean(axis=(0, 2, 3))
    var = x.var(axis=(0, 2, 3))                                  # biased: what normalises
    invstd = 1.0 / np.sqrt(var + eps)
    g = np.ones_like(mean) if gamma is None else np.asarray(gamma, np.float64)
    b = np.zeros_like(mean) if beta is None else np.asarray(beta, np.float64)
    y = (x - mean[None, :, None, None]) * (invstd * g)[None, :, None, None] + b[None, :, None, None]
    return y, mean, invstd, var * m / max(m - 1, 1)


def bn_train_bwd(dy, x, mean, invstd, gamma=None):
    """(dx, dgamma, dbeta): the batch statistics depend on x."""
    dy, x = np.asarray(dy, np.float64), np.asarray(x, np.float64)
    m = x.shape[0] * x.shape[2] * x.shape[3]
    xh = (x - mean[None, :, None, None]) * invstd[None, :, None, None]
    dbeta = dy.sum(axis=(0, 2, 3))
    dgamma = (dy * xh).sum(axis=(0, 2, 3))
    g = np.ones_like(mean) if gamma is None else np.asarray(gamma, np.float64)
    dx = (g * invstd)[None, :, None, None] * (dy - dbeta[None, :, None, None] / m - xh * dgamma[None, :, None, None] / m)
    return dx, dgamma, dbeta


def act_fwd(a, b=None, kind="relu", slope=0.01):
    v = np.asarray(a, np.float64) + (0.0 if b is None else np.asarray(b, np.float64))
    if kind == "relu":
        return np.maximum(v, 0.0)
    if kind == "leaky_relu":
        return np.where(v > 0, v, slope * v)
    return v


def act_bwd(dy, y, kind="relu", slope=0.01):
    """From the forward's OUTPUT (its sign is the input's for a positive slope)."""
    dy = np.asarray(dy, np.float64)
    if kind == "relu":
        return np.where(np.asarray(y) > 0, dy, 0.0)
    if kind == "leaky_relu":
        return np.where(np.asarray(y) > 0, dy, slope * dy)
    return dy


def _taps(n_in):
    """For every output index of a x2 upsampling of n_in samples (align_corners): (i0, i1, weight of i0, weight of i1)."""
    n_out = 2 * n_in
    scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    src = scale * np.arange(n_out)
    i0 = np.floor(src).astype(np.int64)
    i1 = i0 + (i0 < n_in - 1)
    l1 = src - i0
    return i0, i1, 1.0 - l1, l1


def upsample2x_fwd(x):
    x = np.asarray(x, np.float64)
    y0, y1, hy, ly = _taps(x.shape[2])
    x0, x1, hx, lx = _taps(x.shape[3])
    rows = hy[None, None, :, None] * x[:, :, y0] + ly[None, None, :, None] * x[:, :, y1]
    return hx[None, None, None, :] * rows[:, :, :, x0] + lx[None, None, None, :] * rows[:, :, :, x1]


def upsample2x_bwd(dy):
    """The adjoint of upsample2x_fwd: dx[yi, xi] = sum over the output pixels that read (yi, xi) of weight * dy."""
    dy = np.asarray(dy, np.float64)
    H, W = dy.shape[2] // 2, dy.shape[3] // 2
    y0, y1, hy, ly = _taps(H)
    x0, x1, hx, lx = _taps(W)
    cols = np.zeros(dy.shape[:3] + (W,))
    np.add.at(cols, (slice(None), slice(None), slice(None), x0), dy * hx[None, None, None, :])
    np.add.at(cols, (slice(None), slice(None), slice(None), x1), dy * lx[None, None, None, :])
    dx = np.zeros(dy.shape[:2] + (H, W))
    np.add.at(dx, (slice(None), slice(None), y0), cols * hy[None, None, :, None])
    np.add.at(dx, (slice(None), slice(None), y1), cols * ly[None, None, :, None])
    return dx
