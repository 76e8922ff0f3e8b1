"""oracle/glue_oracle.py (numpy float64 restatement of the training-mode glue of the backbone) against the PyTorch ops the reference's modules
call, on the CPU in float64: nn.BatchNorm2d in train mode, ReLU / LeakyReLU (+ the residual add), F.interpolate(scale_factor=2, bilinear,
align_corners=True) -- src/loftr/backbone/resnet_fpn.py:22-40,66-77,110-116.  The HIP kernels (csrc/train_glue.hip) are held to the same torch
float64 ops in tests/test_hip_train_glue.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import glue_oracle as G


@pytest.mark.parametrize("shape,affine", [((3, 6, 7, 9), True), ((2, 4, 1, 5), True), ((4, 5, 3, 3), False)])
def test_batch_norm_train(shape, affine):
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape) * 3 + 1.5
    gamma = 1 + 0.3 * rng.standard_normal(shape[1]) if affine else None
    beta = 0.2 * rng.standard_normal(shape[1]) if affine else None
    dy = rng.standard_normal(shape)
    xt = torch.tensor(x, requires_grad=True)
    gt = None if gamma is None else torch.tensor(gamma, requires_grad=True)
    bt = None if beta is None else torch.tensor(beta, requires_grad=True)
    rm, rv = torch.zeros(shape[1], dtype=torch.float64), torch.ones(shape[1], dtype=torch.float64)
    yt = F.batch_norm(xt, rm, rv, gt, bt, True, 0.1, 1e-5)
    yt.backward(torch.tensor(dy))
    y, mean, invstd, varu = G.bn_train_fwd(x, gamma, beta, 1e-5)
    dx, dgamma, dbeta = G.bn_train_bwd(dy, x, mean, invstd, gamma)
    assert np.allclose(y, yt.detach().numpy(), rtol=1e-12, atol=1e-12)
    assert np.allclose(dx, xt.grad.numpy(), rtol=1e-10, atol=1e-12)
    if affine:
        assert np.allclose(dgamma, gt.grad.numpy(), rtol=1e-10) and np.allclose(dbeta, bt.grad.numpy(), rtol=1e-10)
    # the running estimates torch derives from this batch: momentum 0.1 on (0, 1): mean and the UNBIASED variance
    assert np.allclose(rm.numpy(), 0.1 * mean, rtol=1e-12) and np.allclose(rv.numpy(), 0.9 + 0.1 * varu, rtol=1e-12)


@pytest.mark.parametrize("kind,slope,with_b", [("relu", 0.0, False), ("relu", 0.0, True), ("leaky_relu", 0.01, True), ("leaky_relu", 0.2, False)])
def test_activation(kind, slope, with_b):
    rng = np.random.default_rng(5)
    a, b, dy = (rng.standard_normal((2, 3, 5, 4)) for _ in range(3))
    at, bt = torch.tensor(a, requires_grad=True), torch.tensor(b, requires_grad=True)
    s = at + bt if with_b else at
    yt = torch.relu(s) if kind == "relu" else F.leaky_relu(s, slope)
    yt.backward(torch.tensor(dy))
    y = G.act_fwd(a, b if with_b else None, kind, slope)
    assert np.array_equal(y, yt.detach().numpy())
    assert np.array_equal(G.act_bwd(dy, y, kind, slope), at.grad.numpy())
    if with_b:
        assert np.array_equal(at.grad.numpy(), bt.grad.numpy())           # the add hands the same gradient to both


@pytest.mark.parametrize("shape", [(2, 3, 6, 8), (1, 2, 1, 5), (1, 1, 7, 1), (2, 2, 15, 20)])
def test_upsample2x_bilinear(shape):
    rng = np.random.default_rng(shape[2] * 10 + shape[3])
    x, dy = rng.standard_normal(shape), rng.standard_normal(shape[:2] + (2 * shape[2], 2 * shape[3]))
    xt = torch.tensor(x, requires_grad=True)
    yt = F.interpolate(xt, scale_factor=2., mode="bilinear", align_corners=True)
    yt.backward(torch.tensor(dy))
    assert np.allclose(G.upsample2x_fwd(x), yt.detach().numpy(), rtol=1e-12, atol=1e-13)
    assert np.allclose(G.upsample2x_bwd(dy), xt.grad.numpy(), rtol=1e-11, atol=1e-13)
