"""ResNet-FPN backbone against goldens produced by the REFERENCE's own modules
(tests/golden/make_golden_backbone.py): pins the architecture (layer order, strides, BN eps, LeakyReLU slope,
align_corners upsampling, parameter layout) of loftr_amd/backbone.py, which tests/test_hip_backbone.py then uses as
the fp64 yardstick of the HIP convolutions -- and, on the GPU, the HIP path directly."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_backbone import CASES, backbone_cfg       # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "backbone.npz")


def _model_and_input(name):
    from loftr_amd.backbone import build_backbone
    from loftr_amd.synth import make_backbone_weights
    res, dims, wseed, xseed, shape = CASES[name]
    m = build_backbone(backbone_cfg(res, dims)).eval()
    m.load_state_dict(make_backbone_weights(wseed, m), strict=True)      # same names and shapes as the reference's
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(xseed))
    return m, x


@pytest.mark.parametrize("name", list(CASES))
def test_torch_forward_matches_reference_backbone(name):
    g = np.load(GOLD)
    m, x = _model_and_input(name)
    assert sum(p.numel() for p in m.parameters()) == int(g[f"{name}_nparams"])
    with torch.no_grad():
        c, f = m(x)
    for got, key in ((c, "coarse"), (f, "fine")):
        ref = g[f"{name}_{key}"]
        assert tuple(got.shape) == ref.shape
        # same operator sequence on the same CPU kernels: equal up to thread-count dependent summation order
        assert np.abs(got.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_forward_matches_reference_backbone(name):
    g = np.load(GOLD)
    m, x = _model_and_input(name)
    m = m.to("cuda:0").to(memory_format=torch.channels_last)
    with torch.no_grad():
        c, f = m.forward_hip(x.to("cuda:0"))
    for got, key in ((c, "coarse"), (f, "fine")):
        ref = g[f"{name}_{key}"]
        assert tuple(got.shape) == ref.shape
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()      # fp32-class (split-fp16 MFMA) vs fp32 CPU


def test_training_conv_module_is_plain_conv_off_the_gpu():
    """backbone.Conv2d (the module whose .train()-mode GPU forward is the HIP autograd node) must be nn.Conv2d everywhere else:
    same parameters and state-dict keys as the reference's conv1x1 / conv3x3 (resnet_fpn.py:5-13), the stock forward on the CPU in
    both modes with gradients, and a loud error -- not a silent fallback -- if the node were asked to run on CPU tensors."""
    import torch.nn as nn
    from loftr_amd import autograd, backbone
    c = backbone._c3(4, 8, stride=2)
    ref = nn.Conv2d(4, 8, kernel_size=3, stride=2, padding=1, bias=False)
    assert isinstance(c, nn.Conv2d) and list(c.state_dict().keys()) == list(ref.state_dict().keys()) == ["weight"]
    ref.load_state_dict(c.state_dict())
    x = torch.randn(2, 4, 9, 11, requires_grad=True)
    for mode in (True, False):
        c.train(mode)
        y = c(x)
        assert torch.equal(y, ref(x)) and y.grad_fn is not None
    calls = autograd._Conv2d.calls
    c.train()
    c(x).sum().backward()
    assert autograd._Conv2d.calls == calls and c.weight.grad is not None          # the HIP node did not run on the CPU
    with pytest.raises(Exception):
        autograd.conv2d(x, c.weight, 2, 1)                                         # CPU tensors: the ops raise (no fallback)
