"""Training-mode glue of the backbone on the device (csrc/train_glue.hip, round 5): BatchNorm with batch statistics,
activations (+ residual add) and the bilinear x2 upsampling -- forward and backward -- against the PyTorch ops the reference's backbone is
made of (src/loftr/backbone/resnet_fpn.py:22-40,66-77,110-116), evaluated in float64.  Floating-point kernels: the oracle is the torch
fp64 op; tolerance = a few x float32's own distance to it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, ref64, ref32=None, what="", k=4.0, floor=2e-6):
    scale = float(ref64.abs().max())
    err = float((got.double() - ref64).abs().max())
    noise = float((ref32.double() - ref64).abs().max()) if ref32 is not None else 0.0
    assert err <= k * noise + floor * scale, (what, err, noise, scale)


@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "channels_last"])
@pytest.mark.parametrize("shape,affine", [((3, 128, 37, 53), True), ((2, 196, 120, 160), True), ((4, 256, 15, 20), True), ((1, 128, 9, 7), False)])
def test_batch_norm_train_forward_backward(shape, affine, cl):
    from loftr_amd import autograd
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g) * 3 + 1.5
    x[:, :5] *= 40.0                                            # uneven channel magnitudes
    gamma = (1 + 0.3 * torch.randn(shape[1], generator=g)) if affine else None
    beta = (0.2 * torch.randn(shape[1], generator=g)) if affine else None
    dy = torch.randn(shape, generator=g) * 1e-2
    outs = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        xx = x.to("cuda", dt).requires_grad_(True)
        ww = None if gamma is None else gamma.to("cuda", dt).requires_grad_(True)
        bb = None if beta is None else beta.to("cuda", dt).requires_grad_(True)
        y = F.batch_norm(xx, None, None, ww, bb, True, 0.1, 1e-5)
        y.backward(dy.to("cuda", dt))
        outs[name] = (y.detach(), xx.grad, None if ww is None else ww.grad, None if bb is None else bb.grad)
    fmt = torch.channels_last if cl else torch.contiguous_format       # channels_last: what the convolution nodes hand over (no layout copy)
    xx = x.cuda().contiguous(memory_format=fmt).requires_grad_(True)
    ww = None if gamma is None else gamma.cuda().requires_grad_(True)
    bb = None if beta is None else beta.cuda().requires_grad_(True)
    y, mean, varu = autograd.batch_norm_train(xx, ww, bb, 1e-5)
    assert y.is_contiguous(memory_format=fmt)
    y.backward(dy.cuda().contiguous(memory_format=fmt))
    assert xx.grad.is_contiguous(memory_format=fmt)
    got = (y.detach(), xx.grad, None if ww is None else ww.grad, None if bb is None else bb.grad)
    for i, what in enumerate(("y", "dx", "dgamma", "dbeta")):
        if got[i] is not None:
            _close(got[i], outs["f64"][i], outs["f32"][i], what)
    x64 = x.double()
    _close(mean.cpu(), x64.mean((0, 2, 3)), None, "mean")
    _close(varu.cpu(), x64.var((0, 2, 3), unbiased=True), None, "unbiased variance", floor=5e-6)


def test_batch_norm_module_updates_running_statistics_like_torch():
    """backbone.BatchNorm2d in .train() mode: output, gradients and the running mean / variance / batch counter after two steps equal
    nn.BatchNorm2d's (momentum update with the unbiased batch variance, resnet_fpn.py: every bn of the backbone)."""
    from loftr_amd import backbone as BB
    torch.manual_seed(3)
    ref = torch.nn.BatchNorm2d(64).cuda().train()
    ours = BB.BatchNorm2d(64).cuda().train()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.normal_(0, 0.2)
    ours.load_state_dict(ref.state_dict())
    for step in range(2):
        x = (torch.randn(4, 64, 23, 31, device="cuda") * (1 + step) + 0.3)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = ref(xa), ours(xb)
        gy = torch.randn_like(ya)
        ya.backward(gy); yb.backward(gy)
        assert float((ya - yb).abs().max()) <= 1e-5 * float(ya.abs().max())           # torch: fp32 Welford statistics; here float64 sums
        assert float((xa.grad - xb.grad).abs().max()) <= 2e-5 * float(xa.grad.abs().max())
    assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked) == 2
    assert float((ours.running_mean - ref.running_mean).abs().max()) <= 1e-6
    assert float((ours.running_var - ref.running_var).abs().max()) <= 1e-5 * float(ref.running_var.abs().max())
    assert float((ours.weight.grad - ref.weight.grad).abs().max()) <= 1e-4 * float(ref.weight.grad.abs().max())
    # eval mode / no graph: the stock module
    ours.eval()
    with torch.no_grad():
        assert torch.equal(ours(x), torch.nn.functional.batch_norm(x, ours.running_mean, ours.running_var, ours.weight, ours.bias, False, 0.1, 1e-5))


@pytest.mark.parametrize("kind,slope,with_b", [("relu", 0.0, False), ("relu", 0.0, True), ("leaky_relu", 0.01, False), ("leaky_relu", 0.2, True)])
def test_activation_forward_backward_exact(kind, slope, with_b):
    from loftr_amd import autograd
    g = torch.Generator().manual_seed(7)
    fmt = torch.channels_last if with_b else torch.contiguous_format     # (elementwise: any dense layout; b is brought to a's)
    a = torch.randn(3, 16, 29, 31, generator=g).cuda().contiguous(memory_format=fmt).requires_grad_(True)
    b = torch.randn(3, 16, 29, 31, generator=g).cuda().requires_grad_(True) if with_b else None
    dy = torch.randn(3, 16, 29, 31, generator=g).cuda()
    s = a + b if with_b else a
    ref = torch.relu(s) if kind == "relu" else F.leaky_relu(s, slope)
    ref.backward(dy)
    ra, rb = a.grad.clone(), (b.grad.clone() if with_b else None)
    a.grad = None
    if with_b:
        b.grad = None
    y = autograd.act(a, b, kind, slope)
    y.backward(dy)
    assert torch.equal(y, ref) and torch.equal(a.grad, ra)
    if with_b:
        assert torch.equal(b.grad, rb)


@pytest.mark.parametrize("shape", [(2, 5, 60, 80), (1, 3, 1, 7), (2, 4, 15, 20), (1, 2, 7, 1), (2, 196, 30, 40), (1, 8, 9, 11)])
def test_upsample2x_bilinear_forward_and_adjoint(shape):
    """F.interpolate(scale_factor=2, bilinear, align_corners=True) (resnet_fpn.py:110,115): forward against torch in fp32 (same arithmetic, up to
    the compilers' contraction: 1e-6 relative) and fp64; the gather-form adjoint against torch.autograd in fp64 and through <up(x), g> == <x, up^T(g)>."""
    from loftr_amd import autograd
    g = torch.Generator().manual_seed(shape[2] * 100 + shape[3])
    x = torch.randn(shape, generator=g)
    dy = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], generator=g)
    x64 = x.double().cuda().requires_grad_(True)
    r64 = F.interpolate(x64, scale_factor=2., mode="bilinear", align_corners=True)
    r64.backward(dy.double().cuda())
    x32 = x.cuda().requires_grad_(True)
    r32 = F.interpolate(x32, scale_factor=2., mode="bilinear", align_corners=True)
    r32.backward(dy.cuda())
    r32 = r32.detach()
    cl = shape[1] % 4 == 0                                            # channel counts of the backbone: the channels-last kernels
    fmt = torch.channels_last if cl else torch.contiguous_format
    xx = x.cuda().contiguous(memory_format=fmt).requires_grad_(True)
    y = autograd.upsample2x(xx)
    y.backward(dy.cuda().contiguous(memory_format=fmt))
    assert y.shape == r32.shape and y.is_contiguous(memory_format=fmt)
    assert float((y - r32).abs().max()) <= 1e-6 * max(1.0, float(r32.abs().max())), float((y - r32).abs().max())      # a few ulp: the compilers' contraction of the lerp
    _close(y.detach(), r64.detach(), r32, "up")
    _close(xx.grad, x64.grad, x32.grad, "adjoint", floor=3e-6)            # (the float32 interpolation weights carry ~1e-5 of noise: torch's own backward too)
    lhs, rhs = float((y.detach().double() * dy.double().cuda()).sum()), float((x.double().cuda() * xx.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))


def test_backbone_training_forward_uses_the_hip_glue(monkeypatch):
    """A training-mode forward / backward of the whole ResNet-FPN: every BatchNorm, activation and upsampling of the graph is a HIP node
    (no aten batch_norm / relu / upsample_bilinear2d in the autograd graph), and the result agrees with the all-PyTorch graph of the same
    module (LOFTR_TRAIN_GLUE = 0 path) to float32 accuracy."""
    from loftr_amd import backbone as BB, LoFTR, get_cfg
    torch.manual_seed(0)
    net = LoFTR(get_cfg(thr=0.0)).backbone.cuda().train()
    x = torch.rand(2, 1, 96, 128, device="cuda")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    res = {}
    for glue in (True, False):
        monkeypatch.setattr(BB, "TRAIN_GLUE_HIP", glue)
        net.load_state_dict(sd)
        net.zero_grad()
        fc, ff = net(x)
        names = set()
        stack, seen = [fc.grad_fn, ff.grad_fn], set()
        while stack:
            f = stack.pop()
            if f is None or f in seen:
                continue
            seen.add(f); names.add(type(f).__name__)
            stack.extend(n for n, _ in f.next_functions)
        (fc.square().mean() + ff.square().mean()).backward()
        res[glue] = (fc.detach(), ff.detach(), net.layer1[0].conv1.weight.grad.clone(), net.bn1.weight.grad.clone(), net.bn1.running_var.clone(), names)
    on, off = res[True][5], res[False][5]
    assert not any("BatchNorm" in n and "Backward" in n and not n.startswith("_") for n in on), on
    assert not any(n.startswith(("ReluBackward", "LeakyReluBackward", "UpsampleBilinear2D", "NativeBatchNormBackward", "CudnnBatchNormBackward", "MiopenBatchNormBackward")) for n in on), on
    assert any(n.startswith(("NativeBatchNormBackward", "CudnnBatchNormBackward", "MiopenBatchNormBackward")) for n in off), off
    # forward quantities to 1e-4; gradients of the early layers are ill-conditioned at float32 (two evaluations of the same graph differ by
    # percent through single ReLU sign changes, profiles/r05_relu_flip_probe.txt; measured here: 3.7e-3): a 1e-1 gross-error guard here, the accuracy statement is the float64 comparison
    # of tools/micro/glue_vs_fp64.py (profiles/r05_glue_vs_fp64.txt) and of the hipglue variant of tests/test_hip_training.py
    for i, (what, tol) in enumerate((("feat_c", 1e-4), ("feat_f", 1e-4), ("dW layer1.0.conv1", 1e-1), ("dgamma bn1", 1e-1), ("running_var bn1", 1e-4))):
        a, b = res[True][i], res[False][i]
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), (what, float((a - b).abs().max()), float(b.abs().max()))
