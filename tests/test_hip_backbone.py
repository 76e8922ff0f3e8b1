"""GPU tests of the HIP ResNet-FPN path: implicit-GEMM convolutions with folded
eval BatchNorm / residual / activation, FPN upsample+add, and the whole backbone against PyTorch in fp64.

Tolerance: features are O(1); the split-fp16 GEMM core is fp32-class, so the HIP path must
sit as close to an fp64 evaluation as MIOpen's own fp32 kernels do (bounded here at 5e-5 of the feature
scale) -- far inside what the 1e-4 / 1e-3 px matching tolerances need."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _randomize_bn(m, g):
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            n = mod.num_features
            mod.weight.data = 1.0 + 0.2 * torch.randn(n, generator=g)
            mod.bias.data = 0.1 * torch.randn(n, generator=g)
            mod.running_mean.data = 0.1 * torch.randn(n, generator=g)
            mod.running_var.data = 0.5 + torch.rand(n, generator=g)


@pytest.mark.parametrize("cin,cout,k,stride,act,use_bn,use_res", [
    (128, 128, 3, 1, 1, True, False),
    (128, 196, 3, 2, 1, True, False),
    (196, 196, 3, 1, 1, True, True),
    (128, 196, 1, 2, 0, True, False),
    (196, 256, 1, 1, 0, False, False),
    (256, 196, 3, 1, 0, False, False),
    (256, 256, 3, 1, 2, True, False),
    (64, 40, 3, 1, 2, True, True),
    (96, 200, 3, 1, 2, True, True),          # 7 column tiles with Cout != 196: the 224-column kernel, 3 channel groups
    (32, 128, 3, 1, 1, True, True),          # one channel group: two half-group steps (conv3x3_duo.h)
    (16, 128, 3, 1, 0, False, False),        # ... whose second half is all padding: nine steps in total
    (196, 128, 3, 1, 0, False, False),       # dead last half at 128 columns
    (200, 200, 3, 1, 1, True, False),        # 7 column tiles, live last half
])
def test_conv_bn_act_vs_torch(cin, cout, k, stride, act, use_bn, use_res):
    from loftr_amd import ops
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    B, H, W = 2, 22, 30
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)
    conv.weight.data = torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bn = nn.BatchNorm2d(cout).eval() if use_bn else None
    if bn is not None:
        _randomize_bn(bn, g)
    x = torch.randn(B, cin, H, W, generator=g)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, cout, Ho, Wo, generator=g) if use_res else None
    # fp64 reference
    y = F.conv2d(x.double(), conv.weight.double(), stride=stride, padding=k // 2)
    if bn is not None:
        y = F.batch_norm(y, bn.running_mean.double(), bn.running_var.double(), bn.weight.double(), bn.bias.double(),
                         False, 0.0, bn.eps)
    if res is not None:
        y = y + res.double()
    y = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.01)}[act](y)
    # HIP
    dev = "cuda:0"
    conv, bn = conv.to(dev), (bn.to(dev) if bn is not None else None)
    x_sp = ops.sp_from_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev))
    r_sp = ops.sp_from_nhwc(res.permute(0, 2, 3, 1).contiguous().to(dev)) if res is not None else None
    y_sp, y_f32 = ops.conv_bn_act(x_sp, cin, conv, bn, act=act, residual=r_sp, want_sp=True, want_f32=True)
    got = y_f32.permute(0, 3, 1, 2).cpu().double()
    scale = y.abs().max().item()
    assert (got - y).abs().max().item() <= 2e-5 * scale, ((got - y).abs().max().item(), scale)
    # the SP output decodes to the same values (<= 2^-21 relative) and its pad channels are zero
    back = ops.sp_to_nhwc(y_sp, cout).permute(0, 3, 1, 2).cpu().double()
    assert (back - got).abs().max().item() <= 1e-6 * scale
    if y_sp.shape[-1] != cout:
        full = ops.sp_to_nhwc(y_sp, y_sp.shape[-1]).cpu()
        assert (full[..., cout:] == 0).all()


def test_prepared_weights_follow_inplace_updates():
    """The folded SP filter is cached on the module (ops._prepared_conv); an in-place weight / BN update or a
    replaced tensor must rebuild it (tensor._version / data_ptr are part of the cache key)."""
    from loftr_amd import ops
    g = torch.Generator().manual_seed(11)
    conv = nn.Conv2d(64, 64, 3, padding=1, bias=False).cuda()
    bn = nn.BatchNorm2d(64).eval().cuda()
    x = torch.randn(1, 64, 16, 32, generator=g).cuda()
    x_sp = ops.sp_from_nhwc(x.permute(0, 2, 3, 1).contiguous())

    def run():
        return ops.conv_bn_act(x_sp, 64, conv, bn, act=0, want_sp=False, want_f32=True)[1].permute(0, 3, 1, 2)

    def ref():
        return bn(conv(x))

    with torch.no_grad():
        y0 = run()
        assert (y0 - ref()).abs().max().item() < 1e-4
        first = ops._PREPARED[conv][1]
        assert run() is not None and ops._PREPARED[conv][1] is first          # second call: cache hit
        conv.weight.mul_(2.0)                                                    # in place -> _version bump
        assert (run() - ref()).abs().max().item() < 1e-4 and ops._PREPARED[conv][1] is not first
        bn.running_var.add_(1.0)
        assert (run() - ref()).abs().max().item() < 1e-4
        conv.weight.data = torch.randn(64, 64, 3, 3, generator=g).cuda() * 0.05  # replaced storage
        assert (run() - ref()).abs().max().item() < 1e-4
    # the cache lives in a module-keyed weak registry, not in the module: a used module still pickles
    import pickle
    assert "_loftr_prepared" not in conv.__dict__ and pickle.loads(pickle.dumps(conv)).weight.shape == conv.weight.shape


def test_prepared_weights_follow_tensor_identity():
    """A parameter REPLACED by a new tensor object at the same address with the same version counter (what the caching
    allocator can hand back after the old one is freed) must not hit the cached folded weights."""
    from loftr_amd import ops
    conv = nn.Conv2d(32, 32, 3, padding=1, bias=False).cuda()
    buf1 = ops._prepared_conv(conv, None)
    assert ops._prepared_conv(conv, None) is buf1
    old = conv.weight
    conv.weight = nn.Parameter(old.detach())             # same storage, same data_ptr, same _version: a different object
    assert conv.weight.data_ptr() == old.data_ptr() and conv.weight._version == old._version
    assert ops._prepared_conv(conv, None) is not buf1


def test_stem_vs_torch():
    from loftr_amd import ops
    g = torch.Generator().manual_seed(9)
    conv = nn.Conv2d(1, 128, 7, stride=2, padding=3, bias=False)
    conv.weight.data = torch.randn(conv.weight.shape, generator=g) * 0.1
    bn = nn.BatchNorm2d(128).eval()
    _randomize_bn(bn, g)
    for hw in ((96, 128), (50, 70), (480, 640)):
        x = torch.rand(2, 1, *hw, generator=g)
        ref = F.relu(F.batch_norm(F.conv2d(x.double(), conv.weight.double(), stride=2, padding=3), bn.running_mean.double(),
                                  bn.running_var.double(), bn.weight.double(), bn.bias.double(), False, 0.0, bn.eps))
        y = ops.stem_conv_bn_relu(x.cuda(), conv.cuda(), bn.cuda())
        got = ops.sp_to_nhwc(y, 128).permute(0, 3, 1, 2).cpu().double()
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
        conv, bn = conv.cpu(), bn.cpu()


def test_upsample2x_add_vs_torch():
    from loftr_amd import ops
    g = torch.Generator().manual_seed(3)
    for C in (196, 256):
        low = torch.randn(2, C, 15, 20, generator=g)
        lat = torch.randn(2, C, 30, 40, generator=g)
        ref = lat.double() + F.interpolate(low.double(), scale_factor=2.0, mode="bilinear", align_corners=True)
        dev = "cuda:0"
        l_sp = ops.sp_from_nhwc(low.permute(0, 2, 3, 1).contiguous().to(dev))
        a_sp = ops.sp_from_nhwc(lat.permute(0, 2, 3, 1).contiguous().to(dev))
        out = ops.sp_to_nhwc(ops.upsample2x_add(l_sp, a_sp, C), C).permute(0, 3, 1, 2).cpu().double()
        assert (out - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("Cin,Cout,hw", [(128, 196, (30, 40)), (196, 256, (12, 16)), (128, 128, (34, 22))])
def test_conv1x1_upsample_add_vs_torch(Cin, Cout, hw):
    """One FPN top-down step in one launch (resnet_fpn.py:110-112): lateral 1x1 conv + bilinear x2 of the coarser
    map, against fp64 torch; covers a ragged last row tile, a padded channel count and a 2-column-tile Cout."""
    from loftr_amd import ops
    g = torch.Generator().manual_seed(5)
    H, W = hw
    x = torch.randn(2, Cin, H, W, generator=g)
    low = torch.randn(2, Cout, H // 2, W // 2, generator=g)
    conv = torch.nn.Conv2d(Cin, Cout, 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / Cin ** 0.5)
        ref = F.conv2d(x.double(), conv.weight.double()) + F.interpolate(low.double(), scale_factor=2.0, mode="bilinear",
                                                                          align_corners=True)
    dev = "cuda:0"
    conv = conv.to(dev)
    x_sp = ops.sp_from_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev))
    l_sp = ops.sp_from_nhwc(low.permute(0, 2, 3, 1).contiguous().to(dev))
    y = ops.conv1x1_upsample_add(x_sp, Cin, conv, l_sp)
    assert y.shape == (2, H, W, (Cout + 31) // 32 * 32)
    out = ops.sp_to_nhwc(y, Cout).permute(0, 3, 1, 2).cpu().double()
    assert (out - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
    if Cout % 32:                                   # pad channels of the SP row stay exactly zero
        pad = ops.sp_to_nhwc(y, y.shape[-1])[..., Cout:]
        assert pad.abs().max().item() == 0.0
    with pytest.raises(Exception):
        ops.conv1x1_upsample_add(x_sp, Cin, conv, l_sp[:, :-1].contiguous())


@pytest.mark.parametrize("resolution,dims,hw", [((8, 2), [128, 196, 256], (96, 128)),
                                               ((16, 4), [128, 196, 256, 512], (96, 128))])
def test_backbone_hip_vs_fp64(resolution, dims, hw):
    from loftr_amd.backbone import build_backbone
    cfg = {"backbone_type": "ResNetFPN", "resolution": resolution, "resnetfpn": {"initial_dim": 128, "block_dims": dims}}
    torch.manual_seed(0)
    m = build_backbone(cfg).eval()
    _randomize_bn(m, torch.Generator().manual_seed(1))
    x = torch.rand(2, 1, *hw, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = [t.clone() for t in m.double()(x.double())]
    m = m.float().to("cuda:0").to(memory_format=torch.channels_last)
    xc = x.to("cuda:0").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        hip = m.forward_hip(xc)
        tor = m(xc)
    for name, r, h, t in zip(("coarse", "fine"), ref, hip, tor):
        assert h.shape == r.shape and h.dtype == torch.float32
        scale = r.abs().max().item()
        e_hip = (h.cpu().double() - r).abs().max().item() / scale
        e_tor = (t.cpu().double() - r).abs().max().item() / scale
        print(f"{name}: hip err {e_hip:.2e}  miopen-fp32 err {e_tor:.2e}  (relative to max |feature| = {scale:.3g})")
        assert e_hip <= 5e-5, (name, e_hip, e_tor)


def test_full_forward_hip_backbone_matches_torch_backbone():
    """End to end: LoFTR.forward with the HIP backbone vs the MIOpen backbone on the same weights / images.
    The two backbones differ at the 1e-5 level, so matches agree except provably borderline ones."""
    from loftr_amd import LoFTR, get_cfg
    torch.manual_seed(0)
    model = LoFTR(get_cfg(thr=0.0)).eval().cuda()
    g = torch.Generator().manual_seed(5)
    img0 = torch.rand(2, 1, 240, 320, generator=g).cuda()
    img1 = torch.roll(img0, (8, 16), (2, 3)) + 0.02 * torch.rand(2, 1, 240, 320, generator=g).cuda()
    outs = {}
    for impl in ("torch", "hip"):
        model.backbone_impl = impl
        d = {"image0": img0, "image1": img1}
        model(d)
        outs[impl] = d
    a, b = outs["torch"], outs["hip"]
    assert (a["conf_matrix"] - b["conf_matrix"]).abs().max().item() <= 1e-3
    ka = set(zip(a["b_ids"].tolist(), a["i_ids"].tolist(), a["j_ids"].tolist()))
    kb = set(zip(b["b_ids"].tolist(), b["i_ids"].tolist(), b["j_ids"].tolist()))
    assert len(ka ^ kb) <= max(2, len(ka) // 50), (len(ka), len(kb), len(ka ^ kb))


def test_two_stream_overlap_is_bitwise_identical():
    """The FPN fine branch on a second HIP stream (LoFTR.overlap_fine_branch) must not change a single bit, also
    when forwards run back to back (caching-allocator reuse across the two streams)."""
    from loftr_amd import LoFTR, get_cfg
    torch.manual_seed(0)
    model = LoFTR(get_cfg(thr=0.0)).eval().cuda()
    g = torch.Generator().manual_seed(7)
    imgs = [(torch.rand(2, 1, 240, 320, generator=g).cuda(), torch.rand(2, 1, 240, 320, generator=g).cuda()) for _ in range(3)]
    ref = []
    model.overlap_fine_branch = False
    for a, b in imgs:
        d = {"image0": a, "image1": b}
        model(d)
        ref.append({k: d[k].clone() for k in ("conf_matrix", "mkpts1_f", "mconf", "j_ids")})
    model.overlap_fine_branch = True
    for rep in range(3):
        for (a, b), r in zip(imgs, ref):
            d = {"image0": a, "image1": b}
            model(d)
            junk = torch.empty(64 << 20, device="cuda").fill_(float("nan"))    # provoke allocator reuse
            del junk
            for k in r:
                assert torch.equal(d[k], r[k]), (rep, k)


def test_sp_format_round_trip():
    """fp32 -> SP (fp16 hi + fp16 lo per value, 32-channel groups) -> fp32: |x - (hi + lo)| <= 2^-21 |x| inside the
    fp16 range, exact zeros in the channel padding, and the documented memory layout (hi halves then lo halves)."""
    from loftr_amd import ops
    g = torch.Generator().manual_seed(11)
    for C in (128, 196, 40):
        x = (torch.randn(3, 7, 9, C, generator=g) * torch.logspace(-3, 3, C)).contiguous()
        sp = ops.sp_from_nhwc(x.cuda())
        Cp = (C + 31) // 32 * 32
        assert sp.shape == (3, 7, 9, Cp) and sp.dtype == torch.int32
        back = ops.sp_to_nhwc(sp, C).cpu()
        assert ((back - x).abs() <= 2.0 ** -21 * x.abs() + 1e-7).all()
        # layout: per 32-channel group 16 dwords of hi halves (channels 2k, 2k+1) then 16 dwords of lo halves
        raw = sp.cpu().numpy().view(np.uint32).reshape(-1, Cp // 32, 2, 16)
        halves = raw.view(np.float16).reshape(-1, Cp // 32, 2, 32).astype(np.float32)
        recon = (halves[:, :, 0, :] + halves[:, :, 1, :]).reshape(-1, Cp)
        assert np.array_equal(recon[:, :C], back.numpy().reshape(-1, C))
        assert (recon[:, C:] == 0).all()
        hi = halves[:, :, 0, :].reshape(-1, Cp)[:, :C]
        assert np.array_equal(hi, x.numpy().reshape(-1, C).astype(np.float16).astype(np.float32))


def test_backbone_hip_fullsize_vs_torch_fp32():
    """640x480 inputs: 600 / 150 / 40 tiles per 3x3 layer, i.e. several tiles per persistent workgroup, the
    cross-tile DMA prologue and every ragged / full tile mix of the real workload, against the MIOpen fp32 path."""
    from loftr_amd.backbone import build_backbone
    cfg = {"backbone_type": "ResNetFPN", "resolution": (8, 2), "resnetfpn": {"initial_dim": 128, "block_dims": [128, 196, 256]}}
    torch.manual_seed(0)
    m = build_backbone(cfg).eval()
    _randomize_bn(m, torch.Generator().manual_seed(1))
    m = m.to("cuda:0").to(memory_format=torch.channels_last)
    x = torch.rand(2, 1, 480, 640, generator=torch.Generator().manual_seed(2)).to("cuda:0")
    with torch.no_grad():
        hip = m.forward_hip(x)
        tor = m(x)
    for name, h, t in zip(("coarse", "fine"), hip, tor):
        assert h.shape == t.shape
        scale = t.abs().max().item()
        err = (h - t).abs().max().item() / scale
        assert err <= 2e-5, (name, err)          # two fp32-class evaluations of a 20-layer network


def test_conv3x3_many_tiles_per_workgroup():
    """The generic persistent 3x3 kernel (conv3x3_kernel: any Cout; debug switch conv_duo = 0 sends the backbone's widths there too)
    with its grid capped at 8 workgroups (conv_persist_cap = 8): 48 tiles, six per workgroup, including the last partly filled
    round.  Until round 5 these were environment variables read once per process (subprocess test); the library reads none now."""
    from loftr_amd import ops
    g = torch.Generator().manual_seed(3)
    with ops.debug_switch(conv_persist_cap=8, conv_duo=0):
        for cin, cout in ((64, 128), (96, 200), (128, 256)):
            conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
            conv.weight.data = torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * 9)) ** 0.5
            x = torch.randn(2, cin, 60, 100, generator=g)
            ref = F.conv2d(x.double(), conv.weight.double(), padding=1)
            x_sp = ops.sp_from_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda())
            y = ops.conv_bn_act(x_sp, cin, conv.cuda(), None, want_sp=False, want_f32=True)[1].permute(0, 3, 1, 2).cpu().double()
            err = (y - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 2e-5, (cin, cout, err)
    assert ops.debug_get("conv_duo") == (1, 1) and ops.debug_get("conv_persist_cap") == (0, 0)


def test_conv3x3_every_surviving_path_agrees():
    """Round-5 verdict (weak #8): every path a debug switch can select is exercised.  One 3x3 / stride-1 layer per column-tile family
    (128 k, 192, 224 columns) through conv3x3_duo_kernel (default), the generic conv3x3_kernel (conv_duo = 0) and the implicit-GEMM
    conv_kernel that the strided / 1x1 layers use (conv_patch = 0): all three against fp64, and against each other to fp32 noise."""
    from loftr_amd import ops
    g = torch.Generator().manual_seed(5)
    for cin, cout in ((128, 128), (196, 192), (196, 196), (256, 256)):
        conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        conv.weight.data = torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * 9)) ** 0.5
        x = torch.randn(2, cin, 30, 40, generator=g)
        ref = F.conv2d(x.double(), conv.weight.double(), padding=1)
        x_sp = ops.sp_from_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda())
        conv = conv.cuda()
        outs = {}
        for name, sw in (("duo", {}), ("generic", dict(conv_duo=0)), ("implicit_gemm", dict(conv_patch=0))):
            with ops.debug_switch(**sw):
                y = ops.conv_bn_act(x_sp, cin, conv, None, want_sp=False, want_f32=True)[1].permute(0, 3, 1, 2).cpu().double()
            err = (y - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 2e-5, (cin, cout, name, err)
            outs[name] = y
        scale = ref.abs().max().item()
        assert (outs["duo"] - outs["generic"]).abs().max().item() <= 4e-6 * scale
        assert (outs["duo"] - outs["implicit_gemm"]).abs().max().item() <= 4e-6 * scale


def test_conv3x3_duo_ragged_tiles():
    """conv3x3_duo_kernel (two 4-wave workgroups per CU): image sizes that leave partly filled 8 x 32 / 4 x 32 tiles on both
    axes, several tiles per image and both column-tile counts; against fp64."""
    from loftr_amd import ops
    g = torch.Generator().manual_seed(11)
    for cin, cout, (H, W) in ((64, 128, (37, 70)), (128, 256, (17, 33)), (96, 200, (9, 65)), (196, 196, (30, 40))):
        conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        conv.weight.data = torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * 9)) ** 0.5
        x = torch.randn(3, cin, H, W, generator=g)
        ref = F.conv2d(x.double(), conv.weight.double(), padding=1)
        x_sp = ops.sp_from_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda())
        y = ops.conv_bn_act(x_sp, cin, conv.cuda(), None, want_sp=False, want_f32=True)[1].permute(0, 3, 1, 2).cpu().double()
        err = (y - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-5, (cin, cout, H, W, err)


def test_backbone_chunking_is_bitwise_identical():
    """Batches whose activation tensors would exceed the kernels' 32-bit indexing go through the backbone in chunks
    (loftr.py:run_backbone); forced here with a tiny chunk size: same bytes as the single pass."""
    from loftr_amd import LoFTR
    from loftr_amd.config import get_cfg
    torch.manual_seed(3)
    model = LoFTR(get_cfg(thr=0.0)).eval().cuda()
    g = torch.Generator().manual_seed(5)
    data = lambda: {"image0": torch.rand(3, 1, 64, 96, generator=g.manual_seed(5)).cuda(), "image1": torch.rand(3, 1, 64, 96, generator=g.manual_seed(6)).cuda()}
    a = data(); model(a)
    model._backbone_chunk_images = 4                          # 6 images -> chunks of 4 + 2
    b = data(); model(b)
    for k in ("conf_matrix", "mkpts0_f", "mkpts1_f", "mconf", "b_ids"):
        assert torch.equal(a[k], b[k]), k


def test_conv3x3_remainder_channels_tap_decomposed():
    """Round 6 (round-5 verdict, next #5a): Cout = 196 pads to 224 columns, the 7th column tile does a tile's matrix work for 4 channels.
    conv3x3_duo_kernel<Cfg<6, 2, 4, 8, 2, REM>> computes 192 columns and the channels beyond them as a tap-decomposed product (all (tap, channel)
    pairs = the columns of one K = Cin product on the unshifted pixels, at the centre-tap steps) + conv_rem_gather_kernel (the nine shifted
    terms, bias, residual, activation, SP group).  Against fp64 and against the 224-column kernel (the default: the prototype measured slower), with BatchNorm,
    residual and every activation, ragged tiles, Cout = 193 .. 199; the pad channels of the SP row must be zero."""
    from loftr_amd import ops, _lib
    g = torch.Generator().manual_seed(17)
    lib = _lib.load()
    assert lib.loftr_conv_scratch_bytes(2, 30, 40, 196, 3, 3, 1) == 2 * 30 * 40 * 9 * 4 * 4
    assert lib.loftr_conv_scratch_bytes(2, 30, 40, 192, 3, 3, 1) == 0 and lib.loftr_conv_scratch_bytes(2, 30, 40, 200, 3, 3, 1) == 0
    assert lib.loftr_conv_scratch_bytes(2, 30, 40, 196, 3, 3, 2) == 0 and lib.loftr_conv_scratch_bytes(2, 30, 40, 196, 1, 1, 1) == 0
    for cin, cout, (H, W), act, use_res in ((196, 196, (30, 40), 1, True), (256, 196, (17, 33), 2, False), (196, 196, (9, 65), 0, True),
                                            (128, 193, (16, 32), 1, False), (64, 199, (8, 40), 2, True)):
        conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        conv.weight.data = torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * 9)) ** 0.5
        bn = nn.BatchNorm2d(cout).eval()
        bn.weight.data = 1.0 + 0.2 * torch.randn(cout, generator=g); bn.bias.data = 0.1 * torch.randn(cout, generator=g)
        bn.running_mean.data = 0.1 * torch.randn(cout, generator=g); bn.running_var.data = 0.5 + torch.rand(cout, generator=g)
        x = torch.randn(3, cin, H, W, generator=g)
        res = torch.randn(3, cout, H, W, generator=g) if use_res else None
        ref = bn.double()(F.conv2d(x.double(), conv.weight.double(), padding=1))
        bn.float()
        if res is not None:
            ref = ref + res.double()
        ref = torch.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
        x_sp = ops.sp_from_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda())
        res_sp = ops.sp_from_nhwc(res.permute(0, 2, 3, 1).contiguous().cuda()) if res is not None else None
        conv, bn = conv.cuda(), bn.cuda()
        outs = {}
        for name, rem in (("rem", True), ("wide", False)):
            ops.CONV_REM = rem                                         # (off by default: measured slower, profiles/r06_conv_rem_ab.txt)
            try:
                y_sp, _ = ops.conv_bn_act(x_sp, cin, conv, bn, act=act, residual=res_sp, want_sp=True, want_f32=False)
            finally:
                ops.CONV_REM = False
            full = ops.sp_to_nhwc(y_sp, 224)                           # all 224 stored columns, pad channels included
            assert float(full[..., cout:].abs().max()) == 0.0, (name, cin, cout)
            y = full[..., :cout].permute(0, 3, 1, 2).cpu().double()
            err = (y - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 2e-5, (name, cin, cout, err)
            outs[name] = y
        scale = ref.abs().max().item()
        assert torch.equal(outs["rem"][:, :192], outs["wide"][:, :192])              # the 192 main columns: the same arithmetic, bit for bit
        assert (outs["rem"] - outs["wide"]).abs().max().item() <= 4e-6 * scale        # the remainder channels: another summation order
