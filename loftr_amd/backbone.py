"""ResNet-FPN local feature CNN.

Architecture and parameter names follow src/loftr/backbone/resnet_fpn.py:43-199 of the reference
so that its checkpoints load with strict=True (``backbone.conv1.weight``,
``backbone.layer1.0.conv1.weight`` ...).  It *feeds* the matching path (feat_c at 1/8 or 1/16
resolution, feat_f at 1/2 or 1/4).

Two execution paths over the same parameters:
  * ``forward``      -- plain PyTorch-ROCm / MIOpen fp32 (what BASELINE.json's north_star keeps in
                        PyTorch; also the reference the HIP path is tested against);
  * ``forward_hip``  -- SURVEY.md §8(f) rank 1, the caller side of the hot path: every conv3x3 / conv1x1
                        (+ folded eval BatchNorm + residual + ReLU / LeakyReLU) is one implicit-GEMM
                        launch of the library's split-fp16 MFMA core (csrc/conv.hip), the FPN
                        upsample+add one more kernel, the 1-channel 7x7 stem a direct convolution.
``LoFTR`` uses ``forward_hip`` on the GPU in eval mode (``backbone_impl='hip'``).

Training (``.train()``, gradients wanted): ``forward`` with every convolution an autograd node on the HIP kernels
(``Conv2d`` below -> ``autograd.conv2d``: forward, input gradient, weight gradient); BatchNorm with batch statistics,
the activations, the adds and the bilinear upsampling stay PyTorch autograd.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd, ops

# A/B switch: LOFTR_FUSE_TOPDOWN=0 runs the lateral conv and the upsample-add as two launches
FUSE_TOPDOWN = os.environ.get("LOFTR_FUSE_TOPDOWN", "1") != "0"


# LOFTR_TRAIN_CONV=0: training runs the convolutions through PyTorch (MIOpen) instead of the HIP autograd node (A/B)
TRAIN_CONV_HIP = os.environ.get("LOFTR_TRAIN_CONV", "1") != "0"


class Conv2d(nn.Conv2d):
    """nn.Conv2d(bias=False) of the backbone.  In .train() mode on the GPU, with gradients wanted, it is the HIP autograd node
    (autograd.conv2d: forward, input gradient and weight gradient on the library's convolutions); everywhere else the stock
    module (the inference path does not go through forward() at all: forward_hip reads the weights)."""

    def forward(self, x):
        # (in_channels > 256: loftr_conv_wgrad's column limit -- custom block_dims above 256 train through PyTorch's convolution)
        if (TRAIN_CONV_HIP and self.training and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and self.in_channels <= 256 and (x.requires_grad or self.weight.requires_grad)):
            return autograd.conv2d(x, self.weight, self.stride[0], self.padding[0])
        return super().forward(x)


# LOFTR_TRAIN_GLUE=0: BatchNorm (batch statistics), activations, residual adds and the bilinear upsampling of a training step stay PyTorch (A/B)
TRAIN_GLUE_HIP = os.environ.get("LOFTR_TRAIN_GLUE", "1") != "0"
GLUE_PARTS = set(os.environ.get("LOFTR_TRAIN_GLUE_PARTS", "bn,act,add,up").split(","))      # (bisecting aid)


def _glue(module, x):
    """The training-mode glue runs on csrc/train_glue.hip: .train() mode, fp32 on the GPU, a graph wanted."""
    return (TRAIN_GLUE_HIP and module.training and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
            and (x.requires_grad or any(p.requires_grad for p in module.parameters(recurse=False))))


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d of the backbone.  In .train() mode on the GPU with a graph wanted: batch statistics, normalisation and their backward
    on the library (autograd.batch_norm_train), running statistics updated here exactly like torch (momentum update, unbiased variance,
    num_batches_tracked); everywhere else the stock module (inference folds the eval-mode statistics into the convolutions)."""

    def forward(self, x):
        if not (_glue(self, x) and "bn" in GLUE_PARTS and x.dim() == 4 and self.track_running_stats and self.running_mean is not None):
            return super().forward(x)
        y, mean, varu = autograd.batch_norm_train(x, self.weight, self.bias, self.eps)
        with torch.no_grad():
            self.num_batches_tracked += 1
            m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
            self.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
            self.running_var.mul_(1.0 - m).add_(varu, alpha=m)
        return y


class ReLU(nn.ReLU):
    def forward(self, x):
        return autograd.act(x, None, "relu") if _glue(self, x) and "act" in GLUE_PARTS else super().forward(x)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        # (a negative slope has no HIP node: the backward reads the derivative off the sign of the output)
        return autograd.act(x, None, "leaky_relu", self.negative_slope) if _glue(self, x) and "act" in GLUE_PARTS and self.negative_slope >= 0 else super().forward(x)


def _c1(cin, cout, stride=1):
    return Conv2d(cin, cout, kernel_size=1, stride=stride, padding=0, bias=False)


def _c3(cin, cout, stride=1):
    return Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = _c3(in_planes, planes, stride)
        self.conv2 = _c3(planes, planes)
        self.bn1 = BatchNorm2d(planes)
        self.bn2 = BatchNorm2d(planes)
        self.relu = ReLU(inplace=True)
        self.downsample = None
        if stride != 1:
            self.downsample = nn.Sequential(_c1(in_planes, planes, stride=stride), BatchNorm2d(planes))

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.downsample is not None:
            x = self.downsample(x)
        if _glue(self.relu, y) and "add" in GLUE_PARTS:
            return autograd.act(x, y, "relu")                 # relu(x + y) in one pass (resnet_fpn.py:40)
        return self.relu(x + y)


def _fuse_head(cin, cout):
    return nn.Sequential(_c3(cin, cin), BatchNorm2d(cin), LeakyReLU(), _c3(cin, cout))


class _ResNetFPN(nn.Module):
    def _stage(self, dim, stride):
        blocks = nn.Sequential(BasicBlock(self.in_planes, dim, stride=stride), BasicBlock(dim, dim, stride=1))
        self.in_planes = dim
        return blocks

    def _init(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def _up(x):
        if TRAIN_GLUE_HIP and "up" in GLUE_PARTS and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and x.requires_grad:
            return autograd.upsample2x(x)                     # training: forward and adjoint on csrc/train_glue.hip
        return F.interpolate(x, scale_factor=2., mode="bilinear", align_corners=True)

    # ---- HIP path helpers: an activation is (SP int32 tensor [B,H,W,ceil32(C)], C) -----------------
    def _stem_hip(self, x):
        """conv1 (7x7, one input channel) + bn1 + relu: direct-convolution kernel, output already SP NHWC."""
        return ops.stem_conv_bn_relu(x, self.conv1, self.bn1), self.conv1.out_channels

    @staticmethod
    def _block_hip(blk, a):
        """BasicBlock.forward (resnet_fpn.py:33-40) on an SP activation."""
        x, cin = a
        planes = blk.conv1.out_channels
        y, _ = ops.conv_bn_act(x, cin, blk.conv1, blk.bn1, act=1)
        res = x
        if blk.downsample is not None:
            res, _ = ops.conv_bn_act(x, cin, blk.downsample[0], blk.downsample[1], act=0)
        y, _ = ops.conv_bn_act(y, planes, blk.conv2, blk.bn2, act=1, residual=res)      # relu(x + y)
        return y, planes

    @classmethod
    def _stage_hip(cls, stage, a):
        for blk in stage:
            a = cls._block_hip(blk, a)
        return a

    @staticmethod
    def _topdown_hip(a, lateral_conv, low_sp, d):
        """One FPN top-down step, lateral 1x1 conv + upsampled coarser map (resnet_fpn.py:110-112): one launch
        when the map halves exactly, otherwise the lateral conv followed by the stand-alone upsample-add."""
        x, cin = a
        if FUSE_TOPDOWN and x.shape[1] == 2 * low_sp.shape[1] and x.shape[2] == 2 * low_sp.shape[2]:
            return ops.conv1x1_upsample_add(x, cin, lateral_conv, low_sp)
        lat, _ = ops.conv_bn_act(x, cin, lateral_conv)
        return ops.upsample2x_add(low_sp, lat, d)

    @staticmethod
    def _head_hip(head, a, want_f32, shared_gpu=False):
        """_fuse_head: conv3x3 + BN + LeakyReLU + conv3x3 (resnet_fpn.py:66-77).  shared_gpu: the branch runs on a
        side stream next to the coarse matching stage (ops.conv_bn_act)."""
        x, cin = a
        y, _ = ops.conv_bn_act(x, cin, head[0], head[1], act=2, shared_gpu=shared_gpu)
        return ops.conv_bn_act(y, head[0].out_channels, head[3], None, act=0, want_sp=not want_f32, want_f32=want_f32,
                               shared_gpu=shared_gpu)

    @staticmethod
    def _nchw_view(y_nhwc):
        """fp32 [B,H,W,C] -> logical [B,C,H,W] with channels-last strides (no copy)."""
        return y_nhwc.permute(0, 3, 1, 2)


class ResNetFPN_8_2(_ResNetFPN):
    """Outputs at 1/8 (coarse) and 1/2 (fine).  resnet_fpn.py:43-118."""

    def __init__(self, config):
        super().__init__()
        d0 = config["initial_dim"]
        d1, d2, d3 = config["block_dims"]
        self.in_planes = d0
        self.conv1 = Conv2d(1, d0, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(d0)
        self.relu = ReLU(inplace=True)
        self.layer1 = self._stage(d1, 1)     # 1/2
        self.layer2 = self._stage(d2, 2)     # 1/4
        self.layer3 = self._stage(d3, 2)     # 1/8
        self.layer3_outconv = _c1(d3, d3)
        self.layer2_outconv = _c1(d2, d3)
        self.layer2_outconv2 = _fuse_head(d3, d2)
        self.layer1_outconv = _c1(d1, d2)
        self.layer1_outconv2 = _fuse_head(d2, d1)
        self._init()

    def forward(self, x):
        x0 = self.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x3_out = self.layer3_outconv(x3)
        x2_out = self.layer2_outconv2(self.layer2_outconv(x2) + self._up(x3_out))
        x1_out = self.layer1_outconv2(self.layer1_outconv(x1) + self._up(x2_out))
        return [x3_out, x1_out]

    @torch.no_grad()
    def forward_hip(self, x, defer_fine=False):
        """Same function as ``forward`` (resnet_fpn.py:100-118) on the HIP implicit-GEMM convolutions.

        defer_fine=True returns ``[coarse, fine_fn]``: ``fine_fn()`` runs the FPN top-down branch (everything
        after ``layer3_outconv``) and returns the fine map.  The coarse map does not depend on it, so the caller
        can enqueue it on a second HIP stream next to the coarse matching stage (``fine_fn.reads`` lists the
        tensors it consumes, for ``Tensor.record_stream``)."""
        a0 = self._stem_hip(x)
        a1 = self._stage_hip(self.layer1, a0)       # 1/2
        a2 = self._stage_hip(self.layer2, a1)       # 1/4
        a3 = self._stage_hip(self.layer3, a2)       # 1/8
        d3 = self.layer3_outconv.out_channels
        x3_sp, x3_f32 = ops.conv_bn_act(a3[0], a3[1], self.layer3_outconv, want_f32=True)

        def fine():
            t2 = self._topdown_hip(a2, self.layer2_outconv, x3_sp, d3)
            x2_out, _ = self._head_hip(self.layer2_outconv2, (t2, d3), want_f32=False, shared_gpu=defer_fine)
            d2 = self.layer2_outconv2[3].out_channels
            t1 = self._topdown_hip(a1, self.layer1_outconv, x2_out, d2)
            _, x1_f32 = self._head_hip(self.layer1_outconv2, (t1, d2), want_f32=True, shared_gpu=defer_fine)
            return self._nchw_view(x1_f32)

        if defer_fine:
            fine.reads = (a1[0], a2[0], x3_sp)       # tensors of this stream the deferred branch reads
            return [self._nchw_view(x3_f32), fine]
        return [self._nchw_view(x3_f32), fine()]


class ResNetFPN_16_4(_ResNetFPN):
    """Outputs at 1/16 (coarse) and 1/4 (fine).  resnet_fpn.py:121-199."""

    def __init__(self, config):
        super().__init__()
        d0 = config["initial_dim"]
        d1, d2, d3, d4 = config["block_dims"]
        self.in_planes = d0
        self.conv1 = Conv2d(1, d0, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(d0)
        self.relu = ReLU(inplace=True)
        self.layer1 = self._stage(d1, 1)     # 1/2
        self.layer2 = self._stage(d2, 2)     # 1/4
        self.layer3 = self._stage(d3, 2)     # 1/8
        self.layer4 = self._stage(d4, 2)     # 1/16
        self.layer4_outconv = _c1(d4, d4)
        self.layer3_outconv = _c1(d3, d4)
        self.layer3_outconv2 = _fuse_head(d4, d3)
        self.layer2_outconv = _c1(d2, d3)
        self.layer2_outconv2 = _fuse_head(d3, d2)
        self._init()

    def forward(self, x):
        x0 = self.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x4 = self.layer4(x3)
        x4_out = self.layer4_outconv(x4)
        x3_out = self.layer3_outconv2(self.layer3_outconv(x3) + self._up(x4_out))
        x2_out = self.layer2_outconv2(self.layer2_outconv(x2) + self._up(x3_out))
        return [x4_out, x2_out]

    @torch.no_grad()
    def forward_hip(self, x, defer_fine=False):
        """Same function as ``forward`` (resnet_fpn.py:178-199) on the HIP implicit-GEMM convolutions
        (``defer_fine``: see ResNetFPN_8_2.forward_hip)."""
        a0 = self._stem_hip(x)
        a1 = self._stage_hip(self.layer1, a0)       # 1/2
        a2 = self._stage_hip(self.layer2, a1)       # 1/4
        a3 = self._stage_hip(self.layer3, a2)       # 1/8
        a4 = self._stage_hip(self.layer4, a3)       # 1/16
        d4 = self.layer4_outconv.out_channels
        x4_sp, x4_f32 = ops.conv_bn_act(a4[0], a4[1], self.layer4_outconv, want_f32=True)

        def fine():
            t3 = self._topdown_hip(a3, self.layer3_outconv, x4_sp, d4)
            x3_out, _ = self._head_hip(self.layer3_outconv2, (t3, d4), want_f32=False, shared_gpu=defer_fine)
            d3 = self.layer3_outconv2[3].out_channels
            t2 = self._topdown_hip(a2, self.layer2_outconv, x3_out, d3)
            _, x2_f32 = self._head_hip(self.layer2_outconv2, (t2, d3), want_f32=True, shared_gpu=defer_fine)
            return self._nchw_view(x2_f32)

        if defer_fine:
            fine.reads = (a2[0], a3[0], x4_sp)
            return [self._nchw_view(x4_f32), fine]
        return [self._nchw_view(x4_f32), fine()]


def build_backbone(config):
    """src/loftr/backbone/__init__.py:4-11."""
    if config["backbone_type"] == "ResNetFPN":
        if tuple(config["resolution"]) == (8, 2):
            return ResNetFPN_8_2(config["resnetfpn"])
        if tuple(config["resolution"]) == (16, 4):
            return ResNetFPN_16_4(config["resnetfpn"])
    raise ValueError(f"LOFTR.BACKBONE_TYPE {config['backbone_type']} not supported.")
