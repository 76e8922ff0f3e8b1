"""ResNet-FPN local feature CNN -- stays in PyTorch-ROCm (MIOpen), per BASELINE.json north_star.

Architecture and parameter names follow src/loftr/backbone/resnet_fpn.py:43-199 of the reference
so that its checkpoints load with strict=True (``backbone.conv1.weight``,
``backbone.layer1.0.conv1.weight`` ...).  Out of the hand-written hot path; it only *feeds* it
(feat_c at 1/8 or 1/16 resolution, feat_f at 1/2 or 1/4).
"""
import torch.nn as nn
import torch.nn.functional as F


def _c1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=1, stride=stride, padding=0, bias=False)


def _c3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = _c3(in_planes, planes, stride)
        self.conv2 = _c3(planes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1:
            self.downsample = nn.Sequential(_c1(in_planes, planes, stride=stride), nn.BatchNorm2d(planes))

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


def _fuse_head(cin, cout):
    return nn.Sequential(_c3(cin, cin), nn.BatchNorm2d(cin), nn.LeakyReLU(), _c3(cin, cout))


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
        return F.interpolate(x, scale_factor=2., mode="bilinear", align_corners=True)


class ResNetFPN_8_2(_ResNetFPN):
    """Outputs at 1/8 (coarse) and 1/2 (fine).  resnet_fpn.py:43-118."""

    def __init__(self, config):
        super().__init__()
        d0 = config["initial_dim"]
        d1, d2, d3 = config["block_dims"]
        self.in_planes = d0
        self.conv1 = nn.Conv2d(1, d0, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(d0)
        self.relu = nn.ReLU(inplace=True)
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


class ResNetFPN_16_4(_ResNetFPN):
    """Outputs at 1/16 (coarse) and 1/4 (fine).  resnet_fpn.py:121-199."""

    def __init__(self, config):
        super().__init__()
        d0 = config["initial_dim"]
        d1, d2, d3, d4 = config["block_dims"]
        self.in_planes = d0
        self.conv1 = nn.Conv2d(1, d0, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(d0)
        self.relu = nn.ReLU(inplace=True)
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


def build_backbone(config):
    """src/loftr/backbone/__init__.py:4-11."""
    if config["backbone_type"] == "ResNetFPN":
        if tuple(config["resolution"]) == (8, 2):
            return ResNetFPN_8_2(config["resnetfpn"])
        if tuple(config["resolution"]) == (16, 4):
            return ResNetFPN_16_4(config["resnetfpn"])
    raise ValueError(f"LOFTR.BACKBONE_TYPE {config['backbone_type']} not supported.")
