"""Stand-ins for the encoder trunks the reference pulls from torchvision / torch.hub — TEST INFRASTRUCTURE ONLY.

torchvision is neither under /root/reference nor in this image, and the reference pins no version of it; the
hub repo facebookresearch/WSL-Images (MiDaS.py:110) is architecturally torchvision's resnext101_32x8d.  These classes
restate the PUBLIC architecture definitions (module names = torchvision's, so state_dict keys match) with plain
torch.nn layers; tests/golden/gen_golden.py hands them to the reference's own decoder / head code in place of the
download.  Their arithmetic is torch.nn's; what is pinned by the goldens is everything the reference itself defines.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5: stride on the 3x3), with groups / width_per_group (ResNeXt)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


class ResNeXt(nn.Module):
    """torchvision.models.resnet.ResNet with Bottleneck blocks, groups and width_per_group (no avgpool / fc use)."""

    def __init__(self, layers, groups, width_per_group):
        super().__init__()
        self.inplanes, self.groups, self.base_width = 64, groups, width_per_group
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        mods = [_Bottleneck(self.inplanes, planes, stride, ds, self.groups, self.base_width)]
        self.inplanes = planes * 4
        mods += [_Bottleneck(self.inplanes, planes, groups=self.groups, base_width=self.base_width) for _ in range(1, blocks)]
        return nn.Sequential(*mods)


class ResNetFull(ResNeXt):
    """The WHOLE torchvision ResNet, avgpool and fc included: what `models.resnet50(pretrained=True)` returns and Bts.py:293-307
    keeps as `encoder.base_model` (its forward walk skips 'avgpool' and 'fc', Bts.py:313-315, but their parameters stay in
    the state_dict)."""

    def __init__(self, layers, groups=1, width_per_group=64):
        super().__init__(layers, groups, width_per_group)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)


def resnet50_full(pretrained=False):
    return ResNetFull([3, 4, 6, 3])


def resnet101_full(pretrained=False):
    return ResNetFull([3, 4, 23, 3])


def resnext50_32x4d_full(pretrained=False):
    return ResNetFull([3, 4, 6, 3], 32, 4)


def resnext101_32x8d_full(pretrained=False):
    return ResNetFull([3, 4, 23, 3], 32, 8)


def resnext101_32x8d():
    return ResNeXt([3, 4, 23, 3], 32, 8)


def resnext50_32x4d():
    return ResNeXt([3, 4, 6, 3], 32, 4)


# ---------------------------------------------------------------------------------------------- DenseNet-161 (Bts.py:283-292)
class _DenseLayer(nn.Module):
    """torchvision.models.densenet._DenseLayer: BN -> ReLU -> 1x1 (bn_size * growth) -> BN -> ReLU -> 3x3 (growth) on the
    concatenation of everything the block has produced so far."""

    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1)
        return self.conv2(self.relu2(self.norm2(self.conv1(self.relu1(self.norm1(x))))))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, n, cin, growth, bn_size):
        super().__init__()
        for i in range(n):
            self.add_module("denselayer%d" % (i + 1), _DenseLayer(cin + i * growth, growth, bn_size))

    def forward(self, x):
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class DenseNet(nn.Module):
    """torchvision.models.densenet.DenseNet `.features` (the classifier is never used by Bts.py)."""

    def __init__(self, growth=48, blocks=(6, 12, 36, 24), init=96, bn_size=4):
        super().__init__()
        feats = [("conv0", nn.Conv2d(3, init, 7, 2, 3, bias=False)), ("norm0", nn.BatchNorm2d(init)), ("relu0", nn.ReLU(inplace=True)),
                 ("pool0", nn.MaxPool2d(3, 2, 1))]
        c = init
        for i, n in enumerate(blocks):
            feats.append(("denseblock%d" % (i + 1), _DenseBlock(n, c, growth, bn_size)))
            c += n * growth
            if i != len(blocks) - 1:
                feats.append(("transition%d" % (i + 1), nn.Sequential(OrderedDict([
                    ("norm", nn.BatchNorm2d(c)), ("relu", nn.ReLU(inplace=True)), ("conv", nn.Conv2d(c, c // 2, 1, bias=False)),
                    ("pool", nn.AvgPool2d(2, 2))]))))
                c //= 2
        feats.append(("norm5", nn.BatchNorm2d(c)))
        self.features = nn.Sequential(OrderedDict(feats))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)


def densenet161(pretrained=False):
    return DenseNet(48, (6, 12, 36, 24), 96)


# ---------------------------------------------------------------------------------------------- VGG-19 (BN) features (Eigen.py:74)
class VGG19BN(nn.Module):
    """torchvision.models.vgg19_bn(...)._modules['features']: configuration E with BatchNorm (conv3x3 -> BN -> ReLU, 'M' =
    MaxPool2d(2, 2)); the classifier is never used by Eigen.py."""
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]

    def __init__(self):
        super().__init__()
        layers, c = [], 3
        for v in self.cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)


def vgg19_bn(pretrained=False):
    return VGG19BN()
