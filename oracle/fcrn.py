"""CPU fp32 oracle of the FCRN (Laina) network — TEST INFRASTRUCTURE ONLY.

Restates, in plain torch.nn fp32 ops:
  * reference network/FCRN.py:297-371  (ResNet: trunk + conv2/bn2 + UpProj + conv3
    + bilinear(align_corners=True) + sigmoid)
  * reference network/FCRN.py:31-44    (Unpool: zero insertion x2)
  * reference network/FCRN.py:167-205  (UpProj / UpProjModule)
  * reference network/FCRN.py:14-28    (weights_init)
  * the torchvision ResNet trunk the reference pulls in at FCRN.py:305
    (not in /root/reference; public "v1.5" definition: stride on the 3x3 conv).

state_dict keys equal the reference's so one dict loads into
the reference model, this oracle and the HIP module alike.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- trunk
class Bottleneck(nn.Module):
    """torchvision Bottleneck, v1.5 (stride sits on conv2). expansion 4."""
    expansion = 4

    def __init__(self, cin, width, stride=1, project=False):
        super().__init__()
        cout = width * self.expansion
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class BasicBlock(nn.Module):
    """torchvision BasicBlock (resnet18/34). expansion 1."""
    expansion = 1

    def __init__(self, cin, width, stride=1, project=False):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(width, width, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, width, 1, stride=stride, bias=False), nn.BatchNorm2d(width))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


_TRUNKS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]),
           50: (Bottleneck, [3, 4, 6, 3]), 101: (Bottleneck, [3, 4, 23, 3]),
           152: (Bottleneck, [3, 8, 36, 3])}


def _make_stage(block, cin, width, nblocks, stride):
    cout = width * block.expansion
    mods = [block(cin, width, stride, project=(stride != 1 or cin != cout))]
    mods += [block(cout, width) for _ in range(1, nblocks)]
    return nn.Sequential(*mods), cout


class ResNetTrunk(nn.Module):
    """conv1/bn1/relu/maxpool/layer1..4 with torchvision's attribute names."""

    def __init__(self, layers=50, in_channels=3):
        super().__init__()
        block, counts = _TRUNKS[layers]
        self.conv1 = nn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        c = 64
        self.layer1, c = _make_stage(block, c, 64, counts[0], 1)
        self.layer2, c = _make_stage(block, c, 128, counts[1], 2)
        self.layer3, c = _make_stage(block, c, 256, counts[2], 2)
        self.layer4, c = _make_stage(block, c, 512, counts[3], 2)
        self.out_channels = c
        # torchvision default init (pretrained=False): He fan-out normal, BN (1, 0)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)


# ----------------------------------------------------------------------------- decoder
def unpool2x(x):
    """Zero-insertion upsampling: out[..., 2i, 2j] = x[..., i, j], zeros elsewhere.
    Same result as the reference's depthwise conv_transpose2d with a one-hot 2x2
    weight (FCRN.py:39-44)."""
    n, c, h, w = x.shape
    out = x.new_zeros(n, c, 2 * h, 2 * w)
    out[:, :, ::2, ::2] = x
    return out


class UpProjModule(nn.Module):
    """FCRN.py:170-198: unpool -> {5x5-BN-ReLU-3x3-BN} + {5x5-BN} -> add -> ReLU."""

    def __init__(self, cin):
        super().__init__()
        cout = cin // 2
        self.upper_branch = nn.Sequential(OrderedDict([
            ("conv1", nn.Conv2d(cin, cout, 5, padding=2, bias=False)),
            ("batchnorm1", nn.BatchNorm2d(cout)),
            ("relu", nn.ReLU()),
            ("conv2", nn.Conv2d(cout, cout, 3, padding=1, bias=False)),
            ("batchnorm2", nn.BatchNorm2d(cout)),
        ]))
        self.bottom_branch = nn.Sequential(OrderedDict([
            ("conv", nn.Conv2d(cin, cout, 5, padding=2, bias=False)),
            ("batchnorm", nn.BatchNorm2d(cout)),
        ]))

    def forward(self, x):
        u = unpool2x(x)
        return F.relu(self.upper_branch(u) + self.bottom_branch(u))


class UpProj(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.layer1 = UpProjModule(cin)
        self.layer2 = UpProjModule(cin // 2)
        self.layer3 = UpProjModule(cin // 4)
        self.layer4 = UpProjModule(cin // 8)

    def forward(self, x):
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class _Unpool(nn.Module):
    def forward(self, x):
        return unpool2x(x)


class UpConv(nn.Module):
    """FCRN.py:91-110: four x {unpool -> 5x5 conv -> BN -> ReLU}, channels halving."""

    def __init__(self, cin):
        super().__init__()
        for i in range(4):
            c = cin // (2 ** i)
            setattr(self, "layer%d" % (i + 1), nn.Sequential(OrderedDict([
                ("unpool", _Unpool()),
                ("conv", nn.Conv2d(c, c // 2, 5, padding=2, bias=False)),
                ("batchnorm", nn.BatchNorm2d(c // 2)),
                ("relu", nn.ReLU()),
            ])))

    def forward(self, x):
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class DeConv(nn.Module):
    """FCRN.py:68-88: four x {ConvTranspose2d(k, stride 2, pad (k-1)//2, output_padding k%2) -> BN -> ReLU}."""

    def __init__(self, cin, k):
        super().__init__()
        for i in range(4):
            c = cin // (2 ** i)
            setattr(self, "layer%d" % (i + 1), nn.Sequential(OrderedDict([
                ("deconv%d" % k, nn.ConvTranspose2d(c, c // 2, k, 2, (k - 1) // 2, k % 2, bias=False)),
                ("batchnorm", nn.BatchNorm2d(c // 2)),
                ("relu", nn.ReLU(inplace=True)),
            ])))

    def forward(self, x):
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class FasterUpconv(nn.Module):
    """FCRN.py:209-246: four biased convs on differently zero-padded copies of x, BN each, cat, PixelShuffle(2)."""

    def __init__(self, cin, with_relu=False):
        super().__init__()
        for name, ks in (("conv1_", 3), ("conv2_", (2, 3)), ("conv3_", (3, 2)), ("conv4_", 2)):
            setattr(self, name, nn.Sequential(OrderedDict([("conv1", nn.Conv2d(cin, cin // 2, ks)),
                                                           ("bn1", nn.BatchNorm2d(cin // 2))])))
        self.ps = nn.PixelShuffle(2)
        self.relu = nn.ReLU(inplace=True)
        self.with_relu = with_relu       # FasterUpConv's module applies it (FCRN.py:160), FasterUpProj's does not (:245)

    def forward(self, x):
        xs = [self.conv1_(F.pad(x, (1, 1, 1, 1))), self.conv2_(F.pad(x, (1, 1, 0, 1))),
              self.conv3_(F.pad(x, (0, 1, 1, 1))), self.conv4_(F.pad(x, (0, 1, 0, 1)))]
        y = self.ps(torch.cat(xs, 1))
        return self.relu(y) if self.with_relu else y


class FasterUpProjModule(nn.Module):
    """FCRN.py:248-269."""

    def __init__(self, cin):
        super().__init__()
        c = cin // 2
        self.upper_branch = nn.Sequential(OrderedDict([
            ("faster_upconv", FasterUpconv(cin)), ("relu", nn.ReLU(inplace=True)),
            ("conv", nn.Conv2d(c, c, 3, padding=1, bias=False)), ("batchnorm", nn.BatchNorm2d(c))]))
        self.bottom_branch = FasterUpconv(cin)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.relu(self.upper_branch(x) + self.bottom_branch(x))


class FasterUpProj(nn.Module):
    def __init__(self, cin):
        super().__init__()
        for i in range(4):
            setattr(self, "layer%d" % (i + 1), FasterUpProjModule(cin // (2 ** i)))

    def forward(self, x):
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class FasterUpConv(nn.Module):
    """FCRN.py:113-164: four x faster_upconv_module (the four convs + BN, shuffle, ReLU)."""

    def __init__(self, cin):
        super().__init__()
        for i in range(4):
            setattr(self, "layer%d" % (i + 1), FasterUpconv(cin // (2 ** i), with_relu=True))

    def forward(self, x):
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def make_decoder(decoder, cin):
    """FCRN.py:282-294 (+ 'fasterupconv' for the class the reference defines at :113 but never selects)."""
    if decoder == "fasterupconv":
        return FasterUpConv(cin)
    if decoder == "fasterupproj":
        return FasterUpProj(cin)
    if decoder[:6] == "deconv":
        return DeConv(cin, int(decoder[6]))
    if decoder == "upproj":
        return UpProj(cin)
    if decoder == "upconv":
        return UpConv(cin)
    raise ValueError(decoder)


def he_init_(m):
    """FCRN.py:14-28: N(0, sqrt(2 / (kh*kw*Cout))) for convs (ConvTranspose: Cin), BN -> (1, 0)."""
    if isinstance(m, nn.ConvTranspose2d):
        fan = m.kernel_size[0] * m.kernel_size[1] * m.in_channels
        m.weight.data.normal_(0, math.sqrt(2.0 / fan))
    elif isinstance(m, nn.Conv2d):
        fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
        m.weight.data.normal_(0, math.sqrt(2.0 / fan))
        if m.bias is not None:
            m.bias.data.zero_()
    elif isinstance(m, nn.BatchNorm2d):
        m.weight.data.fill_(1)
        m.bias.data.zero_()


class FCRNOracle(nn.Module):
    """FCRN.ResNet restated (decoders upproj / upconv / deconvK / fasterupproj). Returns sigmoid(depth map)."""

    def __init__(self, layers=50, output_size=(228, 304), in_channels=3, out_channels=20, decoder="upproj"):
        super().__init__()
        t = ResNetTrunk(layers, in_channels)
        for name in ("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4"):
            setattr(self, name, getattr(t, name))
        nch = t.out_channels
        self.output_size = tuple(output_size)
        self.conv2 = nn.Conv2d(nch, nch // 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(nch // 2)
        self.upSample = make_decoder(decoder, nch // 2)
        self.conv3 = nn.Conv2d(nch // 32, out_channels, 3, padding=1, bias=False)
        for m in (self.conv2, self.bn2, self.upSample, self.conv3):
            m.apply(he_init_)

    def features(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def forward(self, x):
        x = self.bn2(self.conv2(self.features(x)))
        x = self.conv3(self.upSample(x))
        x = F.interpolate(x, size=self.output_size, mode="bilinear", align_corners=True)
        return torch.sigmoid(x)

    def get_1x_lr_params(self):
        for m in (self.conv1, self.bn1, self.layer1, self.layer2, self.layer3, self.layer4):
            yield from (p for p in m.parameters() if p.requires_grad)

    def get_10x_lr_params(self):
        for m in (self.conv2, self.bn2, self.upSample, self.conv3):
            yield from (p for p in m.parameters() if p.requires_grad)
