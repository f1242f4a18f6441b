"""Drop-in for reference network/Dorn.py on MI355X: `DORN(args)` with the attribute paths modules/dorn.py walks
(`.backbone` — frozen by `freeze_encoder`, the 1x learning-rate group — and `.SceneUnderstandingModule`, the 10x group),
identical state_dict keys and construction order (so the same seed gives the same initial weights), and a forward that
returns `(decode_c, ord_c1)`: the int64 ordinal label map N x 1 x H x W and the fp32 ordinal probabilities N x K x H x W —
computed by hand-written gfx950 kernels through libmde_hip.so (mono_depth_estimation_amd/graph.py tape).  The submodules
below only hold parameters.

Network (Dorn.py:124-353): a dilated ResNet-101 at output stride 8 (three 3x3 stem convs, ceil-mode max-pool, layer3 / layer4
dilated by 2 / 4), the scene understanding module — full-image encoder (padded average pool, Dropout2d, nn.Linear, 1x1 conv,
broadcast) beside a 1x1 and three dilated 3x3 ASPP branches over 2048 channels, concatenated, Dropout2d, two 1x1 convs —
a bilinear (align_corners) resize to `args.input_size`, and the ordinal regression layer.
"""
from pathlib import Path

import torch
import torch.nn as nn
from torch.nn import BatchNorm2d

from .. import graph as G


class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("%s is a parameter container of the HIP DORN path; call the DORN module instead" % type(self).__name__)


class _Seq(nn.Sequential):
    def forward(self, *a, **k):
        raise RuntimeError("this Sequential is a parameter container of the HIP DORN path; call the DORN module instead")


def consistent_padding_with_dilation(padding, dilation, dim=2):
    """Dorn.py:20-35: a dilated conv pads by its dilation."""
    assert dim == 2, "2-D convolutions only"
    padding = padding if isinstance(padding, (tuple, list)) else (padding, padding)
    dilation = dilation if isinstance(dilation, (tuple, list)) else (dilation, dilation)
    return tuple(d if d > 1 else p for p, d in zip(padding, dilation)), tuple(dilation)


def conv_bn_relu(batchNorm, in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1, bias=True):
    """Dorn.py:38-54."""
    padding, dilation = consistent_padding_with_dilation(padding, dilation, dim=2)
    if batchNorm:
        return _Seq(nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation, bias=False),
                    nn.BatchNorm2d(out_planes), nn.ReLU(inplace=True))
    return _Seq(nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation, bias=bias),
                nn.ReLU(inplace=True))


class FullImageEncoder(_Container):
    """Dorn.py:57-80."""

    def __init__(self, h, w, kernel_size, dropout_prob=0.5):
        super().__init__()
        self.global_pooling = nn.AvgPool2d(kernel_size, stride=kernel_size, padding=kernel_size // 2)
        self.dropout = nn.Dropout2d(p=dropout_prob)
        self.h = h // kernel_size + 1
        self.w = w // kernel_size + 1
        self.global_fc = nn.Linear(2048 * self.h * self.w, 512)
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(512, 512, 1)


class SceneUnderstandingModule(_Container):
    """Dorn.py:83-124."""

    def __init__(self, ord_num, size, kernel_size, pyramid=[6, 12, 18], dropout_prob=0.5, batch_norm=False):
        super().__init__()
        assert len(size) == 2
        assert len(pyramid) == 3
        self.size = size
        h, w = self.size
        self.encoder = FullImageEncoder(h // 8, w // 8, kernel_size, dropout_prob)
        self.aspp1 = _Seq(conv_bn_relu(batch_norm, 2048, 512, kernel_size=1, padding=0),
                          conv_bn_relu(batch_norm, 512, 512, kernel_size=1, padding=0))
        for i in range(3):
            setattr(self, "aspp%d" % (i + 2),
                    _Seq(conv_bn_relu(batch_norm, 2048, 512, kernel_size=3, padding=pyramid[i], dilation=pyramid[i]),
                         conv_bn_relu(batch_norm, 512, 512, kernel_size=1, padding=0)))
        self.concat_process = _Seq(nn.Dropout2d(p=dropout_prob),
                                   conv_bn_relu(batch_norm, 512 * 5, 2048, kernel_size=1, padding=0),
                                   nn.Dropout2d(p=dropout_prob),
                                   nn.Conv2d(2048, int(ord_num * 2), 1))


affine_par = True


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class Bottleneck(_Container):
    """Dorn.py:135-175: stride and dilation on the 3x3."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, fist_dilation=1, multi_grid=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation * multi_grid, dilation=dilation * multi_grid,
                               bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.relu_inplace = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.dilation = dilation
        self.stride = stride


class ResNet(_Container):
    """Dorn.py:221-275."""

    def __init__(self, block, layers):
        self.inplanes = 128
        super().__init__()
        self.conv1 = conv3x3(3, 64, stride=2)
        self.bn1 = BatchNorm2d(64)
        self.relu1 = nn.ReLU(inplace=False)
        self.conv2 = conv3x3(64, 64)
        self.bn2 = BatchNorm2d(64)
        self.relu2 = nn.ReLU(inplace=False)
        self.conv3 = conv3x3(64, 128)
        self.bn3 = BatchNorm2d(128)
        self.relu3 = nn.ReLU(inplace=False)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=True)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4, multi_grid=(1, 1, 1))

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, multi_grid=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = _Seq(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                              BatchNorm2d(planes * block.expansion, affine=affine_par))
        grid = lambda index, grids: grids[index % len(grids)] if isinstance(grids, tuple) else 1
        layers = [block(self.inplanes, planes, stride, dilation=dilation, downsample=downsample, multi_grid=grid(0, multi_grid))]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=dilation, multi_grid=grid(i, multi_grid)))
        return _Seq(*layers)

    def freeze(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()


WEIGHTS_FILE = './network/pretrained_models/resnet101-imagenet.pth'


class ResNetBackbone(_Container):
    """Dorn.py:196-219.  `pretrained`: the reference downloads resnet101-imagenet.pth when the file is missing; there is no
    network here, so the file must already be in place (same path, same key filter: everything but `fc.*`)."""

    def __init__(self, pretrained=True):
        super().__init__()
        self.backbone = ResNet(Bottleneck, [3, 4, 23, 3])
        if pretrained:
            weights_file = Path(WEIGHTS_FILE).resolve()
            if not weights_file.exists():
                raise FileNotFoundError("DORN(pretrained=1) needs %s (the reference fetches it from sceneparsing.csail.mit.edu; this build "
                                        "does not download): put the file there or pass pretrained=0" % weights_file)
            saved = torch.load(weights_file.as_posix(), map_location='cpu')
            new_params = self.backbone.state_dict().copy()
            for k in saved:
                if not k.split('.')[0] == 'fc':
                    new_params[k] = saved[k]
            self.backbone.load_state_dict(new_params)


class OrdinalRegressionLayer(_Container):
    """Dorn.py:278-318 (no parameters; the HIP plan runs mde_ordinal_fwd / _bwd)."""


# ---------------------------------------------------------------------------------------------- launch plan
class DORNEngine(G.TapeEngine):
    """The tape of DORN.forward (Dorn.py:340-344)."""

    def _cbr(self, x, seq, out=None):
        """conv_bn_relu (Dorn.py:38-54): conv -> BN -> ReLU, or biased conv -> ReLU."""
        conv = seq[0]
        if isinstance(seq[1], nn.BatchNorm2d):
            return self.conv_bn(x, conv, seq[1], True, out=out)
        k, s, p, d = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
        c = self.add(G.Conv(self, x, conv.weight, k, s, p, d)).out
        return self.pw(c, bias=conv.bias, act="relu", out=out)

    def _plan(self):
        m, N, H, W = self.m, self.N, self.H, self.W
        net, su = m.backbone.backbone, m.SceneUnderstandingModule
        s1 = self._site([net.bn1])
        self.stem = self.add(G.ImageStem(self, net.conv1, s1, N, H, W))
        x = self.add(G.BN(self, self.stem.out, s1, True)).out
        x = self.conv_bn(x, net.conv2, net.bn2, True)
        x = self.conv_bn(x, net.conv3, net.bn3, True)
        x = self.add(G.MaxPool(self, x, ceil_mode=True)).out
        for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
            for blk in layer:
                ds = blk.downsample
                x = self.bottleneck(x, blk.conv1, blk.bn1, blk.conv2, blk.bn2, blk.conv3, blk.bn3,
                                    ds[0] if ds is not None else None, ds[1] if ds is not None else None)
        # scene understanding module (Dorn.py:110-124): five 512-channel branches side by side in one tensor
        enc = su.encoder
        self.dropouts = []
        pooled = self.add(G.PooledFlat(self, x, enc.global_pooling.kernel_size, enc.dropout.p))
        if (pooled.oh, pooled.ow) != (enc.h, enc.w):
            raise ValueError("DORN: a %dx%d image gives a %dx%d pooled map, the full-image encoder was built for %dx%d (input_size %s)"
                             % (H, W, pooled.oh, pooled.ow, enc.h, enc.w, tuple(su.size)))
        self.dropouts.append(pooled)
        cat = self.buf(N, x.H, x.W, 5 * 512)
        f = self.add(G.Conv(self, pooled.out, enc.global_fc.weight, 1)).out
        f = self.pw(f, bias=enc.global_fc.bias, act="relu")
        f = self.add(G.Conv(self, f, enc.conv1.weight, 1)).out
        f = self.pw(f, bias=enc.conv1.bias)
        self.add(G.Broadcast(self, f, cat.slice(0, 512)))             # bilinear(align_corners) of a 1x1 map
        for i in range(4):
            aspp = getattr(su, "aspp%d" % (i + 1))
            self._cbr(self._cbr(x, aspp[0]), aspp[1], out=cat.slice(512 * (i + 1), 512))
        cp = su.concat_process
        d = self.add(G.ChannelDropout(self, cat, cp[0].p))
        self.dropouts.append(d)
        t = self._cbr(d.out, cp[1])
        d = self.add(G.ChannelDropout(self, t, cp[2].p))
        self.dropouts.append(d)
        c = self.add(G.Conv(self, d.out, cp[3].weight, 1)).out
        c = self.pw(c, bias=cp[3].bias)
        K = cp[3].out_channels // 2
        up = self.add(G.Resize(self, c, int(su.size[0]), int(su.size[1]), True)).out     # (c.C = 2K rounded up to 8, the padding is zero)
        self.heads = [self.add(G.OrdinalHead(self, up, K))]


class DORN(G.TapeModule):
    """reference Dorn.py:321-344."""

    _engine_cls = DORNEngine

    def __init__(self, args):
        self.args = args
        super().__init__()
        assert len(self.args.input_size) == 2
        assert isinstance(self.args.kernel_size, int)
        self.ord_num = self.args.ord_num
        self.alpha = self.args.alpha
        self.beta = self.args.beta
        self.discretization = self.args.discretization
        self.backbone = ResNetBackbone(pretrained=self.args.pretrained)
        self.SceneUnderstandingModule = SceneUnderstandingModule(self.ord_num, size=self.args.input_size,
                                                                 kernel_size=self.args.kernel_size,
                                                                 pyramid=self.args.pyramid,
                                                                 batch_norm=self.args.batch_norm,
                                                                 dropout_prob=self.args.dropout)
        self.regression_layer = OrdinalRegressionLayer()
        self._init_runtime()

    def _make_store(self, device):
        return G.NetStore(self, device, is_encoder=lambda n: n.startswith("backbone."))    # dorn.py:188-191: backbone 1x, the rest 10x

    def forward(self, image):
        prob, label = self._run(image)
        return prob, label
