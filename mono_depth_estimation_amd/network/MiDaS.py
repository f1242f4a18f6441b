"""Drop-in for reference network/MiDaS.py on MI355X: `MidasNet(path=None, features=256, non_negative=True)` with the
`.pretrained` / `.scratch` attributes modules/midas.py:44,96-97 uses, identical state_dict keys, `BaseModel.load`, and a
forward that returns the N x 7 x H x W sigmoid maps of MiDaS.py:49-57 — computed by hand-written gfx950 kernels
(mono_depth_estimation_amd/graph.py tape).  The submodules only hold parameters.

Network (MiDaS.py:59-87,114-229): ResNeXt-101 32x8d trunk (the reference fetches it from torch.hub
"facebookresearch/WSL-Images" — never executed here, the architecture is torchvision's resnext101_32x8d, built in place),
four 3x3 "reassemble" convs to `features` channels, four FeatureFusionBlocks (ResidualConvUnits with biased 3x3 convs
and IN-PLACE ReLUs, bilinear x2 with align_corners=True), and the output head 3x3 -> bilinear x2 (align_corners=False)
-> 3x3 -> ReLU -> 1x1 (7 channels) -> sigmoid.
"""
import torch
import torch.nn as nn

from .. import graph as G


class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("%s is a parameter container of the HIP MiDaS path; call the MidasNet instead" % type(self).__name__)


class _Seq(nn.Sequential):
    def forward(self, *a, **k):
        raise RuntimeError("this Sequential is a parameter container of the HIP MiDaS path; call the MidasNet instead")


class Bottleneck(_Container):
    """torchvision Bottleneck (v1.5) with groups / width_per_group; names as torchvision's."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, project=False, groups=32, base_width=8):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = _Seq(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))


def _stage(inplanes, planes, blocks, stride, groups, base_width):
    mods = [Bottleneck(inplanes, planes, stride, True, groups, base_width)]
    mods += [Bottleneck(planes * 4, planes, 1, False, groups, base_width) for _ in range(1, blocks)]
    return _Seq(*mods)


def _make_pretrained_resnext101_wsl(use_pretrained, blocks=(3, 4, 23, 3), groups=32, base_width=8):
    """MiDaS.py:93-111: `pretrained.layer1 = Sequential(conv1, bn1, relu, maxpool, resnet.layer1)`, layer2..4 the stages.
    use_pretrained: the reference downloads the WSL weights through torch.hub; there is no network here — load a
    MiDaS checkpoint with `MidasNet.load(path)` instead."""
    pretrained = _Container()
    pretrained.layer1 = _Seq(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1),
                             _stage(64, 64, blocks[0], 1, groups, base_width))
    pretrained.layer2 = _stage(256, 128, blocks[1], 2, groups, base_width)
    pretrained.layer3 = _stage(512, 256, blocks[2], 2, groups, base_width)
    pretrained.layer4 = _stage(1024, 512, blocks[3], 2, groups, base_width)
    for m in pretrained.modules():                   # torchvision's default initialisation of the trunk
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return pretrained


def _make_scratch(in_shape, out_shape):
    scratch = _Container()
    for i, c in enumerate(in_shape):
        setattr(scratch, "layer%d_rn" % (i + 1), nn.Conv2d(c, out_shape, kernel_size=3, stride=1, padding=1, bias=False))
    return scratch


def _make_encoder(features, use_pretrained):
    return _make_pretrained_resnext101_wsl(use_pretrained), _make_scratch([256, 512, 1024, 2048], features)


class Interpolate(_Container):
    def __init__(self, scale_factor, mode):
        super().__init__()
        self.scale_factor, self.mode = scale_factor, mode


class ResidualConvUnit(_Container):
    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.relu = nn.ReLU(inplace=True)


class FeatureFusionBlock(_Container):
    def __init__(self, features):
        super().__init__()
        self.resConfUnit1 = ResidualConvUnit(features)
        self.resConfUnit2 = ResidualConvUnit(features)


class MidasEngine(G.TapeEngine):
    """The tape of MidasNet.forward (MiDaS.py:59-87)."""

    def _rcu(self, a, rcu, r2=None):
        """ResidualConvUnit on a = relu(x) (the unit's ReLU is in place, MiDaS.py:196: the skip is relu(x))."""
        c1 = self.add(G.Conv(self, a, rcu.conv1.weight, 3, 1, 1)).out
        h = self.pw(c1, bias=rcu.conv1.bias, act="relu")
        c2 = self.add(G.Conv(self, h, rcu.conv2.weight, 3, 1, 1)).out
        return self.pw(c2, bias=rcu.conv2.bias, r=a)

    def _ffb(self, ffb, x0, x1=None):
        if x1 is not None:
            t = self._rcu(self.pw(x1, act="relu"), ffb.resConfUnit1)
            a = self.pw(x0, r=t, act="relu")          # relu(x0 + rcu1(x1)), the in-place ReLU of unit 2
        else:
            a = self.pw(x0, act="relu")
        y = self._rcu(a, ffb.resConfUnit2)
        return self.add(G.Resize(self, y, 2 * y.H, 2 * y.W, True)).out

    def _plan(self):
        m, N, H, W = self.m, self.N, self.H, self.W
        if H % 32 or W % 32:
            raise ValueError("MidasNet: image sizes must be multiples of 32 (got %d x %d): the fusion blocks double each map" % (H, W))
        pre, sc = m.pretrained, m.scratch
        l1 = pre.layer1
        self.stem = self.add(G.Stem(self, l1[0], l1[1], N, H, W))
        x, feats = self.stem.out, []
        for stage in (l1[4], pre.layer2, pre.layer3, pre.layer4):
            for blk in stage:
                ds = blk.downsample
                x = self.bottleneck(x, blk.conv1, blk.bn1, blk.conv2, blk.bn2, blk.conv3, blk.bn3,
                                    ds[0] if ds is not None else None, ds[1] if ds is not None else None)
            feats.append(x)
        rn = [self.add(G.Conv(self, f, getattr(sc, "layer%d_rn" % (i + 1)).weight, 3, 1, 1)).out for i, f in enumerate(feats)]
        p = self._ffb(sc.refinenet4, rn[3])
        p = self._ffb(sc.refinenet3, p, rn[2])
        p = self._ffb(sc.refinenet2, p, rn[1])
        p = self._ffb(sc.refinenet1, p, rn[0])
        oc = sc.output_conv
        c = self.add(G.Conv(self, p, oc[0].weight, 3, 1, 1)).out
        c = self.pw(c, bias=oc[0].bias)                # (the bias before the resize: padding of the next conv sees it)
        u = self.add(G.Resize(self, c, 2 * c.H, 2 * c.W, False)).out
        c = self.add(G.Conv(self, u, oc[2].weight, 3, 1, 1)).out
        h = self.pw(c, bias=oc[2].bias, act="relu")
        c = self.add(G.Conv(self, h, oc[4].weight, 1)).out
        self.heads = [self.add(G.ToNCHW(self, c, oc[4].bias, oc[4].out_channels, "sigmoid"))]


class BaseModel(G.TapeModule):
    def load(self, path):
        """MiDaS.py:10-23: a state_dict file, or a training checkpoint {"optimizer": ..., "model": state_dict}."""
        parameters = torch.load(path, map_location=torch.device('cpu'))
        if "optimizer" in parameters:
            parameters = parameters["model"]
        self.load_state_dict(parameters)


class MidasNet(BaseModel):
    _engine_cls = MidasEngine

    def __init__(self, path=None, features=256, non_negative=True):
        print("Loading weights: ", path)
        super(MidasNet, self).__init__()
        use_pretrained = False if path is None else True
        self.pretrained, self.scratch = _make_encoder(features, use_pretrained)
        self.scratch.refinenet4 = FeatureFusionBlock(features)
        self.scratch.refinenet3 = FeatureFusionBlock(features)
        self.scratch.refinenet2 = FeatureFusionBlock(features)
        self.scratch.refinenet1 = FeatureFusionBlock(features)
        self.scratch.output_conv = _Seq(
            nn.Conv2d(features, 128, kernel_size=3, stride=1, padding=1),
            Interpolate(scale_factor=2, mode="bilinear"),
            nn.Conv2d(128, 32, kernel_size=3, stride=1, padding=1),
            nn.ReLU(True),
            nn.Conv2d(32, 7, kernel_size=1, stride=1, padding=0),
            nn.Sigmoid()
        )
        self._init_runtime()
        if path:
            self.load(path)

    def _make_store(self, device):
        raw = [n for n, p in self.named_parameters() if p.dim() == 4 and (n == "pretrained.layer1.0.weight" or (n.startswith("pretrained.") and n.endswith(".conv2.weight")))]
        return G.NetStore(self, device, is_encoder=lambda n: n.startswith("pretrained."), raw=raw)   # midas.py:96-97: 0.1 x LR

    def forward(self, x):
        return self._run(x)[0]
