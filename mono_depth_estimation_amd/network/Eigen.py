"""Drop-in for reference network/Eigen.py on MI355X: `Eigen(scale1='vgg', pretrained=...)` with `.scale1` / `.scale2` / `.scale3`
(the three parameter groups of modules/eigen.py:55-60), identical state_dict keys (138) and construction order, and a forward
that returns the fp32 N x 1 x 109 x 149 map of a 240 x 320 image -- computed by hand-written gfx950 kernels through
libmde_hip.so (graph.py tape).  The submodules below only hold parameters.

Network (Eigen.py:5-90):
  scale 1  VGG-19-BN `features` (torchvision's configuration 'E' with BatchNorm: sixteen 3x3 convs with bias -> BN -> ReLU, five
           MaxPool2d(2, 2); absent from the image and from /root/reference, restated from its public definition) -> flatten in
           NCHW order -> Linear(512 * 10 * 7, 4096) -> Linear(4096, 64 * 19 * 14) (no activation between them) -> reshape
           (64, 14, 19) -> ConvTranspose2d(64, 64, 3, stride 4): 55 x 75;
  scale 2  Conv2d(3, 96, 9, stride 2) on the image -> ReLU -> MaxPool2d(3, 2) cropped by one row / column on every side, cat with
           scale 1 (160 channels) -> three 5x5 convs + ReLU -> ConvTranspose2d(64, 1, 5, stride 2, padding 2): 109 x 149;
  scale 3  Conv2d(3, 96, 9, stride 2) on the image cropped [2:-3] -> ReLU -> MaxPool2d(3, 1), cat with scale 2 (97 channels) ->
           three 5x5 convs + ReLU -> Conv2d(64, 1, 5) + ReLU.
The two Linear layers fix the input at 240 x 320 (Eigen.py:77-78; the reference itself rejects anything else).

How it maps to the kernels: the image convs are tap launches of the GEMM kernel over the image converted ONCE to NHWC bf16
(shared by the three of them); nn.Linear is a 1x1 GEMM over the flattened row, its bias in the GEMM's epilogue; the reshape is a
layout kernel; a transposed conv is the input gradient of the strided conv with the same weights (graph.ConvT: stride 4 =
sixteen output phases, seven of them bias only); the cropped pools are one max-pool kernel over a spatial view
(graph.MaxPoolView); concatenations are never executed (producers write channel slices); the 97-channel concatenation and the
one-channel maps are stored padded to a multiple of 8 channels, the padding provably zero.
`pretrained=True` would download torchvision weights (Eigen.py:74): there is no network here, it raises.
"""
import torch.nn as nn

from .. import graph as G

_VGG19 = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("%s is a parameter container of the HIP Eigen path; call the Eigen model instead" % type(self).__name__)


class _Seq(nn.Sequential):
    def forward(self, *a, **k):
        raise RuntimeError("this Sequential is a parameter container of the HIP Eigen path; call the Eigen model instead")


def _vgg19_bn_features():
    """torchvision.models.vgg19_bn().features: Conv2d(3x3, padding 1, bias) -> BatchNorm2d -> ReLU(inplace) per entry, MaxPool2d(2, 2)
    per 'M' (module indices as in torchvision's make_layers, so the state_dict keys are `feature_extractor.<i>.*`)."""
    layers, c = [], 3
    for v in _VGG19:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            c = v
    return _Seq(*layers)


class VGG(_Container):
    """Eigen.py:71-89."""

    def __init__(self, pretrained=True):
        super(VGG, self).__init__()
        if pretrained:
            raise NotImplementedError("Eigen(pretrained=True) downloads torchvision's vgg19_bn weights (Eigen.py:74): no network "
                                      "here; build with pretrained=False and load a state_dict")
        self.feature_extractor = _vgg19_bn_features()
        self.flatten = nn.Flatten()
        self.mlp1 = nn.Linear(512 * 10 * 7, 4096)
        self.mlp2 = nn.Linear(4096, 64 * 19 * 14)
        self.upsample = nn.ConvTranspose2d(64, 64, kernel_size=3, stride=4)


class Scale2(_Container):
    """Eigen.py:20-44."""

    def __init__(self):
        super(Scale2, self).__init__()
        self.conv = nn.Conv2d(3, 96, kernel_size=9, stride=2)
        self.pool = nn.MaxPool2d(kernel_size=3, stride=2)
        self.relu = nn.ReLU()
        self.scale2_onestack = _Seq(
            nn.Conv2d(160, 64, kernel_size=5, stride=1, padding=2, padding_mode='zeros'),
            nn.ReLU(),
            nn.Conv2d(64, 64, kernel_size=5, stride=1, padding=2, padding_mode='zeros'),
            nn.ReLU(),
            nn.Conv2d(64, 64, kernel_size=5, stride=1, padding=2, padding_mode='zeros'),
            nn.ReLU(),
            nn.ConvTranspose2d(64, 1, kernel_size=5, padding=2, padding_mode='zeros', stride=2))


class Scale3(_Container):
    """Eigen.py:46-69."""

    def __init__(self):
        super(Scale3, self).__init__()
        self.conv = nn.Conv2d(3, 96, kernel_size=9, stride=2)
        self.pool = nn.MaxPool2d(kernel_size=3, stride=1)
        self.relu = nn.ReLU()
        self.scale3_onestack = _Seq(
            nn.Conv2d(97, 64, kernel_size=5, padding=2, padding_mode="zeros"),
            nn.ReLU(),
            nn.Conv2d(64, 64, kernel_size=5, padding=2, padding_mode="zeros"),
            nn.ReLU(),
            nn.Conv2d(64, 64, kernel_size=5, padding=2, padding_mode="zeros"),
            nn.ReLU(),
            nn.Conv2d(64, 1, kernel_size=5, padding=2, padding_mode="zeros"),
            nn.ReLU())


# ---------------------------------------------------------------------------------------------- launch plan
class EigenEngine(G.TapeEngine):
    """The tape of Eigen.forward (Eigen.py:14-18 -> VGG.forward :81-89, Scale2.forward :36-43, Scale3.forward :62-69)."""

    def _conv_relu(self, x, conv):
        c = self.add(G.Conv(self, x, conv.weight, conv.kernel_size[0], conv.stride[0], conv.padding[0])).out
        return self.pw(c, bias=conv.bias, act="relu")

    def _plan(self):
        m, N, H, W = self.m, self.N, self.H, self.W
        if (H, W) != (240, 320):
            raise ValueError("Eigen: the two Linear layers (Eigen.py:77-78) fix the input at 240 x 320, got %d x %d" % (H, W))
        s1, s2, s3 = m.scale1, m.scale2, m.scale3
        # ---- scale 1: VGG-19-BN features
        f = s1.feature_extractor
        site0 = self._site([f[1]])
        self.stem = self.add(G.ImageStem(self, f[0], site0, N, H, W))
        x = self.add(G.BN(self, self.stem.out, site0, True, bias=f[0].bias)).out
        i = 3
        while i < len(f):
            if isinstance(f[i], nn.MaxPool2d):
                x = self.add(G.MaxPoolView(self, x, 2, 2)).out
                i += 1
            else:
                x = self.conv_bn(x, f[i], f[i + 1], True)
                i += 3
        assert (x.H, x.W, x.C) == (7, 10, 512)
        flat = self.add(G.PooledFlat(self, x, 1, 0.0)).out                    # nn.Flatten over NCHW: a 1x1 "pool" into the flatten order
        h = self.pw(self.add(G.Conv(self, flat, s1.mlp1.weight, 1)).out, bias=s1.mlp1.bias)
        h = self.pw(self.add(G.Conv(self, h, s1.mlp2.weight, 1)).out, bias=s1.mlp2.bias)
        up = self.add(G.ConvT(self, self.add(G.Unflatten(self, h, 64, 14, 19)).out, s1.upsample.weight, 3, 0, stride=4)).out
        assert (up.H, up.W) == (55, 75)
        # ---- scale 2: cat([pool(relu(conv(img)))[1:-1, 1:-1], scale 1]) -> 5x5 stack -> transposed conv
        cat2 = self.buf(N, 55, 75, 160)
        self.pw(up, bias=s1.upsample.bias, out=cat2.slice(96, 64))
        c2 = self.add(G.ImageStem(self, s2.conv, None, N, H, W, xin=self.stem.xin)).out       # 116 x 156
        r2 = self.pw(c2, bias=s2.conv.bias, act="relu")
        # MaxPool2d(3, 2) then [1:-1]: the kept windows start at rows / columns 2, 4, ...: a view at (2, 2) of 111 x 151
        self.add(G.MaxPoolView(self, r2, 3, 2, 2, 2, 2 * 54 + 3, 2 * 74 + 3, out=cat2.slice(0, 96)))
        st = s2.scale2_onestack
        y = self._conv_relu(self._conv_relu(self._conv_relu(cat2, st[0]), st[2]), st[4])
        t = self.add(G.ConvT(self, y, st[6].weight, 5, 2, stride=2)).out                       # 109 x 149, one channel (stored as 8)
        assert (t.H, t.W) == (109, 149)
        # ---- scale 3: cat([pool(relu(conv(img)[2:-3, 2:-3])), scale 2]) -> 5x5 stack; 97 channels are stored as 104
        c1 = t.C
        cat3 = self.buf(N, 109, 149, 96 + c1)
        self.pw(t, bias=st[6].bias, out=cat3.slice(96, c1))
        c3 = self.add(G.ImageStem(self, s3.conv, None, N, H, W, xin=self.stem.xin)).out
        r3 = self.pw(c3, bias=s3.conv.bias, act="relu")                        # (the ReLU commutes with the crop)
        self.add(G.MaxPoolView(self, r3, 3, 1, 2, 2, c3.H - 5, c3.W - 5, out=cat3.slice(0, 96)))
        st = s3.scale3_onestack
        z = self._conv_relu(self._conv_relu(self._conv_relu(cat3, st[0]), st[2]), st[4])
        z = self.add(G.Conv(self, z, st[6].weight, 5, 1, 2)).out
        self.heads = [self.add(G.ToNCHW(self, z, st[6].bias, 1, "relu"))]


class Eigen(G.TapeModule):
    """reference Eigen.py:5-18."""

    _engine_cls = EigenEngine

    def __init__(self, scale1='vgg', pretrained=True):
        super(Eigen, self).__init__()
        if scale1 == 'vgg':
            self.scale1 = VGG(pretrained=pretrained)
        else:
            raise NotImplementedError("Eigen: scale1=%r (the reference builds nothing for it either: Eigen.py:8-9)" % (scale1,))
        self.scale2 = Scale2()
        self.scale3 = Scale3()
        self._init_runtime()

    def _make_store(self, device):
        # modules/eigen.py:55-60: three groups at one learning rate; the flat store's two ranges are scale 1 and scales 2 + 3
        return G.NetStore(self, device, is_encoder=lambda n: n.startswith("scale1."))

    def forward(self, img):
        return self._run(img)[0]
