"""Drop-in for reference network/MyNet.py on MI355X: `MyModel(input_size, encoder_version)` with `.encoder` / `.decoder` (the two
learning-rate groups of modules/my.py:67-69), identical state_dict keys and construction order, and a forward that returns
the fp32 N x 1 x H x W depth — computed by hand-written gfx950 kernels through libmde_hip.so (graph.py tape).  The submodules
below only hold parameters.

Network (MyNet.py:4-282): the DenseNet encoder of the BTS path (relu0 / pool0 / transition1 / transition2 / norm5), one
residual conv unit per skip (FeatureFusionBlock with a single input), three decoder branches — "global consistency"
(nearest x2, concatenation, two ELU -> BN -> 3x3 convs), "details" (PixelShuffle, strided conv, concatenation, three
ELU -> BN -> 3x3 convs, nearest x2) and "sharpness" (three 4x4 / 2 transposed convs, concatenation, two nearest-x2 + biased
3x3 + ReLU) — a shared nearest-x2 + 3x3 + sigmoid depth head applied to each branch, and the "weighter": a shared
ELU -> BN -> 3x3 / 2 conv, a Linear over the flattened pixels, a sum over channels and a sigmoid give one scale per image and
branch; depth = 10 / 3 * sum of scale * branch depth.

`nn.AdaptiveMaxPool2d((H / 2, W / 2))` of GlobalConsitency (MyNet.py:21,26-27) is the identity when the image has the size the
model was built for (`input_size`), the only case this plan runs; the ResNet / ResNeXt encoders of MyNet.py:164-179 have no
plan here (as in network/Bts.py).
"""
import torch
import torch.nn as nn

from .. import graph as G
from .Bts import BtsEngine, _Container, _Seq, encoder  # noqa: F401  (the reference's MyNet.py defines the same encoder class)


class Conv2d(_Container):
    """MyNet.py:4-15: ELU -> BatchNorm(in_channels) -> conv (no bias)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        self.activation = nn.ELU()
        self.bn = nn.BatchNorm2d(in_channels)


class GlobalConsitency(_Container):
    def __init__(self, channels, input_size=(384, 384), out_feat=64):
        super().__init__()
        self.inc = nn.Upsample(scale_factor=2)
        self.avg = nn.AdaptiveMaxPool2d((input_size[0] // 2, input_size[1] // 2))
        self.conv = Conv2d(channels, channels // 2, kernel_size=3, padding=1, stride=1)
        self.conv_final = Conv2d(channels // 2, out_feat, kernel_size=3, padding=1, stride=1)


class Details(_Container):
    def __init__(self, channels, scale=2, out_feat=64):
        super().__init__()
        self.c = int(channels / (scale * scale))
        self.shuffle = nn.PixelShuffle(scale)
        self.down = Conv2d(self.c, self.c * 2, kernel_size=3, stride=2, padding=1)
        self.conv = Conv2d(self.c * 4, self.c * 2, kernel_size=3, stride=1, padding=1)
        self.conv2 = Conv2d(self.c * 2, self.c, kernel_size=3, stride=1, padding=1)
        self.conv_final = Conv2d(self.c, out_feat, kernel_size=3, stride=1, padding=1)
        self.up = nn.Upsample(scale_factor=2)


class Sharpness(_Container):
    def __init__(self, encoder_feature_sizes, out_feat=64):
        super().__init__()
        [feat0, feat1, feat2] = encoder_feature_sizes[2:5]
        self.tconv0 = nn.ConvTranspose2d(feat1, feat1 // 2, kernel_size=4, padding=1, stride=2)
        self.tconv1 = nn.ConvTranspose2d(feat2, feat2 // 4, kernel_size=4, padding=1, stride=2)
        self.tconv2 = nn.ConvTranspose2d(feat2 // 4, feat2 // 8, kernel_size=4, padding=1, stride=2)
        self.up0 = _Seq(nn.Upsample(scale_factor=2),
                        nn.Conv2d(feat0 + feat1 // 2 + feat2 // 8, out_feat * 2, kernel_size=3, stride=1, padding=1),
                        nn.ReLU())
        self.up1 = _Seq(nn.Upsample(scale_factor=2),
                        nn.Conv2d(out_feat * 2, out_feat, kernel_size=3, stride=1, padding=1),
                        nn.ReLU())


class Weighter(_Container):
    def __init__(self, input_size, in_feat):
        super().__init__()
        self.conv = Conv2d(in_feat, in_feat // 2, kernel_size=3, stride=2, padding=1)
        self.mlp = nn.Linear(input_size[0] * input_size[1] // 16, 1)


class ResidualConvUnit(_Container):
    """MyNet.py:196-230 (the ReLU is NOT in place here, unlike MiDaS')."""

    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.relu = nn.ReLU(inplace=False)


class FeatureFusionBlock(_Container):
    """MyNet.py:232-263.  my_decoder calls it with ONE input: output = resConfUnit2(x); resConfUnit1 holds parameters that
    never receive a gradient."""

    def __init__(self, features):
        super(FeatureFusionBlock, self).__init__()
        self.resConfUnit1 = ResidualConvUnit(features)
        self.resConfUnit2 = ResidualConvUnit(features)


class my_decoder(_Container):
    def __init__(self, input_size, encoder_feature_sizes):
        super().__init__()
        self.refine0 = FeatureFusionBlock(encoder_feature_sizes[0])
        self.refine1 = FeatureFusionBlock(encoder_feature_sizes[1])
        self.refine2 = FeatureFusionBlock(encoder_feature_sizes[2])
        self.refine3 = FeatureFusionBlock(encoder_feature_sizes[3])
        self.global_con = GlobalConsitency(encoder_feature_sizes[0] + encoder_feature_sizes[1], input_size=input_size, out_feat=64)
        self.details = Details(encoder_feature_sizes[1], out_feat=64)
        self.sharpness = Sharpness(encoder_feature_sizes, out_feat=64)
        self.weighter = Weighter(input_size=input_size, in_feat=64)
        self.get_depth = _Seq(nn.Upsample(scale_factor=2), nn.Conv2d(64, 1, 3, 1, 1, bias=False), nn.Sigmoid())
        self.input_size = tuple(input_size)


# ---------------------------------------------------------------------------------------------- launch plan
class MyEngine(BtsEngine):
    """The tape of MyModel.forward (MyNet.py:270-272 -> encoder.forward :181-192 -> my_decoder.forward :135-157)."""

    def _rcu(self, x, u, out=None):
        r = self.pw(x, act="relu")
        c = self.add(G.Conv(self, r, u.conv1.weight, 3, 1, 1)).out
        t = self.pw(c, bias=u.conv1.bias, act="relu")
        c = self.add(G.Conv(self, t, u.conv2.weight, 3, 1, 1)).out
        return self.pw(c, bias=u.conv2.bias, r=x, out=out)

    def _pre(self, x, blk, out=None):
        """Conv2d.forward (MyNet.py:11-15)."""
        a = self.pw(x, act="elu")
        b = self._bn_after_elu(a, blk.bn, None)
        cv = blk.conv
        return self.add(G.Conv(self, b, cv.weight, cv.kernel_size[0], cv.stride[0], cv.padding[0], out=out)).out

    def _tconv(self, x, tc, out=None):
        k, p = tc.kernel_size[0], tc.padding[0]
        assert tc.stride == (2, 2) and tc.output_padding == (0, 0)
        y = self.add(G.ConvT(self, x, tc.weight, k, p)).out
        return self.pw(y, bias=tc.bias, out=out)

    def _up_conv_relu(self, x, seq):
        u = self.add(G.Nearest2(self, x)).out
        c = self.add(G.Conv(self, u, seq[1].weight, 3, 1, 1)).out
        return self.pw(c, bias=seq[1].bias, act="relu")

    def _plan(self):
        m, N, H, W = self.m, self.N, self.H, self.W
        d = m.decoder
        if (H, W) != d.input_size:
            raise NotImplementedError("HIP MyModel: built for %s images, got %d x %d (GlobalConsitency's AdaptiveMaxPool2d and the Weighter's "
                                      "Linear layer fix the size; only the identity pooling case has a plan)" % (d.input_size, H, W))
        if H % 32 or W % 32:
            raise ValueError("MyModel: image sizes must be multiples of 32 (got %d x %d)" % (H, W))
        skip0, skip1, skip2, skip3, dense = self._dense_trunk(m.encoder.base_model, N, H, W)
        gc, dt, sh = d.global_con, d.details, d.sharpness
        # global consistency: cat([refine0(skip0), up(refine1(skip1))]) at 1/2
        gcat = self.buf(N, skip0.H, skip0.W, skip0.C + skip1.C)
        self._rcu(skip0, d.refine0.resConfUnit2, out=gcat.slice(0, skip0.C))
        x1 = self._rcu(skip1, d.refine1.resConfUnit2)
        x2 = self._rcu(skip2, d.refine2.resConfUnit2)
        x3 = self._rcu(skip3, d.refine3.resConfUnit2)
        self.add(G.Nearest2(self, x1, out=gcat.slice(skip0.C, skip1.C)))
        glob = self._pre(self._pre(gcat, gc.conv), gc.conv_final)
        # details: cat([down(shuffle(x1)), shuffle(x2)]) at 1/4
        c = dt.c
        dcat = self.buf(N, skip1.H, skip1.W, 4 * c)
        s1 = self.add(G.PixelShuffle2(self, x1)).out
        self._pre(s1, dt.down, out=dcat.slice(0, 2 * c))
        self.add(G.PixelShuffle2(self, x2, out=dcat.slice(2 * c, 2 * c)))
        t = self._pre(self._pre(self._pre(dcat, dt.conv), dt.conv2), dt.conv_final)
        detail = self.add(G.Nearest2(self, t)).out
        # sharpness: cat([x2, tconv0(x3), tconv2(tconv1(relu(norm5)))]) at 1/8
        c0 = self.store.sdims[id(sh.tconv0.weight)][3]
        c2 = self.store.sdims[id(sh.tconv2.weight)][3]                 # (276 output channels are stored as 280, the padding is zero)
        scat = self.buf(N, skip2.H, skip2.W, x2.C + c0 + c2)
        self._copy(x2, scat.slice(0, x2.C))
        self._tconv(x3, sh.tconv0, out=scat.slice(x2.C, c0))
        self._tconv(self._tconv(dense, sh.tconv1), sh.tconv2, out=scat.slice(x2.C + c0, c2))
        sharp = self._up_conv_relu(self._up_conv_relu(scat, sh.up0), sh.up1)
        # the shared depth head on each branch, the weighter's scale per image and branch, and their combination
        branches = (glob, detail, sharp)
        maps = []
        for b in branches:
            u = self.add(G.Nearest2(self, b)).out
            cv = self.add(G.Conv(self, u, d.get_depth[1].weight, 3, 1, 1)).out
            maps.append(self.add(G.SigmoidMap(self, cv)).map)
        comb = G.Combine3(self, maps, 10.0 / 3.0)
        pools = []
        for k, b in enumerate(branches):
            a = self._pre(b, d.weighter.conv)
            pools.append(self.add(G.WeightedPool(self, a, d.weighter.mlp, comb.ds[k])))
        comb.scales = [p.scale for p in pools]
        self.heads = [self.add(comb)]


class MyModel(G.TapeModule):
    """reference MyNet.py:264-272."""

    _engine_cls = MyEngine

    def __init__(self, input_size=(384, 384), encoder_version='densenet161_bts'):
        super(MyModel, self).__init__()
        self.encoder = encoder(encoder_version)
        self.decoder = my_decoder(input_size, self.encoder.feat_out_channels)
        self._init_runtime()

    def _make_store(self, device):
        return G.NetStore(self, device, is_encoder=lambda n: n.startswith("encoder."), raw=G.stem7_weights(self))      # my.py:67-69

    def forward(self, x):
        return self._run(x)[0]
