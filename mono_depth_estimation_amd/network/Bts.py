"""Drop-in for reference network/Bts.py on MI355X: `BtsModel(bts_size, max_depth, out_channels, image_residuals,
encoder_version)` with the `.encoder` / `.decoder` attributes modules/bts.py:90,140-141 uses, identical state_dict keys,
and a forward that returns bts.forward's 5-tuple (depth_8x8_scaled, depth_4x4_scaled, depth_2x2_scaled, reduc1x1,
final_depth), fp32 — computed by hand-written gfx950 kernels (mono_depth_estimation_amd/graph.py tape).  The submodules
only hold parameters.

Network (Bts.py:51-146,148-333): DenseNet-161 / -121 trunk (torchvision's architecture, built in place: BN -> ReLU -> 1x1 ->
BN -> ReLU -> 3x3 dense layers whose 48 new channels are written straight into the block's concatenation tensor; the
batch moments of every channel group are reduced once and shared by all the BatchNorms that normalise it again), the
decoder's nearest-x2 up-convolutions with ELU, dense ASPP (BN -> ReLU -> 1x1 -> BN -> ReLU -> dilated 3x3, dilations
3 / 6 / 12 / 18 / 24), the reduction_1x1 chains down to three plane parameters, local planar guidance at 8x / 4x / 2x, and
the sigmoid depth head, with `image_residuals` the colour channels as residuals on the input image (Bts.py:264-271).  Encoders:
densenet121 / 161 and the whole-model ResNet-50 / -101, ResNeXt-50 32x4d / -101 32x8d of Bts.py:293-307 (grouped 3x3 convs as
block-diagonal tiles); ResNet-50 and ResNeXt-50 are pinned by goldens, the 101-layer variants by their parameter tree only.
"""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import graph as G


def bn_init_as_tf(m):
    """Bts.py:26-31."""
    if isinstance(m, nn.BatchNorm2d):
        m.track_running_stats = True
        m.eval()
        m.affine = True
        m.requires_grad = True


def weights_init_xavier(m):
    """Bts.py:34-38."""
    if isinstance(m, nn.Conv2d):
        torch.nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            torch.nn.init.zeros_(m.bias)


class silog_loss(nn.Module):
    """Bts.py:41-48: SILog over an explicit boolean mask (criteria.silog_loss masks gt > 1e-2 itself; pixels outside `mask`
    are handed to it as invalid depth)."""

    def __init__(self, variance_focus):
        super(silog_loss, self).__init__()
        self.variance_focus = variance_focus

    def forward(self, depth_est, depth_gt, mask):
        from .. import criteria
        if bool((depth_gt[mask] <= 1e-2).any()):
            raise ValueError("Bts.silog_loss: the mask selects pixels with depth_gt <= 1e-2, which the SILog kernel treats as invalid")
        return criteria.silog_loss(self.variance_focus)(depth_est, torch.where(mask, depth_gt, torch.zeros_like(depth_gt)))


class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("%s is a parameter container of the HIP BTS path; call the BtsModel instead" % type(self).__name__)


class _Seq(nn.Sequential):
    def forward(self, *a, **k):
        raise RuntimeError("this Sequential is a parameter container of the HIP BTS path; call the BtsModel instead")


# ---------------------------------------------------------------------------------------------- decoder parameter tree
class atrous_conv(_Seq):
    """Bts.py:51-66."""

    def __init__(self, in_channels, out_channels, dilation, apply_bn_first=True):
        super(atrous_conv, self).__init__()
        self.dilation = dilation
        self.atrous_conv = _Seq()
        if apply_bn_first:
            self.atrous_conv.add_module('first_bn', nn.BatchNorm2d(in_channels, momentum=0.01, affine=True, track_running_stats=True, eps=1.1e-5))
        self.atrous_conv.add_module('aconv_sequence', _Seq(
            nn.ReLU(),
            nn.Conv2d(in_channels=in_channels, out_channels=out_channels * 2, bias=False, kernel_size=1, stride=1, padding=0),
            nn.BatchNorm2d(out_channels * 2, momentum=0.01, affine=True, track_running_stats=True),
            nn.ReLU(),
            nn.Conv2d(in_channels=out_channels * 2, out_channels=out_channels, bias=False, kernel_size=3, stride=1,
                      padding=(dilation, dilation), dilation=dilation)))


class upconv(_Container):
    """Bts.py:69-80."""

    def __init__(self, in_channels, out_channels, ratio=2):
        super(upconv, self).__init__()
        if ratio != 2:
            raise NotImplementedError("HIP upconv: ratio 2")
        self.elu = nn.ELU()
        self.conv = nn.Conv2d(in_channels=in_channels, out_channels=out_channels, bias=False, kernel_size=3, stride=1, padding=1)
        self.ratio = ratio


class reduction_1x1(_Seq):
    """Bts.py:83-122."""

    def __init__(self, num_in_filters, num_out_filters, max_depth, is_final=False):
        super(reduction_1x1, self).__init__()
        self.max_depth = max_depth
        self.is_final = is_final
        self.sigmoid = nn.Sigmoid()
        self.reduc = _Seq()
        while num_out_filters >= 4:
            if num_out_filters < 8:
                if self.is_final:
                    self.reduc.add_module('final', _Seq(nn.Conv2d(num_in_filters, out_channels=1, bias=False, kernel_size=1, stride=1, padding=0),
                                                        nn.Sigmoid()))
                else:
                    self.reduc.add_module('plane_params', nn.Conv2d(num_in_filters, out_channels=3, bias=False, kernel_size=1, stride=1, padding=0))
                break
            else:
                self.reduc.add_module('inter_{}_{}'.format(num_in_filters, num_out_filters),
                                      _Seq(nn.Conv2d(in_channels=num_in_filters, out_channels=num_out_filters, bias=False, kernel_size=1,
                                                     stride=1, padding=0), nn.ELU()))
            num_in_filters = num_out_filters
            num_out_filters = num_out_filters // 2


class local_planar_guidance(_Container):
    """Bts.py:124-146 (no parameters)."""

    def __init__(self, upratio):
        super(local_planar_guidance, self).__init__()
        self.upratio = float(upratio)


class bts(_Container):
    """Bts.py:148-203."""

    def __init__(self, max_depth, feat_out_channels, out_channels=20, image_residuals=False, num_features=512, dataset='nyu'):
        super(bts, self).__init__()
        self.max_depth, self.image_residuals, self.dataset, self.out_channels = max_depth, image_residuals, dataset, out_channels
        nf, f = num_features, feat_out_channels
        bn = lambda c: nn.BatchNorm2d(c, momentum=0.01, affine=True, eps=1.1e-5)
        conv_elu = lambda i, o: _Seq(nn.Conv2d(i, o, 3, 1, 1, bias=False), nn.ELU())
        self.upconv5 = upconv(f[4], nf)
        self.bn5 = bn(nf)
        self.conv5 = conv_elu(nf + f[3], nf)
        self.upconv4 = upconv(nf, nf // 2)
        self.bn4 = bn(nf // 2)
        self.conv4 = conv_elu(nf // 2 + f[2], nf // 2)
        self.bn4_2 = bn(nf // 2)
        self.daspp_3 = atrous_conv(nf // 2, nf // 4, 3, apply_bn_first=False)
        self.daspp_6 = atrous_conv(nf // 2 + nf // 4 + f[2], nf // 4, 6)
        self.daspp_12 = atrous_conv(nf + f[2], nf // 4, 12)
        self.daspp_18 = atrous_conv(nf + nf // 4 + f[2], nf // 4, 18)
        self.daspp_24 = atrous_conv(nf + nf // 2 + f[2], nf // 4, 24)
        self.daspp_conv = conv_elu(nf + nf // 2 + nf // 4, nf // 4)
        self.reduc8x8 = reduction_1x1(nf // 4, nf // 4, self.max_depth)
        self.lpg8x8 = local_planar_guidance(8)
        self.upconv3 = upconv(nf // 4, nf // 4)
        self.bn3 = bn(nf // 4)
        self.conv3 = conv_elu(nf // 4 + f[1] + 1, nf // 4)
        self.reduc4x4 = reduction_1x1(nf // 4, nf // 8, self.max_depth)
        self.lpg4x4 = local_planar_guidance(4)
        self.upconv2 = upconv(nf // 4, nf // 8)
        self.bn2 = bn(nf // 8)
        self.conv2 = conv_elu(nf // 8 + f[0] + 1, nf // 8)
        self.reduc2x2 = reduction_1x1(nf // 8, nf // 16, self.max_depth)
        self.lpg2x2 = local_planar_guidance(2)
        self.upconv1 = upconv(nf // 8, nf // 16)
        self.reduc1x1 = reduction_1x1(nf // 16, nf // 32, self.max_depth, is_final=True)
        self.conv1 = conv_elu(nf // 16 + 4, nf // 16)
        self.get_depth = _Seq(nn.Conv2d(nf // 16, out_channels, 3, 1, 1, bias=False), nn.Sigmoid())


# ---------------------------------------------------------------------------------------------- encoder parameter tree
class _DenseLayer(_Container):
    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)


class _DenseBlock(_Container):
    def __init__(self, n, cin, growth, bn_size):
        super().__init__()
        for i in range(n):
            self.add_module("denselayer%d" % (i + 1), _DenseLayer(cin + i * growth, growth, bn_size))


def _densenet_features(growth, blocks, init, bn_size=4):
    """torchvision.models.densenet.DenseNet(...).features: the same child names, hence the same state_dict keys."""
    feats = [("conv0", nn.Conv2d(3, init, 7, 2, 3, bias=False)), ("norm0", nn.BatchNorm2d(init)), ("relu0", nn.ReLU(inplace=True)),
             ("pool0", nn.MaxPool2d(3, 2, 1))]
    c = init
    for i, n in enumerate(blocks):
        feats.append(("denseblock%d" % (i + 1), _DenseBlock(n, c, growth, bn_size)))
        c += n * growth
        if i != len(blocks) - 1:
            feats.append(("transition%d" % (i + 1), _Seq(OrderedDict([
                ("norm", nn.BatchNorm2d(c)), ("relu", nn.ReLU(inplace=True)), ("conv", nn.Conv2d(c, c // 2, 1, bias=False)),
                ("pool", nn.AvgPool2d(2, 2))]))))
            c //= 2
    feats.append(("norm5", nn.BatchNorm2d(c)))
    seq = _Seq(OrderedDict(feats))
    for m in seq.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
    return seq


def _torchvision_resnet(blocks, groups=1, base_width=64):
    """torchvision.models.resnet50 / resnet101 / resnext50_32x4d / resnext101_32x8d kept WHOLE, as Bts.py:293-307 keeps them:
    conv1, bn1, relu, maxpool, layer1 .. layer4, avgpool, fc under torchvision's names (absent from the image and from
    /root/reference; restated from the public definition).  The forward walk skips avgpool and fc (Bts.py:313-315), but their
    parameters are part of the state_dict.  fc never receives a gradient: torch's optimisers skip it (grad is None); it is
    frozen here so that the fused flat-range AdamW step, which decays every entry of a range, leaves it alone as well."""
    from .MiDaS import _stage
    m = _Container()
    m.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
    m.bn1 = nn.BatchNorm2d(64)
    m.relu = nn.ReLU(inplace=True)
    m.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    m.layer1 = _stage(64, 64, blocks[0], 1, groups, base_width)
    m.layer2 = _stage(256, 128, blocks[1], 2, groups, base_width)
    m.layer3 = _stage(512, 256, blocks[2], 2, groups, base_width)
    m.layer4 = _stage(1024, 512, blocks[3], 2, groups, base_width)
    m.avgpool = nn.AdaptiveAvgPool2d((1, 1))
    m.fc = nn.Linear(2048, 1000)
    for mod in m.modules():                          # torchvision's default initialisation of the trunk
        if isinstance(mod, nn.Conv2d):
            nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
    for p in m.fc.parameters():
        p.requires_grad_(False)
    return m


_RESNETS = {'resnet50_bts': ((3, 4, 6, 3), 1, 64), 'resnet101_bts': ((3, 4, 23, 3), 1, 64),
            'resnext50_bts': ((3, 4, 6, 3), 32, 4), 'resnext101_bts': ((3, 4, 23, 3), 32, 8)}


class encoder(_Container):
    """Bts.py:280-321.  `pretrained=True` there downloads torchvision weights; here the trunk keeps its random initialisation
    until a checkpoint is loaded (load_state_dict with the reference's keys)."""

    def __init__(self, version):
        super(encoder, self).__init__()
        if version == 'densenet121_bts':
            self.base_model = _densenet_features(32, (6, 12, 24, 16), 64)
            self.feat_names = ['relu0', 'pool0', 'transition1', 'transition2', 'norm5']
            self.feat_out_channels = [64, 64, 128, 256, 1024]
        elif version == 'densenet161_bts':
            self.base_model = _densenet_features(48, (6, 12, 36, 24), 96)
            self.feat_names = ['relu0', 'pool0', 'transition1', 'transition2', 'norm5']
            self.feat_out_channels = [96, 96, 192, 384, 2208]
        elif version in _RESNETS:
            self.base_model = _torchvision_resnet(*_RESNETS[version])
            self.feat_names = ['relu', 'layer1', 'layer2', 'layer3', 'layer4']
            self.feat_out_channels = [64, 256, 512, 1024, 2048]
        else:
            raise NotImplementedError("BTS encoder %r (the reference prints 'Not supported encoder' and builds nothing, Bts.py:306-307)"
                                      % (version,))
        self.version = version


class _Part:
    def __init__(self, part):
        self.part = part


# ---------------------------------------------------------------------------------------------- launch plan
class BtsEngine(G.TapeEngine):
    """The tape of BtsModel.forward (Bts.py:329-333 -> encoder.forward :309-321 -> bts.forward :205-278)."""

    def _conv_elu(self, x, conv, out=None, **kw):
        c = self.add(G.Conv(self, x, conv.weight, conv.kernel_size[0], 1, conv.padding[0], conv.dilation[0], **kw)).out
        return self.pw(c, act="elu", out=out)

    def _upconv(self, x, up, out=None):
        return self._conv_elu(self.add(G.Nearest2(self, x)).out, up.conv, out=out)

    def _bn_after_elu(self, x, bn, out):
        """conv -> ELU -> BatchNorm (Bts.py:216-229).  Where the ELU rides in the conv's epilogue and this BatchNorm is its only
        reader, EVAL mode stores the conv's pre-activation and lets the BatchNorm's pass apply ELU in fp32 (mde_bn_apply, relu = 2):
        a stored ELU output piles up on -1 -- every value of (-1, -1 + 2^-10) rounds to -1 exactly -- and that bias, the same
        sign for every saturated element, is what the BatchNorm's 1 / sigma amplifies (measured on the off-grid fixture with the
        oracle: this ONE rounding, behind upconv2, moved AbsRel by 5.8e-5 of the 6.9e-5 all 412 roundings moved it by)."""
        prod = self.tape[-1] if self.tape else None
        s = self._site([bn])
        self.add(G.StatsPass(self, x, s))
        op = self.add(G.BN(self, x, s, False, out=out))
        if isinstance(prod, G.Conv) and prod.fused and prod.f_act == "elu" and prod.f_res is None and prod.out is x:
            prod.eval_raw, op.pre_elu = True, True
        return op.out

    def _copy(self, x, out):
        return self.pw(x, out=out)

    def _atrous(self, x, ac, out):
        seq = ac.atrous_conv.aconv_sequence
        if hasattr(ac.atrous_conv, "first_bn"):
            s = self._site([ac.atrous_conv.first_bn])
            self.add(G.StatsPass(self, x, s))
            r = self.add(G.BN(self, x, s, True)).out
        else:
            r = self.pw(x, act="relu")
        a = self.conv_bn(r, seq[1], seq[2], True)
        return self.add(G.Conv(self, a, seq[4].weight, 3, 1, ac.dilation, ac.dilation, out=out)).out

    def _reduc(self, x, red):
        for name, mod in red.reduc.named_children():
            if name.startswith("inter_"):
                x = self._conv_elu(x, mod[0])
        last = red.reduc.final[0] if red.is_final else red.reduc.plane_params
        return self.add(G.Conv(self, x, last.weight, 1)).out

    def _dense_trunk(self, feats, N, H, W):
        dev = self.dev
        s0 = self._site([feats.norm0])
        self.stem = self.add(G.image_stem(self, feats.conv0, s0, N, H, W))     # (96 channels for densenet161, 64 for densenet121)
        relu0 = self.add(G.BN(self, self.stem.out, s0, True)).out
        pool0 = self.add(G.MaxPool(self, relu0)).out
        skips, x, c = [relu0, pool0], pool0, pool0.C
        blocks = [m for n, m in feats.named_children() if n.startswith("denseblock")]
        for b, blk in enumerate(blocks):
            layers = list(blk.children())
            growth = layers[0].conv2.out_channels
            ctot = c + len(layers) * growth
            h, w = (x.H, x.W) if b == 0 else (x.H // 2, x.W // 2)
            buf = self.buf(N, h, w, ctot)
            mean, var = torch.zeros(ctot, device=dev), torch.ones(ctot, device=dev)
            head = buf.slice(0, c)
            if b == 0:
                self._copy(x, head)
            else:
                self.add(G.AvgPool2(self, x, out=head))
                if b < 3:
                    skips.append(head)
            self.add(G.Moments(self, head, mean, var, 0))
            for i, L in enumerate(layers):
                k = c + i * growth
                t = self.add(G.PrefixBN(self, buf.slice(0, k), L.norm1, mean, var)).out
                a = self.conv_bn(t, L.conv1, L.norm2, True)
                new = buf.slice(k, growth)
                part = _Part(G.ops.new_stat_buffer(growth, dev))
                self.add(G.Conv(self, a, L.conv2.weight, 3, 1, 1, out=new, site=part))
                self.add(G.Moments(self, new, mean, var, k, part=part.part))
            c = ctot
            if b < len(blocks) - 1:
                tr = getattr(feats, "transition%d" % (b + 1))
                t = self.add(G.PrefixBN(self, buf, tr.norm, mean, var)).out
                x = self.add(G.Conv(self, t, tr.conv.weight, 1)).out
                c //= 2
            else:
                # norm5, then the decoder's torch.nn.ReLU()(features[5]) (Bts.py:207) in the same pass
                skips.append(self.add(G.PrefixBN(self, buf, feats.norm5, mean, var)).out)
        return skips

    def _resnet_trunk(self, rn, N, H, W):
        """encoder.forward (Bts.py:309-321) over a whole torchvision ResNet / ResNeXt: 'relu' (after conv1 / bn1), then
        layer1 .. layer4 are the five features; the 3x3 convs of the ResNeXt variants run as block-diagonal grouped tiles."""
        s0 = self._site([rn.bn1])
        self.stem = self.add(G.image_stem(self, rn.conv1, s0, N, H, W))
        relu = self.add(G.BN(self, self.stem.out, s0, True)).out
        x = self.add(G.MaxPool(self, relu)).out
        skips = [relu]
        for stage in (rn.layer1, rn.layer2, rn.layer3, rn.layer4):
            for b in stage:
                ds = b.downsample
                x = self.bottleneck(x, b.conv1, b.bn1, b.conv2, b.bn2, b.conv3, b.bn3,
                                    ds[0] if ds is not None else None, ds[1] if ds is not None else None)
            skips.append(x)
        return skips                                  # (layer4 ends in a ReLU: the decoder's ReLU on it, Bts.py:207, is the identity)

    def _plan(self):
        m, N, H, W = self.m, self.N, self.H, self.W
        if H % 32 or W % 32:
            raise ValueError("BtsModel: image sizes must be multiples of 32 (got %d x %d)" % (H, W))
        d, md = m.decoder, float(m.decoder.max_depth)
        base = m.encoder.base_model
        trunk = self._resnet_trunk if hasattr(base, "layer1") else self._dense_trunk
        skip0, skip1, skip2, skip3, dense = trunk(base, N, H, W)
        nf = d.bn5.num_features
        # 1/16: upconv5 -> bn5 | skip3 -> conv5
        cat5 = self.buf(N, skip3.H, skip3.W, nf + skip3.C)
        self._bn_after_elu(self._upconv(dense, d.upconv5), d.bn5, cat5.slice(0, nf))
        self._copy(skip3, cat5.slice(nf, skip3.C))
        i5 = self._conv_elu(cat5, d.conv5[0])
        # 1/8: upconv4 -> bn4 | skip2 -> conv4 -> bn4_2, then the dense ASPP (Bts.py:218-232)
        q, c4 = nf // 4, nf // 2 + skip2.C
        grow = self.buf(N, skip2.H, skip2.W, c4 + 5 * q)                  # concat4, then + daspp_3 / 6 / 12 / 18 / 24
        self._bn_after_elu(self._upconv(i5, d.upconv4), d.bn4, grow.slice(0, nf // 2))
        self._copy(skip2, grow.slice(nf // 2, skip2.C))
        dcat = self.buf(N, skip2.H, skip2.W, nf // 2 + 5 * q)             # cat([iconv4, daspp_3 .. daspp_24])
        i4 = self._bn_after_elu(self._conv_elu(grow.slice(0, c4), d.conv4[0]), d.bn4_2, dcat.slice(0, nf // 2))
        for j, ac in enumerate((d.daspp_3, d.daspp_6, d.daspp_12, d.daspp_18, d.daspp_24)):
            src = i4 if j == 0 else grow.slice(0, c4 + j * q)
            o = self._atrous(src, ac, grow.slice(c4 + j * q, q))
            self._copy(o, dcat.slice(nf // 2 + j * q, q))
        feat = self._conv_elu(dcat, d.daspp_conv[0])
        d8 = self.add(G.PlaneDepth(self, self._reduc(feat, d.reduc8x8), 8, md))
        # 1/4
        c3 = q + skip1.C
        cat3 = self.buf(N, skip1.H, skip1.W, (c3 + 1 + 7) // 8 * 8)
        self._bn_after_elu(self._upconv(feat, d.upconv3), d.bn3, cat3.slice(0, q))
        self._copy(skip1, cat3.slice(q, skip1.C))
        self.add(G.MapSlot(self, d8.map, cat3, c3, 4))
        i3 = self._conv_elu(cat3, d.conv3[0])
        d4 = self.add(G.PlaneDepth(self, self._reduc(i3, d.reduc4x4), 4, md))
        # 1/2
        e, c2 = nf // 8, nf // 8 + skip0.C
        cat2 = self.buf(N, skip0.H, skip0.W, (c2 + 1 + 7) // 8 * 8)
        self._bn_after_elu(self._upconv(i3, d.upconv2), d.bn2, cat2.slice(0, e))
        self._copy(skip0, cat2.slice(e, skip0.C))
        self.add(G.MapSlot(self, d4.map, cat2, c2, 2))
        i2 = self._conv_elu(cat2, d.conv2[0])
        d2 = self.add(G.PlaneDepth(self, self._reduc(i2, d.reduc2x2), 2, md))
        # 1/1
        s = nf // 16
        cat1 = self.buf(N, H, W, (s + 4 + 7) // 8 * 8)
        up1 = self._upconv(i2, d.upconv1, out=cat1.slice(0, s))
        r1 = self.add(G.SigmoidMap(self, self._reduc(up1, d.reduc1x1)))
        for j, mp in enumerate((r1.map, d2.map, d4.map, d8.map)):
            self.add(G.MapSlot(self, mp, cat1, s + j, 1))
        i1 = self._conv_elu(cat1, d.conv1[0])
        oc = d.get_depth[0].out_channels
        if oc == 1 and i1.C in (8, 16, 32, 64) and i1.ld == i1.C and (H * W) % 4 == 0 and os.environ.get("MDE_BTS_HEADMAP", "1") != "0":   # ("0": the GEMM path, diagnostics)
            final = self.add(G.HeadConvMap(self, i1, d.get_depth[0].weight, "sigmoid", md))       # one output channel: the head kernels
        else:
            c = self.add(G.Conv(self, i1, d.get_depth[0].weight, 3, 1, 1)).out
            if d.out_channels == 10 and d.image_residuals:          # Bts.py:264-271 (no max_depth factor on this branch)
                final = self.add(G.ImageResidualHead(self, c, oc))
            else:
                final = self.add(G.ToNCHW(self, c, None, oc, "sigmoid", md))
        self.heads = [d8, d4, d2, r1, final]


class BtsModel(G.TapeModule):
    """reference Bts.py:324-333."""

    _engine_cls = BtsEngine

    def __init__(self, bts_size=512, max_depth=10, out_channels=20, image_residuals=False, encoder_version='densenet161_bts'):
        super(BtsModel, self).__init__()
        self.encoder = encoder(encoder_version)
        self.decoder = bts(max_depth, self.encoder.feat_out_channels, num_features=bts_size, out_channels=out_channels,
                           image_residuals=image_residuals)
        self._init_runtime()

    def _make_store(self, device):
        # (the grouped 3x3 weights of the ResNeXt encoders, [O][G][3][3] with G = 4 or 8, keep their exact shape: their
        #  block-diagonal packings are built from it)
        raw = [n + ".weight" for n, m in self.named_modules() if isinstance(m, nn.Conv2d) and m.groups > 1]
        raw += G.stem7_weights(self)               # (the 7x7 / 2 image conv: [O][7][7][3], the stem kernels' operand)
        return G.NetStore(self, device, is_encoder=lambda n: n.startswith("encoder."), raw=raw)      # bts.py:140-141

    def forward(self, x, focal=518.8579):
        return self._run(x)
