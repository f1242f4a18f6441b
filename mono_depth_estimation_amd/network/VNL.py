"""Drop-in for reference network/VNL.py on MI355X: `MetricDepthModel(params)` with the attribute paths
modules/vnl.py:165-179 walks (`.depth_model.encoder_modules`, `.depth_model.decoder_modules.top / topdown_fcn1..5 /
topdown_predict`), identical state_dict keys, the reference's initialisation rules, and a forward that returns
`(logits, softmax)` as fp32 N x 150 x H x W — computed by hand-written gfx950 kernels through libmde_hip.so
(mono_depth_estimation_amd/graph.py tape).  The submodules below only hold parameters.

Network (VNL.py:97-387,539-693): ResNeXt-50/101 32x4d body at output stride 16 (grouped 3x3 convs, res5 dilated by 2),
ASPP on res5 (1x1 + three dilated 3x3 + image pooling), FTB lateral blocks, a top-down decoder of AFA gates + FTB
blocks with bilinear(align_corners) upsampling, and a dilated 3x3 prediction conv + softmax over 150 depth bins.
The MobileNetV2 encoder option (`mobilenetv2_body_stride8`, VNL.py:389-537: inverted residuals around depthwise 3x3
convolutions, ReLU6, a global-pooling block in the ASPP's place) runs on csrc/dwconv.hip's streaming kernels.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import graph as G


class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("%s is a parameter container of the HIP VNL path; call the MetricDepthModel instead" % type(self).__name__)


class _Seq(nn.Sequential):
    def forward(self, *a, **k):
        raise RuntimeError("this Sequential is a parameter container of the HIP VNL path; call the MetricDepthModel instead")


# ---------------------------------------------------------------------------------------------- parameter tree
class FTB_block(_Container):
    """VNL.py:330-350."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.conv1 = nn.Conv2d(dim_in, dim_out, 1, stride=1, padding=0, bias=False)
        self.conv2 = nn.Conv2d(dim_out, dim_out, 3, stride=1, padding=2, dilation=2, bias=True)
        self.bn1 = nn.BatchNorm2d(dim_out, momentum=0.5)
        self.relu = nn.ReLU(inplace=True)
        self.conv3 = nn.Conv2d(dim_out, dim_out, 3, stride=1, padding=2, dilation=2, bias=False)


class AFA_block(_Container):
    """VNL.py:353-373."""

    def __init__(self, dim):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_mid = dim * 2, dim, int(dim / 8)
        self.globalpool = nn.AdaptiveAvgPool2d(1)
        self.conv1 = nn.Conv2d(self.dim_in, self.dim_mid, 1, stride=1, padding=0, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(self.dim_mid, self.dim_out, 1, stride=1, padding=0, bias=False)
        self.sigmd = nn.Sigmoid()


class lateral_block(_Container):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.lateral = FTB_block(dim_in, dim_out)


class ASPP_block(_Container):
    """VNL.py:189-228."""

    def __init__(self, dim_in, dim_out, dilate_rates, output_stride):
        super().__init__()
        self.dim_in, self.dim_out, self.dilate_rates = dim_in, dim_out, dilate_rates
        self.aspp_conv1x1 = nn.Conv2d(dim_in, dim_out, 1, stride=1, padding=0, bias=False)
        for i, d in enumerate(dilate_rates):
            setattr(self, "aspp_conv3_%d" % (i + 1), nn.Conv2d(dim_in, dim_out, 3, stride=1, padding=d, dilation=d, bias=False))
        self.aspp_bn1x1 = nn.BatchNorm2d(dim_out, momentum=0.5)
        for i in range(3):
            setattr(self, "aspp_bn3_%d" % (i + 1), nn.BatchNorm2d(dim_out, momentum=0.5))
        self.globalpool = nn.AdaptiveAvgPool2d((1, 1))
        self.globalpool_conv1x1 = nn.Conv2d(dim_in, dim_out, 1, stride=1, padding=0, bias=False)
        self.globalpool_bn = nn.BatchNorm2d(dim_out, momentum=0.5)


class ResNeXtBottleneck(_Container):
    """VNL.py:618-669 (type C: stride and dilation on the grouped 3x3)."""

    def __init__(self, in_channels, out_channels, stride, dilate, cardinality=32, base_width=4):
        super().__init__()
        D = cardinality * base_width * int(out_channels / 256.)
        self.stride, self.dilate, self.cardinality = stride, dilate, cardinality
        self.conv1 = nn.Conv2d(in_channels, D, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn1 = nn.BatchNorm2d(D)
        self.conv2 = nn.Conv2d(D, D, kernel_size=3, stride=stride, padding=dilate, dilation=dilate, groups=cardinality, bias=False)
        self.bn2 = nn.BatchNorm2d(D)
        self.conv3 = nn.Conv2d(D, out_channels, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn3 = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        if in_channels != out_channels:
            self.shortcut = _Seq()
            self.shortcut.add_module('conv', nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, padding=0, bias=False))
            self.shortcut.add_module('bn', nn.BatchNorm2d(out_channels))
        else:
            self.shortcut = None


def basic_bn_stem():
    return _Seq(OrderedDict([
        ('conv1', nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)),
        ('bn1', nn.BatchNorm2d(64)),
        ('relu', nn.ReLU(inplace=True)),
        ('maxpool', nn.MaxPool2d(kernel_size=3, stride=2, padding=1))]))


def add_stage(inplanes, outplanes, nblocks, cardinality, base_width, dilation=1, stride_init=2):
    blocks, stride = [], stride_init
    for _ in range(nblocks):
        blocks.append(ResNeXtBottleneck(inplanes, outplanes, stride, dilation, cardinality, base_width))
        inplanes, stride = outplanes, 1
    return _Seq(*blocks), outplanes


class ResNeXt_body(_Container):
    """VNL.py:547-591."""

    def __init__(self, block_counts, cardinality, base_width, output_stride, freeze_backbone):
        super().__init__()
        self.block_counts = block_counts
        self.convX = len(block_counts) + 1
        self.num_layers = (sum(block_counts) + 3 * (self.convX == 4)) * 3 + 2
        self.freeze_backbone = freeze_backbone
        self.res1 = basic_bn_stem()
        dim_in = 64
        res5_dilate = int(32 / output_stride)
        res5_stride = 2 if res5_dilate == 1 else 1
        res4_dilate = 1 if res5_dilate <= 2 else 2
        res4_stride = 2 if res4_dilate == 1 else 1
        self.res2, dim_in = add_stage(dim_in, 256, block_counts[0], cardinality, base_width, dilation=1, stride_init=1)
        self.res3, dim_in = add_stage(dim_in, 512, block_counts[1], cardinality, base_width, dilation=1, stride_init=2)
        self.res4, dim_in = add_stage(dim_in, 1024, block_counts[2], cardinality, base_width, dilation=res4_dilate, stride_init=res4_stride)
        self.res5, dim_in = add_stage(dim_in, 2048, block_counts[3], cardinality, base_width, dilation=res5_dilate, stride_init=res5_stride)
        self.spatial_scale = 1 / output_stride
        self.dim_out = dim_in
        if freeze_backbone:                       # VNL.py:586-591: the BatchNorm affine parameters only
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    for p in m.parameters():
                        p.requires_grad = False


def ResNeXt50_32x4d_body_stride16(freeze_backbone):
    return ResNeXt_body((3, 4, 6, 3), 32, 4, 16, freeze_backbone)


def ResNeXt101_32x4d_body_stride16(freeze_backbone):
    return ResNeXt_body((3, 4, 23, 3), 32, 4, 16, freeze_backbone)


# ---------------------------------------------------------------------------------------------- MobileNetV2 body (VNL.py:389-537)
def _mbv2_conv_bn(inp, oup, stride):
    return _Seq(nn.Conv2d(inp, oup, 3, stride, 1, bias=False), nn.BatchNorm2d(oup), nn.ReLU6(inplace=True))


class InvertedResidual(_Container):
    """VNL.py:416-457: [1x1 expand -> BN -> ReLU6 ->] depthwise 3x3 (stride, dilation) -> BN -> ReLU6 -> 1x1 -> BN [+ x]; `conv`
    holds the layers under the reference's Sequential indices."""

    def __init__(self, inp, oup, stride, expand_ratio, dilation=1):
        super().__init__()
        assert stride in (1, 2)
        hidden = round(inp * expand_ratio)
        self.stride, self.dilation, self.expand = stride, dilation, expand_ratio != 1
        self.use_res_connect = stride == 1 and inp == oup
        dw = lambda: nn.Conv2d(hidden, hidden, 3, stride, groups=hidden, bias=False, padding=dilation, dilation=dilation)
        tail = [dw(), nn.BatchNorm2d(hidden), nn.ReLU6(inplace=True), nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        head = [nn.Conv2d(inp, hidden, 1, 1, 0, bias=False), nn.BatchNorm2d(hidden), nn.ReLU6(inplace=True)] if self.expand else []
        self.conv = _Seq(*(head + tail))


def _mbv2_stage(setting, inp, dilation=1):
    blocks = []
    for t, c, n, s in setting:
        for i in range(n):
            blocks.append(InvertedResidual(inp, c, s if i == 0 else 1, expand_ratio=t, dilation=dilation))
            inp = c
    return _Seq(*blocks), inp


class MobileNetV2(_Container):
    """VNL.py:471-537 (width multiplier 1): res1 = 3x3 / 2 stem, res2 .. res5 = the inverted-residual stages; at output stride 8
    res4 and res5 keep the 1/8 map and dilate by 2 and 4."""

    def __init__(self, output_stride=32):
        super().__init__()
        self.convX, self.last_channel = 5, 320
        stride1 = 1 if 32 / output_stride == 4 else 2
        stride2 = 1 if 32 / output_stride > 1 else 2
        dilation1 = 1 if stride1 == 2 else 2
        dilation2 = 1 if stride2 == 2 else (2 if stride1 == 2 else 4)
        self.res1 = _Seq(_mbv2_conv_bn(3, 32, 2))
        self.res2, c = _mbv2_stage([[1, 16, 1, 1], [6, 24, 2, 2]], 32)
        self.res3, c = _mbv2_stage([[6, 32, 3, 2]], c)
        self.res4, c = _mbv2_stage([[6, 64, 4, stride1], [6, 96, 3, 1]], c, dilation1)
        self.res5, c = _mbv2_stage([[6, 160, 3, stride2], [6, 320, 1, 1]], c, dilation2)
        for m in self.modules():                          # VNL.py:523-537 _initialize_weights
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


def MobileNetV2_body_stride8(freeze_backbone):
    return MobileNetV2(output_stride=8)


class Global_pool_block(_Container):
    """VNL.py:172-187: 1x1 conv -> BatchNorm (momentum 0.9) -> global average -> `unpool` back to crop_size / output_stride."""

    def __init__(self, dim_in, dim_out, output_stride, crop_size):
        super().__init__()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.globalpool_conv1x1 = nn.Conv2d(dim_in, dim_out, 1, stride=1, padding=0, bias=False)
        self.globalpool = nn.AdaptiveAvgPool2d((1, 1))
        self.globalpool_bn = nn.BatchNorm2d(dim_out, momentum=0.9)
        self.unpool = nn.AdaptiveAvgPool2d((int(crop_size[0] / output_stride), int(crop_size[1] / output_stride)))


def _conv_init(m, init_type, kaiming):
    if init_type == 'xavier':
        nn.init.xavier_normal_(m.weight)
    if init_type == 'kaiming':
        kaiming(m.weight)
    if init_type == 'gaussian':
        nn.init.normal_(m.weight, std=0.01)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0.0)


class lateral(_Container):
    """VNL.py:97-170."""

    def __init__(self, conv_body_func, args):
        super().__init__()
        self.dim_in = args.enc_dim_in[-1:0:-1]
        self.dim_out = args.enc_dim_out
        self.encoder = args.encoder
        self.pretrained = args.pretrained
        self.num_lateral_stages = len(self.dim_in)
        self.topdown_lateral_modules = nn.ModuleList()
        for i in range(self.num_lateral_stages):
            self.topdown_lateral_modules.append(lateral_block(self.dim_in[i], self.dim_out[i]))
        self.bottomup = conv_body_func(args.freeze_backbone)
        dilation_rate = [4, 8, 12] if 'stride_8' in self.encoder else [2, 4, 6]
        encoder_stride = 8 if 'stride8' in self.encoder else 16
        if 'mobilenetv2' in self.encoder:
            self.bottomup_top = Global_pool_block(self.dim_in[0], self.dim_out[0], encoder_stride, args.crop_size)
        else:
            self.bottomup_top = ASPP_block(self.dim_in[0], self.dim_out[0], dilation_rate, encoder_stride)
        if self.pretrained:
            raise NotImplementedError("pretrained=True reads network/pretrained_models/ResNeXt-ImageNet/*.pth (VNL.py:70-95): load such a "
                                      "file through mono_depth_estimation_amd.checkpoint.load_vnl_imagenet_weights instead")
        self._init_weights(args.init_type)

    def _init_weights(self, init_type='xavier'):
        def init_func(m):
            if isinstance(m, nn.Conv2d):
                _conv_init(m, init_type, nn.init.kaiming_normal_)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight.data, 1.0)
                nn.init.constant_(m.bias.data, 0.0)
        # VNL.py:144-153: children that are ModuleLists (the lateral FTB blocks) are skipped and keep torch's default init
        for child in self.children():
            if not isinstance(child, nn.ModuleList):
                child.apply(init_func)


class fcn_topdown_block(_Container):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.afa_block = AFA_block(dim_in)
        self.ftb_block = FTB_block(dim_in, dim_out)


class fcn_last_block(_Container):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.ftb = FTB_block(dim_in, dim_out)


class fcn_topdown_predict(_Container):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.dropout = nn.Dropout2d(0.0)
        self.conv1 = nn.Conv2d(dim_in, dim_out, 3, stride=1, padding=2, dilation=2, bias=True)
        self.softmax = nn.Softmax(dim=1)


class fcn_topdown(_Container):
    """VNL.py:242-294."""

    def __init__(self, args):
        super().__init__()
        self.dim_in = args.dec_dim_in
        self.dim_out = args.dec_dim_out + [args.dec_out_c]
        self.num_fcn_topdown = len(self.dim_in)
        aspp_blocks_num = 1 if 'mobilenetv2' in args.encoder else 5
        self.top = _Seq(
            nn.Conv2d(self.dim_in[0] * aspp_blocks_num, self.dim_in[0], 1, stride=1, padding=0, bias=False),
            nn.BatchNorm2d(self.dim_in[0], 0.5)            # (sic: the second positional argument is eps)
        )
        self.topdown_fcn1 = fcn_topdown_block(self.dim_in[0], self.dim_out[0])
        self.topdown_fcn2 = fcn_topdown_block(self.dim_in[1], self.dim_out[1])
        self.topdown_fcn3 = fcn_topdown_block(self.dim_in[2], self.dim_out[2])
        self.topdown_fcn4 = fcn_topdown_block(self.dim_in[3], self.dim_out[3])
        self.topdown_fcn5 = fcn_last_block(self.dim_in[4], self.dim_out[4])
        self.topdown_predict = fcn_topdown_predict(self.dim_in[5], self.dim_out[5])
        self.init_type = args.init_type
        self._init_weights(self.init_type)

    def _init_weights(self, init_type='xavier'):
        def init_func(m):
            if isinstance(m, nn.Conv2d):
                _conv_init(m, init_type, nn.init.kaiming_normal_)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.normal_(m.weight.data, 0.0, 1.0)   # VNL.py:280
                nn.init.constant_(m.bias.data, 0.0)
        for child in self.children():
            child.apply(init_func)


def get_func(args):
    if args.encoder == 'resnext50_32x4d_body_stride16':
        return lateral(ResNeXt50_32x4d_body_stride16, args)
    if args.encoder == 'resnext101_32x4d_body_stride16':
        return lateral(ResNeXt101_32x4d_body_stride16, args)
    if args.encoder == 'mobilenetv2_body_stride8':
        return lateral(MobileNetV2_body_stride8, args)
    raise ValueError("Unknown bottom up model")


class DepthModel(_Container):
    def __init__(self, args):
        super().__init__()
        self.encoder_modules = get_func(args)
        self.decoder_modules = fcn_topdown(args)


# ---------------------------------------------------------------------------------------------- launch plan
class VNLEngine(G.TapeEngine):
    """The tape of DepthModel.forward (VNL.py:690-693 -> lateral.forward :155-170 -> fcn_topdown.forward :286-294)."""

    def _bottleneck(self, x, blk):
        sc = blk.shortcut
        return self.bottleneck(x, blk.conv1, blk.bn1, blk.conv2, blk.bn2, blk.conv3, blk.bn3,
                               sc.conv if sc is not None else None, sc.bn if sc is not None else None)

    def _ftb(self, x, ftb):
        c1 = self.add(G.Conv(self, x, ftb.conv1.weight, 1)).out                 # also the residual
        t = self.conv_bn(c1, ftb.conv2, ftb.bn1, True)
        c3 = self.add(G.Conv(self, t, ftb.conv3.weight, 3, 1, 2, 2)).out
        return self.pw(c3, r=c1, act="relu")

    def _afa(self, afa, lat, top):
        C = lat.C
        pooled = self.buf(lat.N, 1, 1, 2 * C)                                   # cat([lateral, top]) after the pooling
        self.add(G.GlobalAvgPool(self, lat, out=pooled.slice(0, C)))
        self.add(G.GlobalAvgPool(self, top, out=pooled.slice(C, C)))
        h = self.add(G.Conv(self, pooled, afa.conv1.weight, 1)).out
        h = self.pw(h, act="relu")
        w = self.add(G.Conv(self, h, afa.conv2.weight, 1)).out
        w = self.pw(w, act="sigmoid")
        return self.add(G.Gate(self, w, lat, top)).out

    def _bn_relu6(self, c, bn, site=None, stats_pass=False):
        """BatchNorm -> ReLU6 (VNL.py:402,410,433,441): the BatchNorm pass without activation, then the clamp as a pointwise pass."""
        site = site if site is not None else self._site([bn])
        if stats_pass:
            self.add(G.StatsPass(self, c, site))
        return self.pw(self.add(G.BN(self, c, site, False)).out, act="relu6")

    def _inverted_residual(self, x, blk):
        L = list(blk.conv)
        h = x
        if blk.expand:
            site = self._site([L[1]])
            h = self._bn_relu6(self.add(G.Conv(self, h, L[0].weight, 1, site=site)).out, L[1], site=site)
            L = L[3:]
        h = self.add(G.DwConv(self, h, L[0].weight, blk.stride, blk.dilation)).out
        h = self._bn_relu6(h, L[1], stats_pass=True)                          # (no conv epilogue behind a depthwise pass: a reduction)
        h = self.conv_bn(h, L[3], L[4], False)
        return self.pw(h, r=x) if blk.use_res_connect else h

    def _plan(self):
        m, N, H, W = self.m.depth_model, self.N, self.H, self.W
        enc, dec = m.encoder_modules, m.decoder_modules
        body = enc.bottomup
        mobile = isinstance(body, MobileNetV2)
        if mobile:
            conv, bn = body.res1[0][0], body.res1[0][1]
            site = self._site([bn])
            self.stem = self.add(G.ImageStem(self, conv, site, N, H, W))
            x, feats = self._bn_relu6(self.stem.out, bn, site=site), []
        else:
            self.stem = self.add(G.Stem(self, body.res1.conv1, body.res1.bn1, N, H, W))
            x, feats = self.stem.out, []
        for i in range(2, body.convX + 1):
            for blk in getattr(body, "res%d" % i):
                x = self._inverted_residual(x, blk) if mobile else self._bottleneck(x, blk)
            feats.append(x)                                                     # res2 .. res5
        if mobile:
            self._plan_decoder(feats, self._global_pool_top(feats[-1], enc.bottomup_top))
            return
        # ASPP (VNL.py:211-228): five branches written side by side into one tensor
        top5, aspp = feats[-1], enc.bottomup_top
        Co = aspp.dim_out
        cat = self.buf(N, top5.H, top5.W, 5 * Co)
        self.conv_bn(top5, aspp.aspp_conv1x1, aspp.aspp_bn1x1, False, out=cat.slice(0, Co))
        for i in range(3):
            self.conv_bn(top5, getattr(aspp, "aspp_conv3_%d" % (i + 1)), getattr(aspp, "aspp_bn3_%d" % (i + 1)), False,
                          out=cat.slice((i + 1) * Co, Co))
        v = self.add(G.GlobalAvgPool(self, top5)).out
        u = self.conv_bn(v, aspp.globalpool_conv1x1, aspp.globalpool_bn, False)
        self.add(G.Broadcast(self, u, cat.slice(4 * Co, Co)))
        self._plan_decoder(feats, cat)

    def _global_pool_top(self, top5, gp):
        """Global_pool_block.forward (VNL.py:181-186): conv1x1 -> BN -> global average -> `unpool` of the 1 x 1 map = a broadcast."""
        c = self.conv_bn(top5, gp.globalpool_conv1x1, gp.globalpool_bn, False)
        v = self.add(G.GlobalAvgPool(self, c)).out
        h, w = gp.unpool.output_size
        out = self.buf(top5.N, h, w, gp.dim_out)
        self.add(G.Broadcast(self, v, out))
        return out

    def _plan_decoder(self, feats, cat):
        m, N, H, W = self.m.depth_model, self.N, self.H, self.W
        enc, dec = m.encoder_modules, m.decoder_modules
        laterals = [cat]
        for i in range(enc.num_lateral_stages):
            laterals.append(self._ftb(feats[-(i + 1)], enc.topdown_lateral_modules[i].lateral))
        # decoder (VNL.py:286-294)
        x = self.conv_bn(laterals[0], dec.top[0], dec.top[1], False)
        for i in range(1, 5):
            blk, lat = getattr(dec, "topdown_fcn%d" % i), laterals[i]
            if (lat.H, lat.W, lat.C) != (x.H, x.W, x.C):
                x = self.add(G.Resize(self, x, lat.H, lat.W, True)).out
            x = self._ftb(self._afa(blk.afa_block, lat, x), blk.ftb_block)
        half = (math.ceil(H / 2.0), math.ceil(W / 2.0))
        x = self.add(G.Resize(self, x, half[0], half[1], True)).out
        x = self._ftb(x, dec.topdown_fcn5.ftb)
        x = self.add(G.Resize(self, x, H, W, True)).out
        pred = dec.topdown_predict
        c = self.add(G.Conv(self, x, pred.conv1.weight, 3, 1, 2, 2)).out
        self.heads = [self.add(G.SoftmaxHead(self, c, pred.conv1.bias, pred.dim_out))]


class MetricDepthModel(G.TapeModule):
    """reference VNL.py:672-682."""

    _engine_cls = VNLEngine

    def __init__(self, args):
        super(MetricDepthModel, self).__init__()
        self.loss_names = ['Weighted_Cross_Entropy', 'Virtual_Normal']
        self.depth_model = DepthModel(args)
        self._init_runtime()

    def _make_store(self, device):
        raw = [n for n, p in self.named_parameters() if p.dim() == 4 and (n.endswith("res1.conv1.weight") or ".conv2.weight" in n and "bottomup.res" in n)]
        # (+ MobileNetV2's depthwise weights [C][1][3][3]: the fp32 master is csrc/dwconv.hip's [C][9] operand as it stands)
        raw += [n + ".weight" for n, mod in self.named_modules() if isinstance(mod, nn.Conv2d) and mod.groups > 1 and mod.groups == mod.in_channels]
        return G.NetStore(self, device, is_encoder=lambda n: 'res' in n, raw=raw)     # vnl.py:298-305: 'res' in key -> encoder LR

    def forward(self, x):
        self.a_real = x
        self.b_fake_logit, self.b_fake_softmax = self._run(x)
        return self.b_fake_logit, self.b_fake_softmax
