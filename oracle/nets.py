"""CPU fp32 oracle of the VNL / MiDaS / BTS / Eigen networks — TEST INFRASTRUCTURE ONLY.

FUNCTIONAL restatements over a state dict: `P` maps the reference's own state_dict keys to tensors (parameters that
require grad for gradient parity, BN running statistics updated in place in training mode), and every function walks the
keys the way the reference's forward walks its modules.  One dict therefore loads into the reference model, drives this
oracle and loads into the HIP module.  Cited lines are under /root/reference.

Pinned by tests/golden/{vnl_net,midas_net,bts_net,eigen,dorn_net,mynet}.npz, minted by tests/golden/gen_golden.py from the reference's own
classes (imported with stand-ins for the absent torchvision / torch.hub trunks, whose architecture is restated from
their public definitions — see that script's docstring).
"""
import math

import torch
import torch.nn.functional as F


class Net:
    """State-dict walker: P[key] lookups under a prefix stack, train/eval BatchNorm, optional bf16 emulation of the HIP
    path's storage roundings (`q`: applied where that path materialises a bf16 tensor)."""

    def __init__(self, P, train, q=None, momentum=None):
        self.P, self.train, self.q, self.momentum = P, train, (q or (lambda t: t)), momentum

    def conv(self, x, key, stride=1, pad=0, dil=1, groups=1, bias=True):
        """(16-bit emulation: the conv result is stored once, WITH its bias -- the HIP path adds it to the fp32 accumulator in
        the conv launch's epilogue, include/mde_hip.h: mde_conv_gemm_act; a conv bias in front of a BatchNorm is dropped by
        the caller's plan and cancels here.)"""
        b = self.P.get(key + ".bias") if bias else None
        y = F.conv2d(x, self.P[key + ".weight"], None, stride, pad, dil, groups)
        return self.q(y if b is None else y + b.view(1, -1, 1, 1))

    def bn(self, x, key, momentum=0.1, eps=1e-5):
        P = self.P
        return F.batch_norm(x, P[key + ".running_mean"], P[key + ".running_var"], P[key + ".weight"], P[key + ".bias"],
                            self.train, momentum if self.momentum is None else self.momentum, eps)


# ---------------------------------------------------------------------------------------------- VNL (network/VNL.py)
def _vnl_ftb(n, x, k):
    """FTB_block.forward (VNL.py:341-350)."""
    r = n.conv(x, k + ".conv1")
    y = n.q(F.relu(n.bn(n.conv(r, k + ".conv2", pad=2, dil=2), k + ".bn1", momentum=0.5)))
    return n.q(F.relu(n.conv(y, k + ".conv3", pad=2, dil=2) + r))


def _vnl_afa(n, lat, top, k):
    """AFA_block.forward (VNL.py:365-373)."""
    w = n.q(torch.cat([lat, top], 1).mean((2, 3), keepdim=True))
    w = n.q(torch.sigmoid(n.conv(n.q(F.relu(n.conv(w, k + ".conv1"))), k + ".conv2")))
    return n.q(w * lat + top)


def _vnl_bottleneck(n, x, k, stride, dil):
    """ResNeXtBottleneck.forward (VNL.py:653-669); cardinality 32."""
    y = n.q(F.relu(n.bn(n.conv(x, k + ".conv1"), k + ".bn1")))
    y = n.q(F.relu(n.bn(n.conv(y, k + ".conv2", stride, dil, dil, 32), k + ".bn2")))
    y = n.bn(n.conv(y, k + ".conv3"), k + ".bn3")
    if k + ".shortcut.conv.weight" in n.P:
        x = n.bn(n.conv(x, k + ".shortcut.conv", stride), k + ".shortcut.bn")
    return n.q(F.relu(y + x))


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rounding_draw(k, to=torch.bfloat16):
    """The k-th REALISATION of storage rounding: round on the grid scaled by 1 + k 2^-12 (k = 0: the plain grid).  One
    realisation of the rounding noise says little about a mean over pixels (the noise of coarse feature maps is coherent over
    whole image regions): tests and tools/rounding_draws.py look at several."""
    s = 1.0 + k * 2.0 ** -12
    return lambda t: (t * s).to(to).to(torch.float32) / s


def fp16_round(t):
    """Storage rounding of the reference's own mixed-precision runs (train.py:139-140: precision=16, amp_level='O2'): 11 bits
    of significand against bf16's 8, overflow to +-inf above 65504 and gradual underflow below 6e-5 as fp16 has them."""
    return t.to(torch.float16).to(torch.float32)


def _mbv2_features(n, x, b):
    """MobileNetV2.forward at output stride 8 (VNL.py:471-521; InvertedResidual :416-457): -> [res2 .. res5]."""
    relu6 = lambda t: n.q(F.relu6(t))
    y = relu6(n.q(n.bn(n.conv(x, b + "res1.0.0", 2, 1), b + "res1.0.1")))
    feats = []
    stages = {2: ([[1, 16, 1, 1], [6, 24, 2, 2]], 1), 3: ([[6, 32, 3, 2]], 1), 4: ([[6, 64, 4, 1], [6, 96, 3, 1]], 2),
              5: ([[6, 160, 3, 1], [6, 320, 1, 1]], 4)}
    for stage in (2, 3, 4, 5):
        setting, dil = stages[stage]
        idx = 0
        for t, c, cnt, s in setting:
            for i in range(cnt):
                k = b + "res%d.%d.conv." % (stage, idx)
                stride, inp = (s if i == 0 else 1), y
                o = 0
                if t != 1:
                    y = relu6(n.q(n.bn(n.conv(y, k + "0"), k + "1")))
                    o = 3
                ch = y.shape[1]
                y = relu6(n.q(n.bn(n.conv(y, k + "%d" % o, stride, dil, dil, groups=ch), k + "%d" % (o + 1))))
                y = n.q(n.bn(n.conv(y, k + "%d" % (o + 3)), k + "%d" % (o + 4)))
                if stride == 1 and inp.shape[1] == y.shape[1]:
                    y = n.q(y + inp)
                idx += 1
        feats.append(y)
    return feats


def vnl_forward(P, x, train, block_counts=(3, 4, 6, 3), momentum=None, q=None, crop_size=None):
    """MetricDepthModel.forward (VNL.py:678-693) for the resnext*_32x4d_body_stride16 encoders and for mobilenetv2_body_stride8
    (told apart by the state dict's keys; crop_size: the size Global_pool_block's `unpool` was built for) -> (logits, softmax).
    momentum: override every BatchNorm's (1.0 = "running statistics := this batch's", weights.calibrate_running_stats).
    q: rounding applied wherever the HIP path stores a bf16 tensor (bf16_round: the bf16-emulating oracle)."""
    n = Net(P, train, q=q, momentum=momentum)
    e, d = "depth_model.encoder_modules.", "depth_model.decoder_modules."
    H, W = x.shape[2:]
    b = e + "bottomup."
    if b + "res1.0.0.weight" in P:
        feats = _mbv2_features(n, x, b)
        a = e + "bottomup_top."
        g = n.q(n.bn(n.conv(feats[-1], a + "globalpool_conv1x1"), a + "globalpool_bn", 0.9)).mean((2, 3), keepdim=True)
        cs = crop_size if crop_size is not None else (H, W)
        lats = [n.q(g).expand(-1, -1, int(cs[0] / 8), int(cs[1] / 8))]
        return _vnl_decoder(n, feats, lats, e, d, H, W)
    y = n.q(F.relu(n.bn(n.conv(x, b + "res1.conv1", 2, 3), b + "res1.bn1")))
    y = F.max_pool2d(y, 3, 2, 1)
    feats = []
    # output stride 16: res3 and res4 open with stride 2, res5 keeps the size and dilates by 2 (VNL.py:557-569)
    for stage, (cnt, stride, dil) in enumerate(zip(block_counts, (1, 2, 2, 1), (1, 1, 1, 2))):
        for i in range(cnt):
            y = _vnl_bottleneck(n, y, b + "res%d.%d" % (stage + 2, i), stride if i == 0 else 1, dil)
        feats.append(y)
    # ASPP_block.forward (VNL.py:211-228)
    a, t = e + "bottomup_top.", feats[-1]
    xs = [n.q(n.bn(n.conv(t, a + "aspp_conv1x1"), a + "aspp_bn1x1", 0.5))]
    for i, r in enumerate((2, 4, 6)):
        xs.append(n.q(n.bn(n.conv(t, a + "aspp_conv3_%d" % (i + 1), pad=r, dil=r), a + "aspp_bn3_%d" % (i + 1), 0.5)))
    g = n.q(n.bn(n.conv(n.q(t.mean((2, 3), keepdim=True)), a + "globalpool_conv1x1"), a + "globalpool_bn", 0.5))
    xs.append(g.expand(-1, -1, t.shape[2], t.shape[3]))          # bilinear(align_corners) of a 1x1 map is a broadcast
    return _vnl_decoder(n, feats, [torch.cat(xs, 1)], e, d, H, W)


def _vnl_decoder(n, feats, lats, e, d, H, W):
    """The lateral FTB blocks (lateral.forward, VNL.py:163-170) and fcn_topdown.forward (VNL.py:286-294)."""
    for i in range(4):
        lats.append(_vnl_ftb(n, feats[-(i + 1)], e + "topdown_lateral_modules.%d.lateral" % i))
    # fcn_topdown.forward (VNL.py:286-294); `top`'s BatchNorm2d(dim, 0.5) has eps = 0.5 (VNL.py:253)
    y = n.q(n.bn(n.conv(lats[0], d + "top.0"), d + "top.1", eps=0.5))
    for i in range(1, 5):
        lat = lats[i]
        if lat.shape != y.shape:
            y = n.q(F.interpolate(y, size=lat.shape[2:], mode="bilinear", align_corners=True))
        y = _vnl_ftb(n, _vnl_afa(n, lat, y, d + "topdown_fcn%d.afa_block" % i), d + "topdown_fcn%d.ftb_block" % i)
    y = n.q(F.interpolate(y, size=(math.ceil(H / 2.0), math.ceil(W / 2.0)), mode="bilinear", align_corners=True))
    y = _vnl_ftb(n, y, d + "topdown_fcn5.ftb")
    y = n.q(F.interpolate(y, size=(H, W), mode="bilinear", align_corners=True))
    logit = n.conv(y, d + "topdown_predict.conv1", pad=2, dil=2)
    return logit, torch.softmax(logit, 1)


def vnl_params(depth_max=1.1, depth_min=0.01, dec_out_c=150, encoder="resnext50_32x4d_body_stride16"):
    """The attribute bag modules/vnl.py:143-163 builds for VNL.MetricDepthModel / criteria.ModelLoss (defaults :327-346)."""
    import types
    import numpy as np
    p = types.SimpleNamespace()
    p.depth_min, p.encoder, p.pretrained, p.freeze_backbone, p.init_type = depth_min, encoder, 0, False, "xavier"
    p.enc_dim_in, p.enc_dim_out = [64, 256, 512, 1024, 2048], [512, 256, 256, 256]
    p.dec_dim_in, p.dec_dim_out, p.dec_out_c = [512, 256, 256, 256, 256, 256], [256, 256, 256, 256, 256], dec_out_c
    p.focal_x, p.focal_y, p.crop_size, p.diff_loss_weight = 519.0, 519.0, (385, 385), 6
    p.depth_min_log = np.log10(depth_min)
    p.depth_bin_interval = (np.log10(depth_max) - np.log10(depth_min)) / dec_out_c
    p.wce_loss_weight = [[np.exp(-0.2 * (i - j) ** 2) for i in range(dec_out_c)] for j in np.arange(dec_out_c)]
    p.depth_bin_border = np.array([np.log10(depth_min) + p.depth_bin_interval * (i + 0.5) for i in range(dec_out_c)])
    return p


def vnl_mobilenet_params(crop_size, **kw):
    """vnl_params for `--encoder mobilenetv2_body_stride8`: the channel lists the VNL authors' MobileNetV2 configuration
    uses (the reference's argparse defaults, VNL.py:702-705, are the ResNeXt ones; its MobileNetV2 widths are
    VNL.py:489-519's) and the crop size Global_pool_block's `unpool` is built for (VNL.py:180, :707)."""
    p = vnl_params(encoder="mobilenetv2_body_stride8", **kw)
    p.enc_dim_in, p.enc_dim_out = [32, 24, 32, 96, 320], [128, 64, 64, 64]
    p.dec_dim_in, p.dec_dim_out = [128, 64, 64, 64, 64, 64], [64, 64, 64, 64, 64]
    p.crop_size = tuple(crop_size)
    return p


# ---------------------------------------------------------------------------------------------- MiDaS (network/MiDaS.py)
def _tv_bottleneck(n, x, k, stride, groups):
    """torchvision Bottleneck (v1.5) as MiDaS.py:93-104 wires it in; downsample present on a stage's first block."""
    y = n.q(F.relu(n.bn(n.conv(x, k + ".conv1"), k + ".bn1")))
    y = n.q(F.relu(n.bn(n.conv(y, k + ".conv2", stride, 1, 1, groups), k + ".bn2")))
    y = n.bn(n.conv(y, k + ".conv3"), k + ".bn3")
    if k + ".downsample.0.weight" in n.P:
        x = n.bn(n.conv(x, k + ".downsample.0", stride), k + ".downsample.1")
    return n.q(F.relu(y + x))


def _midas_rcu(n, a, k):
    """ResidualConvUnit.forward (MiDaS.py:188-201) on a = relu(x): the unit's first ReLU is IN PLACE, so the skip it adds
    back is relu(x), not x."""
    y = n.q(F.relu(n.conv(a, k + ".conv1", pad=1)))
    return n.q(n.conv(y, k + ".conv2", pad=1) + a)


def _midas_ffb(n, k, x0, x1=None):
    """FeatureFusionBlock.forward (MiDaS.py:215-229)."""
    out = x0
    if x1 is not None:
        out = out + _midas_rcu(n, n.q(F.relu(x1)), k + ".resConfUnit1")
    out = _midas_rcu(n, n.q(F.relu(out)), k + ".resConfUnit2")
    return n.q(F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True))


def midas_forward(P, x, train, blocks=(3, 4, 23, 3), groups=32, momentum=None, q=None):
    """MidasNet.forward (MiDaS.py:59-87) over a ResNeXt-101 32x8d trunk -> N x 7 x H x W sigmoid maps."""
    n = Net(P, train, q=q, momentum=momentum)
    p, s = "pretrained.", "scratch."
    y = n.q(F.relu(n.bn(n.conv(x, p + "layer1.0", 2, 3), p + "layer1.1")))
    y = F.max_pool2d(y, 3, 2, 1)
    feats = []
    for li, cnt in enumerate(blocks):
        for i in range(cnt):
            k = p + ("layer1.4.%d" % i if li == 0 else "layer%d.%d" % (li + 1, i))
            y = _tv_bottleneck(n, y, k, 2 if (li > 0 and i == 0) else 1, groups)
        feats.append(y)
    rn = [n.conv(f, s + "layer%d_rn" % (i + 1), pad=1) for i, f in enumerate(feats)]
    path = _midas_ffb(n, s + "refinenet4", rn[3])
    path = _midas_ffb(n, s + "refinenet3", path, rn[2])
    path = _midas_ffb(n, s + "refinenet2", path, rn[1])
    path = _midas_ffb(n, s + "refinenet1", path, rn[0])
    o = s + "output_conv."
    y = n.q(n.conv(path, o + "0", pad=1))
    y = n.q(F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=False))
    y = n.q(F.relu(n.conv(y, o + "2", pad=1)))
    return torch.sigmoid(n.conv(y, o + "4"))


# ---------------------------------------------------------------------------------------------- BTS (network/Bts.py)
def _bts_upconv(n, x, k, bn=None):
    """upconv.forward (Bts.py:76-80): nearest x2 -> 3x3 -> ELU [-> the BatchNorm `bn(t)` behind it, Bts.py:216-229].
    16-bit emulation: in eval mode the HIP path stores the conv's PRE-activation and applies ELU and the BatchNorm's affine in one
    fp32 pass (mde_bn_apply, relu = 2); in training mode the ELU output is stored (the statistics pass reads it)."""
    z = n.conv(n.q(F.interpolate(x, scale_factor=2, mode="nearest")), k + ".conv", pad=1)
    if bn is None:
        return n.q(F.elu(z))
    return n.q(bn(F.elu(z) if not n.train else n.q(F.elu(z))))


def _bts_atrous(n, x, k, dil, bn_first=True):
    """atrous_conv.forward (Bts.py:51-66): [BN(eps 1.1e-5)] -> ReLU -> 1x1 -> BN -> ReLU -> dilated 3x3; momentum 0.01."""
    k = k + ".atrous_conv."
    if bn_first:
        x = n.bn(x, k + "first_bn", 0.01, 1.1e-5)
    x = n.q(F.relu(x))
    x = n.q(F.relu(n.bn(n.conv(x, k + "aconv_sequence.1"), k + "aconv_sequence.2", 0.01)))
    return n.conv(x, k + "aconv_sequence.4", pad=dil, dil=dil)


def _bts_reduc(n, x, k, max_depth, final=False):
    """reduction_1x1.forward (Bts.py:83-122): 1x1 + ELU halvings down to 8 channels, then 3 plane parameters -> (n1, n2, n3,
    distance), or (final) one sigmoid channel."""
    keys = sorted({kk[len(k) + 7:].split(".")[0] for kk in n.P if kk.startswith(k + ".reduc.inter_")}, key=lambda t: (-int(t.split("_")[1]), -int(t.split("_")[2])))
    for name in keys:
        x = n.q(F.elu(n.conv(x, k + ".reduc." + name + ".0")))
    if final:
        return torch.sigmoid(n.conv(x, k + ".reduc.final.0"))
    return _plane_from_params(n.q(n.conv(x, k + ".reduc.plane_params")), max_depth)


def _plane_from_params(x, max_depth):
    """Bts.py:112-120: three conv channels -> (n1, n2, n3, distance)."""
    theta = torch.sigmoid(x[:, 0]) * math.pi / 3
    phi = torch.sigmoid(x[:, 1]) * math.pi * 2
    dist = torch.sigmoid(x[:, 2]) * max_depth
    return torch.stack([torch.sin(theta) * torch.cos(phi), torch.sin(theta) * torch.sin(phi), torch.cos(theta), dist], 1)


def _bts_lpg(plane_eq, up):
    """local_planar_guidance.forward (Bts.py:132-146) -> N x H*up x W*up."""
    e = plane_eq.repeat_interleave(up, 2).repeat_interleave(up, 3)
    N, _, H, W = e.shape
    u = (torch.arange(W, dtype=torch.float32) % up - (up - 1) * 0.5) / up
    v = (torch.arange(H, dtype=torch.float32) % up - (up - 1) * 0.5) / up
    return e[:, 3] / (e[:, 0] * u.view(1, 1, W) + e[:, 1] * v.view(1, H, 1) + e[:, 2])


def _bts_plane_depth(n, feat, k, up, max_depth):
    r = _bts_reduc(n, feat, k, max_depth)
    eq = torch.cat([F.normalize(r[:, :3], 2, 1), r[:, 3:4]], 1)
    return _bts_lpg(eq, up).unsqueeze(1) / max_depth


def densenet_features(n, x, pfx, blocks=(6, 12, 36, 24)):
    """torchvision densenet161 `.features` walked the way Bts.py:309-321 does -> [relu0, pool0, transition1, transition2, norm5]."""
    skips = []
    y = n.q(F.relu(n.bn(n.conv(x, pfx + "conv0", 2, 3), pfx + "norm0")))
    skips.append(y)
    y = F.max_pool2d(y, 3, 2, 1)
    skips.append(y)
    for b, cnt in enumerate(blocks):
        for i in range(cnt):
            k = pfx + "denseblock%d.denselayer%d." % (b + 1, i + 1)
            t = n.q(F.relu(n.bn(y, k + "norm1")))
            t = n.q(F.relu(n.bn(n.conv(t, k + "conv1"), k + "norm2")))
            y = torch.cat([y, n.conv(t, k + "conv2", pad=1)], 1)
        if b < len(blocks) - 1:
            k = pfx + "transition%d." % (b + 1)
            y = n.q(F.avg_pool2d(n.conv(n.q(F.relu(n.bn(y, k + "norm"))), k + "conv"), 2, 2))
            if b < 2:
                skips.append(y)
    skips.append(n.bn(y, pfx + "norm5"))
    return skips


def resnet_features(n, x, pfx):
    """encoder.forward (Bts.py:309-321) over a torchvision ResNet / ResNeXt kept whole as `base_model` (Bts.py:293-307): the walk
    over `_modules` collects 'relu' (after conv1 / bn1) and layer1 .. layer4; block counts and the group count are read off
    the state dict."""
    P = n.P
    y = n.q(F.relu(n.bn(n.conv(x, pfx + "conv1", 2, 3), pfx + "bn1")))
    skips = [y]
    y = F.max_pool2d(y, 3, 2, 1)
    for L in range(1, 5):
        i = 0
        while pfx + "layer%d.%d.conv1.weight" % (L, i) in P:
            k = pfx + "layer%d.%d" % (L, i)
            w2 = P[k + ".conv2.weight"]
            y = _tv_bottleneck(n, y, k, 2 if (i == 0 and L > 1) else 1, w2.shape[0] // w2.shape[1])
            i += 1
        skips.append(y)
    return skips


def bts_forward(P, x, train, max_depth=10.0, momentum=None, q=None, image_residuals=False):
    """BtsModel.forward (Bts.py:324-333) with a densenet*_bts or a resnet*_bts / resnext*_bts encoder (told apart by the state
    dict's keys) -> the 5-tuple of bts.forward (Bts.py:205-278); dataset 'nyu', no image residuals."""
    n = Net(P, train, q=q, momentum=momentum)
    if "encoder.base_model.conv1.weight" in P:
        s0, s1, s2, s3, dense = resnet_features(n, x, "encoder.base_model.")
    else:
        s0, s1, s2, s3, dense = densenet_features(n, x, "encoder.base_model.")
    d = "decoder."
    bnk = lambda t, k: n.bn(t, d + k, 0.01, 1.1e-5)
    dense = n.q(F.relu(dense))
    up5 = _bts_upconv(n, dense, d + "upconv5", lambda t: bnk(t, "bn5"))
    i5 = n.q(F.elu(n.conv(torch.cat([up5, s3], 1), d + "conv5.0", pad=1)))
    up4 = _bts_upconv(n, i5, d + "upconv4", lambda t: bnk(t, "bn4"))
    cat4 = torch.cat([up4, s2], 1)
    z4 = F.elu(n.conv(cat4, d + "conv4.0", pad=1))
    i4 = n.q(bnk(z4 if not train else n.q(z4), "bn4_2"))
    d3 = _bts_atrous(n, i4, d + "daspp_3", 3, bn_first=False)
    c = torch.cat([cat4, d3], 1)
    d6 = _bts_atrous(n, c, d + "daspp_6", 6)
    c = torch.cat([c, d6], 1)
    d12 = _bts_atrous(n, c, d + "daspp_12", 12)
    c = torch.cat([c, d12], 1)
    d18 = _bts_atrous(n, c, d + "daspp_18", 18)
    c = torch.cat([c, d18], 1)
    d24 = _bts_atrous(n, c, d + "daspp_24", 24)
    feat = n.q(F.elu(n.conv(torch.cat([i4, d3, d6, d12, d18, d24], 1), d + "daspp_conv.0", pad=1)))
    d8 = _bts_plane_depth(n, feat, d + "reduc8x8", 8, max_depth)
    up3 = _bts_upconv(n, feat, d + "upconv3", lambda t: bnk(t, "bn3"))
    i3 = n.q(F.elu(n.conv(torch.cat([up3, s1, F.interpolate(d8, scale_factor=0.25, mode="nearest")], 1), d + "conv3.0", pad=1)))
    d4 = _bts_plane_depth(n, i3, d + "reduc4x4", 4, max_depth)
    up2 = _bts_upconv(n, i3, d + "upconv2", lambda t: bnk(t, "bn2"))
    i2 = n.q(F.elu(n.conv(torch.cat([up2, s0, F.interpolate(d4, scale_factor=0.5, mode="nearest")], 1), d + "conv2.0", pad=1)))
    d2 = _bts_plane_depth(n, i2, d + "reduc2x2", 2, max_depth)
    up1 = _bts_upconv(n, i2, d + "upconv1")
    r1 = _bts_reduc(n, up1, d + "reduc1x1", max_depth, final=True)
    i1 = n.q(F.elu(n.conv(torch.cat([up1, r1, d2, d4, d8], 1), d + "conv1.0", pad=1)))
    depth = torch.sigmoid(n.conv(i1, d + "get_depth.0", pad=1))
    if image_residuals and depth.shape[1] == 10:
        # Bts.py:264-271: the colour channels of the two RGBA layers are residuals on the input image; no max_depth factor here
        mean = x.mean(dim=1)
        front = torch.clamp(depth[:, :3] * 2.0 - 1.0 + x, 0.0, 1.0)
        back = torch.clamp(depth[:, 4:7] * 2.0 - 1.0 + x, 0.0, 1.0)
        fronta = torch.clamp(depth[:, 3] * 2.0 - 1.0 + mean, 0.0, 1.0).unsqueeze(1)
        backa = torch.clamp(depth[:, 7] * 2.0 - 1.0 + mean, 0.0, 1.0).unsqueeze(1)
        final = torch.cat([front, fronta, back, backa, depth[:, 8:]], dim=1)
    else:
        final = max_depth * depth
    return d8, d4, d2, r1, final


# ---------------------------------------------------------------------------------------------- Eigen (network/Eigen.py)
def eigen_forward(P, img, train, momentum=None, q=None):
    """Eigen.forward (Eigen.py:14-18): VGG-19-BN coarse net -> two Linear layers -> 3x3/4 transposed conv (scale 1, :81-89),
    the 9x9/2 + 5x5 stack of scale 2 (:36-43) and of scale 3 (:62-69).  img: N x 3 x 240 x 320 -> N x 1 x 109 x 149 (the two
    Linear layers fix the input size: Eigen.py:77-78, 512 * 10 * 7 features).  q: the storage-rounding hook (applied where the
    HIP plan of network/Eigen.py stores a 16-bit tensor: every conv / Linear / transposed-conv result with its bias and
    activation, every BatchNorm + ReLU output; the fp32 head output is not rounded)."""
    n = Net(P, train, q=q, momentum=momentum)
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    x, i = img, 0
    for v in cfg:
        k = "scale1.feature_extractor."
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            i += 1
        else:
            x = n.q(F.relu(n.bn(n.conv(x, k + "%d" % i, pad=1), k + "%d" % (i + 1))))
            i += 3
    x = x.flatten(1)
    x = n.q(F.linear(x, P["scale1.mlp1.weight"], P["scale1.mlp1.bias"]))
    x = n.q(F.linear(x, P["scale1.mlp2.weight"], P["scale1.mlp2.bias"])).reshape(-1, 64, 14, 19)
    x0 = n.q(F.conv_transpose2d(x, P["scale1.upsample.weight"], P["scale1.upsample.bias"], stride=4))
    # Scale2.forward
    y = F.max_pool2d(n.q(F.relu(n.conv(img, "scale2.conv", 2))), 3, 2)[:, :, 1:-1, 1:-1]
    y = torch.cat([y, x0], 1)
    for j in (0, 2, 4):
        y = n.q(F.relu(n.conv(y, "scale2.scale2_onestack.%d" % j, pad=2)))
    x1 = n.q(F.conv_transpose2d(y, P["scale2.scale2_onestack.6.weight"], P["scale2.scale2_onestack.6.bias"], stride=2, padding=2))
    # Scale3.forward
    z = F.max_pool2d(n.q(F.relu(n.conv(img, "scale3.conv", 2)))[:, :, 2:-3, 2:-3], 3, 1)
    z = torch.cat([z, x1], 1)
    for j in (0, 2, 4):
        z = n.q(F.relu(n.conv(z, "scale3.scale3_onestack.%d" % j, pad=2)))
    qs, n.q = n.q, (lambda t: t)                               # the head: fp32 N x 1 x H x W output, no storage rounding
    z = F.relu(n.conv(z, "scale3.scale3_onestack.6", pad=2))
    n.q = qs
    return z


# ---------------------------------------------------------------------------------------------- DORN (network/Dorn.py)
def ordinal_layer(x):
    """OrdinalRegressionLayer.forward (Dorn.py:288-318) -> (decode_c, ord_c1)."""
    N, C, H, W = x.shape
    K = C // 2
    A = x[:, ::2].reshape(N, 1, K * H * W)
    B = x[:, 1::2].reshape(N, 1, K * H * W)
    c = torch.clamp(torch.cat((A, B), dim=1), min=1e-8, max=1e4)
    p1 = F.softmax(c, dim=1)[:, 1].reshape(-1, K, H, W)
    return torch.sum(p1 > 0.5, dim=1).view(-1, 1, H, W), p1


def dorn_forward(P, x, train, size, kernel_size=16, pyramid=(4, 8, 12), dropout=0.5, blocks=(3, 4, 23, 3), masks=None, momentum=None,
                 q=None, return_logits=False):
    """DORN.forward (Dorn.py:340-344): ResNet.forward (:264-273, Bottleneck :152-175) -> SceneUnderstandingModule.forward
    (:110-124, FullImageEncoder :66-80) -> OrdinalRegressionLayer.  The three nn.Dropout2d of the scene module draw from
    torch's global generator in the reference's order (encoder, concat_process.0, concat_process.2) when `masks` is None;
    otherwise masks[i] is the [N][C] scale (0 or 1 / (1 - p)) to apply — how the GPU tests hand over the HIP run's draw."""
    n = Net(P, train, q, momentum)
    b = "backbone.backbone."
    y = x
    for i, st in ((1, 2), (2, 1), (3, 1)):
        y = n.q(F.relu(n.bn(n.conv(y, b + "conv%d" % i, st, 1), b + "bn%d" % i)))
    y = F.max_pool2d(y, 3, 2, 1, ceil_mode=True)
    for li, (nb, stride, dil) in enumerate(zip(blocks, (1, 2, 1, 1), (1, 1, 2, 4))):
        for bi in range(nb):
            k, st = b + "layer%d.%d" % (li + 1, bi), (stride if bi == 0 else 1)
            a = n.q(F.relu(n.bn(n.conv(y, k + ".conv1"), k + ".bn1")))
            a = n.q(F.relu(n.bn(n.conv(a, k + ".conv2", st, dil, dil), k + ".bn2")))
            a = n.bn(n.conv(a, k + ".conv3"), k + ".bn3")
            r = n.bn(n.conv(y, k + ".downsample.0", st), k + ".downsample.1") if k + ".downsample.0.weight" in P else y
            y = n.q(F.relu(a + r))
    drops = [0]

    def drop(t):
        i = drops[0]
        drops[0] += 1
        if masks is not None:
            return t * masks[i].view(t.shape[0], t.shape[1], 1, 1) if train else t
        return F.dropout2d(t, dropout, train)

    def cbr(t, k, pad=0, dil=1):
        if k + ".1.weight" in P:                                                   # conv -> BN -> ReLU
            return n.q(F.relu(n.bn(n.conv(t, k + ".0", 1, pad, dil), k + ".1")))
        return n.q(F.relu(n.conv(t, k + ".0", 1, pad, dil)))                       # biased conv -> ReLU

    s = "SceneUnderstandingModule."
    N, _, Hf, Wf = y.shape
    x1 = n.q(drop(F.avg_pool2d(y, kernel_size, kernel_size, kernel_size // 2))).reshape(N, -1)
    x1 = n.q(F.relu(n.q(F.linear(x1, P[s + "encoder.global_fc.weight"])) + P[s + "encoder.global_fc.bias"])).view(N, 512, 1, 1)
    x1 = n.q(n.conv(x1, s + "encoder.conv1"))
    feats = [F.interpolate(x1, size=(Hf, Wf), mode="bilinear", align_corners=True)]
    feats.append(cbr(cbr(y, s + "aspp1.0"), s + "aspp1.1"))
    for i, d in enumerate(pyramid):
        feats.append(cbr(cbr(y, s + "aspp%d.0" % (i + 2), d, d), s + "aspp%d.1" % (i + 2)))
    z = n.q(drop(torch.cat(feats, 1)))
    z = n.q(drop(cbr(z, s + "concat_process.1")))
    z = n.q(n.conv(z, s + "concat_process.3"))
    z = n.q(F.interpolate(z, size=tuple(size), mode="bilinear", align_corners=True))
    label, prob = ordinal_layer(z)
    return (label, prob, z) if return_logits else (label, prob)


# ---------------------------------------------------------------------------------------------- MyNet (network/MyNet.py)
def _my_pre(n, x, k, stride=1):
    """Conv2d.forward (MyNet.py:11-15): ELU -> BN -> conv."""
    return n.conv(n.q(n.bn(n.q(F.elu(x)), k + ".bn")), k + ".conv", stride, 1)


def _my_rcu(n, x, k):
    """ResidualConvUnit.forward (MyNet.py:218-230)."""
    y = n.q(F.relu(n.conv(n.q(F.relu(x)), k + ".conv1", pad=1)))
    return n.q(n.conv(y, k + ".conv2", pad=1) + x)


def mynet_forward(P, x, train, momentum=None, q=None, blocks=(6, 12, 36, 24)):
    """MyModel.forward (MyNet.py:270-272) with a DenseNet encoder: encoder.forward (:181-192) -> my_decoder.forward (:135-157).
    x: N x 3 x H x W at the model's input_size (GlobalConsitency's adaptive max-pool is then the identity; restated as the
    pooling call all the same)."""
    n = Net(P, train, q=q, momentum=momentum)
    s0, s1, s2, s3, dense = densenet_features(n, x, "encoder.base_model.", blocks)
    d = "decoder."
    dense = n.q(F.relu(dense))
    x0, x1 = _my_rcu(n, s0, d + "refine0.resConfUnit2"), _my_rcu(n, s1, d + "refine1.resConfUnit2")
    x2, x3 = _my_rcu(n, s2, d + "refine2.resConfUnit2"), _my_rcu(n, s3, d + "refine3.resConfUnit2")
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    half = (x.shape[2] // 2, x.shape[3] // 2)
    g = torch.cat([F.adaptive_max_pool2d(x0, half), F.adaptive_max_pool2d(up(x1), half)], 1)
    glob = _my_pre(n, _my_pre(n, g, d + "global_con.conv"), d + "global_con.conv_final")
    a = _my_pre(n, F.pixel_shuffle(x1, 2), d + "details.down", 2)
    t = torch.cat([a, F.pixel_shuffle(x2, 2)], 1)
    t = _my_pre(n, _my_pre(n, _my_pre(n, t, d + "details.conv"), d + "details.conv2"), d + "details.conv_final")
    detail = up(t)
    tc = lambda t, k: n.q(n.q(F.conv_transpose2d(t, P[d + k + ".weight"], None, stride=2, padding=1)) + P[d + k + ".bias"].view(1, -1, 1, 1))
    sc = torch.cat([x2, tc(x3, "sharpness.tconv0"), tc(tc(dense, "sharpness.tconv1"), "sharpness.tconv2")], 1)
    sharp = n.q(F.relu(n.conv(up(sc), d + "sharpness.up0.1", pad=1)))
    sharp = n.q(F.relu(n.conv(up(sharp), d + "sharpness.up1.1", pad=1)))
    depth = lambda t: torch.sigmoid(n.conv(up(t), d + "get_depth.1", pad=1))
    w, b = P[d + "weighter.mlp.weight"], P[d + "weighter.mlp.bias"]

    def weight(t):
        v = n.q(_my_pre(n, t, d + "weighter.conv", 2)).flatten(2)
        return torch.sigmoid(torch.sum(F.linear(v, w, b), dim=1))[:, None, None]
    out = depth(glob) * weight(glob) + depth(detail) * weight(detail) + depth(sharp) * weight(sharp)
    return out / 3.0 * 10.0


def leaf_state(sd, requires_grad=False):
    """A state dict as independent fp32 leaves (parameters optionally requiring grad; buffers never)."""
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if t.is_floating_point() and requires_grad and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        out[k] = t
    return out
