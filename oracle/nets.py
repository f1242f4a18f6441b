"""CPU fp32 oracle of the VNL / MiDaS / BTS / Eigen networks — TEST INFRASTRUCTURE ONLY.

FUNCTIONAL restatements over a state dict: `P` maps the reference's own state_dict keys to tensors (parameters that
require grad for gradient parity, BN running statistics updated in place in training mode), and every function walks the
keys the way the reference's forward walks its modules.  One dict therefore loads into the reference model, drives this
oracle and loads into the HIP module.  Cited lines are under /root/reference.

Pinned by tests/golden/{vnl_net,midas_net,bts_net,eigen}.npz, minted by tests/golden/gen_golden.py from the reference's own
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
        """(bf16 emulation: the conv result is stored once WITHOUT its bias; the bias joins the following pass.)"""
        y = self.q(F.conv2d(x, self.P[key + ".weight"], None, stride, pad, dil, groups))
        b = self.P.get(key + ".bias") if bias else None
        return y if b is None else y + b.view(1, -1, 1, 1)

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


def vnl_forward(P, x, train, block_counts=(3, 4, 6, 3), momentum=None, q=None):
    """MetricDepthModel.forward (VNL.py:678-693) for the resnext*_32x4d_body_stride16 encoders -> (logits, softmax).
    momentum: override every BatchNorm's (1.0 = "running statistics := this batch's", weights.calibrate_running_stats).
    q: rounding applied wherever the HIP path stores a bf16 tensor (bf16_round: the bf16-emulating oracle)."""
    n = Net(P, train, q=q, momentum=momentum)
    e, d = "depth_model.encoder_modules.", "depth_model.decoder_modules."
    H, W = x.shape[2:]
    b = e + "bottomup."
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
    lats = [torch.cat(xs, 1)]
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


def leaf_state(sd, requires_grad=False):
    """A state dict as independent fp32 leaves (parameters optionally requiring grad; buffers never)."""
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if t.is_floating_point() and requires_grad and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        out[k] = t
    return out
