"""Run under MDE_ACT_DTYPE=fp16 (tests/test_fp16_build_gpu.py starts it as a subprocess): end-to-end checks of the fp16 storage
build (libmde_hip_f16.so) that need a loss scale -- the reference's precision=16 run scales its loss with torch's GradScaler
(train.py:139-140), and so must whoever trains through this build: fp16 gradients below 6e-8 are zero.  Prints one JSON line."""
import json
import sys

import numpy as np
import torch

from oracle import fcrn as ofcrn
from oracle import losses as L
from oracle import metrics as OM
from oracle import nets
from oracle import weights as W


def norm_ratios(named_hip, grads_oracle, scale):
    r = []
    for k, p in named_hip:
        go = grads_oracle.get(k)
        if go is None or p.grad is None or float(go.norm()) < 1e-9:
            continue
        r.append(float((p.grad.detach().cpu() / scale).norm() / go.norm()))
    return np.array(r)


def fcrn(out):
    from mono_depth_estimation_amd import criteria, metrics
    from mono_depth_estimation_amd.network import FCRN
    size = (96, 128)
    ora = ofcrn.FCRNOracle(50, size, out_channels=1)
    W.fcrn_conditioned_state(ora, 11)
    rgb, tgt = W.synthetic_batch(11, 2, *size)
    W.calibrate_running_stats(ora, rgb)
    hip = FCRN.ResNet(layers=50, output_size=size, out_channels=1, pretrained=False)
    hip.load_state_dict(ora.state_dict())
    hip = hip.cuda().eval()
    ora.eval()
    with torch.no_grad():
        y, yo = hip(rgb.cuda()), ora(rgb)
    a_h = float(metrics.MetricComputation(["absrel"]).compute(y, tgt.cuda())[0])
    a_o = float(OM.compute(yo, tgt)["absrel"])
    out["fcrn_eval_absrel_delta"] = abs(a_h - a_o)
    out["fcrn_eval_rel_l2"] = float((y.cpu() - yo).norm() / yo.norm())
    hip.train()
    ora.train()
    for scale, tag in ((1.0, "unscaled"), (4096.0, "scaled")):
        hip.zero_grad(set_to_none=True)
        ora.zero_grad(set_to_none=True)
        loss = criteria.silog_loss(0.85)(hip(rgb.cuda()), tgt.cuda())
        (loss * scale).backward()
        lo = L.silog(ora(rgb), tgt, 0.85)
        lo.backward()
        go = {k: p.grad.clone() for k, p in ora.named_parameters() if p.grad is not None}
        r = norm_ratios(hip.named_parameters(), go, scale)
        out["fcrn_train_loss_rel_%s" % tag] = abs(float(loss.detach()) - float(lo.detach())) / float(lo.detach())
        out["fcrn_grad_norm_within_10pct_%s" % tag] = float(np.mean(np.abs(r - 1) < 0.10))


def midas(out):
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import MiDaS
    torch.manual_seed(0)
    net = MiDaS.MidasNet(features=256)
    sd = W.midas_fixture_state(net, 43)
    rgb, tgt = W.synthetic_batch(43, 2, 64, 96)
    P0 = nets.leaf_state(sd)
    with torch.no_grad():
        nets.midas_forward(P0, rgb, True, momentum=1.0)
    net.load_state_dict({k: v.clone() for k, v in P0.items()})
    net = net.cuda().train()
    P = nets.leaf_state(P0, requires_grad=True)
    L.midas_loss(nets.midas_forward(P, rgb, True)[:, :1], tgt, alpha=0.5, loss="ssimse").backward()
    go = {k: v.grad for k, v in P.items() if v.grad is not None}
    for scale, tag in ((1.0, "unscaled"), (1024.0, "scaled")):
        net.zero_grad(set_to_none=True)
        loss = criteria.MidasLoss(alpha=0.5, loss="ssimse")(net(rgb.cuda())[:, :1], tgt.cuda())
        (loss * scale).backward()
        r = norm_ratios(net.named_parameters(), go, scale)
        out["midas_grad_norm_within_15pct_%s" % tag] = float(np.mean(np.abs(r - 1) < 0.15))


def vnl(out):
    """What 16-bit storage costs where the weights are NOT on a 16-bit grid (a trained state; the fixtures' weights are
    bf16-exact and hide it): the VNL fixture with every conv / linear weight moved off the grid by a relative 1e-3 N(0, 1).  The
    fp32 oracle's AbsRel against the oracle with weights and activations rounded to fp16 and to bf16, and the HIP path of this
    build.  (The free-running version of this measurement -- 20 SGD steps of the oracle -- is
    tests/test_vnl_net_gpu.py::test_vnl_loss_curves_agree_with_the_oracle: 1.07e-4 here against 7.8e-4 for the bf16 build.)"""
    from mono_depth_estimation_amd.network import VNL
    params = nets.vnl_params()
    torch.manual_seed(0)
    net = VNL.MetricDepthModel(params)
    sd = W.vnl_fixture_state(net, 41)
    rgb, tgt = W.synthetic_batch(41, 2, 64, 96)
    g = torch.Generator().manual_seed(5)
    sd = {k: (v * (1.0 + 1e-3 * torch.randn(v.shape, generator=g)) if v.dtype.is_floating_point and v.dim() >= 2 else v) for k, v in sd.items()}
    P = nets.leaf_state(sd)
    with torch.no_grad():
        nets.vnl_forward(P, rgb, True, momentum=1.0)
    net.load_state_dict({k: v.clone() for k, v in P.items()})
    net = net.cuda().eval()
    border = torch.tensor(params.depth_bin_border, dtype=torch.float32)
    rw = lambda f: {k: (f(v) if v.dtype.is_floating_point and v.dim() >= 2 else v) for k, v in P.items()}
    with torch.no_grad():
        dh = L.bins_to_depth(net(rgb.cuda())[1].cpu(), border)
        do = L.bins_to_depth(nets.vnl_forward(P, rgb, False)[1], border)
        d16 = L.bins_to_depth(nets.vnl_forward(rw(nets.fp16_round), rgb, False, q=nets.fp16_round)[1], border)
        dbf = L.bins_to_depth(nets.vnl_forward(rw(nets.bf16_round), rgb, False, q=nets.bf16_round)[1], border)
    t = tgt.clamp(min=0)
    m = t > 0
    absrel = lambda d: float(((d - t).abs() / t.clamp(min=1e-9))[m].mean())
    out["vnl_absrel_shift_hip_fp16"] = abs(absrel(dh) - absrel(do))
    out["vnl_absrel_shift_oracle_fp16"] = abs(absrel(d16) - absrel(do))
    out["vnl_absrel_shift_oracle_bf16"] = abs(absrel(dbf) - absrel(do))


def config5(out):
    """BASELINE configuration 5 at its NAMED precision and full per-GPU size: VNL resnext50 stride 16, 150 bins, 16 x 3 x 480 x 640,
    ModelLoss = WCEL + 6 VNL, SGD -- on the fp16 storage build, with the static loss scale a precision=16 run needs (the
    reference: torch's GradScaler, train.py:139-140).  tests/test_full_size_configs_gpu.py runs the same steps on the bf16 build."""
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import VNL
    params = nets.vnl_params()
    params.crop_size = (480, 640)
    torch.manual_seed(1)
    net = VNL.MetricDepthModel(params).cuda()
    with torch.no_grad():
        net.depth_model.decoder_modules.topdown_predict.conv1.weight.mul_(0.1)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    x = torch.rand(16, 3, 480, 640, generator=g, device="cuda")
    gt = 0.05 + 0.95 * torch.rand(16, 1, 480, 640, generator=g, device="cuda")
    gt = gt.masked_fill(torch.rand(16, 1, 480, 640, generator=g, device="cuda") < 0.1, 0.0)
    crit = criteria.ModelLoss(params)
    bins = criteria.depth_to_bins(gt, params.depth_min, 1.1, params.dec_out_c)
    net.train()
    scale, losses, finite = 1024.0, [], True
    for it in range(4):
        np.random.seed(3)
        net.zero_grad(set_to_none=True)
        logit, prob = net(x)
        loss = crit(criteria.bins_to_depth(prob, params.depth_bin_border), logit, bins, gt)
        (loss * scale).backward()
        if it == 0:
            finite = all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
            out["config5_res2_grad_max"] = float(net.depth_model.encoder_modules.bottomup.res2[0].conv2.weight.grad.abs().max()) / scale
        net._store.sgd_step(2e-3, 2e-3, momentum=0.9, weight_decay=5e-4, grad_scale=1.0 / scale)
        losses.append(float(loss))
    out["config5_losses"], out["config5_grads_finite"] = losses, finite
    out["config5_logit_shape"] = list(logit.shape)


if __name__ == "__main__":
    from mono_depth_estimation_amd import _lib, ops
    assert _lib.ACT_NAME == "fp16" and ops.ACT_DTYPE == torch.float16 and _lib.load().mde_act_dtype() == 1
    res = {"lib": _lib.LIB_NAME}
    if sys.argv[1:] == ["config5"]:
        config5(res)
    else:
        fcrn(res)
        midas(res)
        vnl(res)
    sys.stdout.write(json.dumps(res) + "\n")
