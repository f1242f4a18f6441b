#!/usr/bin/env python3
"""Mint golden vectors by running the REFERENCE's own code (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Imports /root/reference/{criteria,stdepth_utils,metrics}.py and network/FCRN.py and
records their outputs on deterministic inputs (oracle/weights.py) as small .npz files
next to this script.  The reference never travels to the GPU box; these vectors do.

What is the reference's own arithmetic here, and what is not:
  * criteria.py, metrics.py pure functions, FCRN.py (Unpool / UpProj / weights_init /
    ResNet.forward)                                       -> reference code, run as is.
  * torchvision (absent from the image and from /root/reference; unpinned in the
    reference's requirements.txt): FCRN.py:305 only takes conv1/bn1/relu/maxpool/
    layer1..4 from ``torchvision.models.resnetNN``.  A module object exposing our
    restated trunk (oracle/fcrn.py:ResNetTrunk, public v1.5 definition) is placed in
    sys.modules so the reference file imports; the trunk's arithmetic is torch.nn.
  * torchmetrics (absent): metrics.py:116-123 merely *references* four torchmetrics
    functions in a dict; a namespace with those names lets the file import.  None of
    them is called.
"""
import os
import statistics  # noqa: F401  (the STDLIB module, bound before /root/reference -- which holds a statistics.py of its own, a Blender
#                                   script -- goes on sys.path: torch.optim pulls torch._inductor in lazily, which imports it)
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import fcrn as ofcrn  # noqa: E402
from oracle import weights as W  # noqa: E402

REF = "/root/reference"


def _import_reference():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    for n in (18, 34, 50, 101, 152):
        tvm.__dict__["resnet%d" % n] = (lambda n: (lambda pretrained=False: ofcrn.ResNetTrunk(n)))(n)
    from oracle import trunks
    tvm.densenet161 = trunks.densenet161
    tvm.vgg19_bn = trunks.vgg19_bn
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    tm = types.ModuleType("torchmetrics")
    fn = types.SimpleNamespace(
        regression=types.SimpleNamespace(mean_absolute_error=None, mean_squared_log_error=None,
                                         mean_squared_error=None),
        structural_similarity_index_measure=None)
    tm.functional = fn
    sys.modules.setdefault("torchmetrics", tm)
    sys.path.insert(0, REF)
    import criteria  # noqa
    import metrics  # noqa
    from network import FCRN  # noqa
    return criteria, metrics, FCRN


def _np(t):
    return t.detach().cpu().numpy()


def _loss_and_grad(fn, pred, *rest):
    p = pred.clone().requires_grad_(True)
    out = fn(p, *rest)
    out.backward()
    return _np(out), _np(p.grad)


def depth_pair(seed, shape, hole=0.1):
    pred = W.uniform(seed, "pred", shape, 0.05, 1.2)
    tgt = W.uniform(seed, "tgt", shape, 0.02, 1.0)
    tgt = tgt.masked_fill(W.uniform(seed, "hole", shape) < hole, 0.0)
    return pred, tgt


def gen_losses(criteria):
    out = {}
    # G1 silog
    est, gt = depth_pair(11, (4, 1, 64, 64))
    out["g1_est"], out["g1_gt"] = _np(est), _np(gt)
    for lam in (0.85, 0.5):
        l, g = _loss_and_grad(criteria.silog_loss(lam), est, gt)
        out["g1_silog_%g" % lam], out["g1_silog_%g_grad" % lam] = l, g
    # G2 family on 4x1x64x96
    pred, tgt = depth_pair(12, (4, 1, 64, 96))
    out["g2_pred"], out["g2_tgt"] = _np(pred), _np(tgt)
    l, g = _loss_and_grad(criteria.MaskedDepthLoss(), pred, tgt)
    out["g2_masked_depth"], out["g2_masked_depth_grad"] = l, g
    for name, mod in (("masked_mse", criteria.MaskedMSELoss()), ("masked_l1", criteria.MaskedL1Loss()),
                      ("berhu", criteria.berHuLoss())):
        l, g = _loss_and_grad(mod, pred, tgt)
        out["g2_" + name], out["g2_%s_grad" % name] = l, g
    p3, t3 = pred[:, 0], tgt[:, 0]
    m3 = (t3 > 0).float()
    s, t = criteria.compute_scale_and_shift(p3, t3, m3)
    out["g2_scale"], out["g2_shift"] = _np(s), _np(t)
    for red in ("batch-based", "image-based"):
        gl = criteria.GradientLoss(scales=4, reduction=red)
        l, g = _loss_and_grad(lambda p: gl(p, t3, m3), p3)
        out["g2_gradient_%s" % red[:5]], out["g2_gradient_%s_grad" % red[:5]] = l, g
    for kind in ("ssimse", "ssil1", "mse", "l1", "trim", "ssitrim"):
        l, g = _loss_and_grad(criteria.MidasLoss(alpha=0.5, loss=kind), pred, tgt)
        out["g2_midas_" + kind], out["g2_midas_%s_grad" % kind] = l, g
    l, g = _loss_and_grad(criteria.TrimmedProcrustesLoss(alpha=0.5), pred, tgt)
    out["g2_procrustes"], out["g2_procrustes_grad"] = l, g
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses.npz", len(out), "arrays")


def gen_vnl(criteria):
    """G3: VNL_Loss on 2x1x48x64 with the sampled indices stored; WCEL_Loss / ModelLoss on 2x150x24x32.

    criteria.py:924-930 uses ``np.int``, removed from numpy >= 1.24 (this image has 2.2): the alias is restored
    HERE, in the generator's process, before the call; the reference file is untouched.  VNL_Loss draws its
    sample indices from the global numpy RNG, so the stream is seeded and the very same draw is recorded by
    calling the reference's own select_index() under the same seed."""
    if not hasattr(np, "int"):
        np.int = int
    out = {}
    Hh, Ww, fx = 48, 64, 60.0
    pred, gt = depth_pair(31, (2, 1, Hh, Ww))
    gt[:, :, :4] = 0.0                                   # "padding" rows: mask_pad must reject them
    pred = pred.clone()
    pred[0, 0, 5:9, 7:23] = 0.0                          # exact zeros: the z == 0 overwrite at criteria.py:1004
    vn = criteria.VNL_Loss(focal_x=fx, focal_y=fx, input_size=(Hh, Ww))
    np.random.seed(1234)
    p123 = vn.select_index()
    out["g3_p123"] = np.stack([p123["p%d_y" % i] * Ww + p123["p%d_x" % i] for i in (1, 2, 3)]).astype(np.int32)
    out["g3_pred"], out["g3_gt"], out["g3_fx"] = _np(pred), _np(gt), np.float32(fx)
    for select in (True, False):
        np.random.seed(1234)
        l, g = _loss_and_grad(lambda p: vn(gt, p, select=select), pred)
        out["g3_vnl_%d" % select], out["g3_vnl_%d_grad" % select] = l, g
    # WCEL / ModelLoss at the reference's 150 bins (modules/vnl.py:160-163 builds these fields)
    C, h, w = 150, 24, 32
    dmin, dmax = 0.01, 1.7
    interval = (np.log10(dmax) - np.log10(dmin)) / C
    args = types.SimpleNamespace(
        dec_out_c=C, focal_x=30.0, focal_y=30.0, crop_size=(h, w), diff_loss_weight=6,
        wce_loss_weight=[[np.exp(-0.2 * (i - j) ** 2) for i in range(C)] for j in np.arange(C)])
    border = np.array([np.log10(dmin) + interval * (i + 0.5) for i in range(C)])
    logit = W.normal(32, "logit", (2, C, h, w), std=2.0)        # regenerated by the tests, not stored
    _, dgt = depth_pair(33, (2, 1, h, w))
    dgt = dgt * 1.6
    dgt[:, :, :, :3] = -1.0                              # invalid side padding, label C + 1 (vnl.py:209-216)
    d = dgt.clamp(dmin, dmax)
    bins = ((torch.log10(d) - np.log10(dmin)) / interval).to(torch.int)
    bins[dgt < 0] = C + 1
    bins[bins == C] = C - 1
    out["g3_logit_seed"], out["g3_dgt"], out["g3_bins"] = np.int32(32), _np(dgt), _np(bins)
    out["g3_border"] = border.astype(np.float32)
    wl = criteria.WCEL_Loss(args)
    l, g = _loss_and_grad(lambda x: wl(x, bins, dgt), logit)
    # the full gradient is 0.9 MB of noise-like floats: keep a strided sample plus sums over each axis group
    out["g3_wcel"], out["g3_wcel_grad_sample"] = l, g[:, ::7, ::3, ::5].copy()
    out["g3_wcel_grad_csum"], out["g3_wcel_grad_psum"] = g.sum((2, 3)), g.sum(1)
    out["g3_wcel_grad_abs"] = np.abs(g).sum((2, 3))
    # ModelLoss: depth = 10 ** sum(softmax * border) (modules/vnl.py:219-230), differentiated through to the logits
    args.wce_loss_weight = [[np.exp(-0.2 * (i - j) ** 2) for i in range(C)] for j in np.arange(C)]
    ml = criteria.ModelLoss(args)
    np.random.seed(99)
    p123 = ml.virtual_normal_loss.select_index()
    out["g3_model_p123"] = np.stack([p123["p%d_y" % i] * w + p123["p%d_x" % i] for i in (1, 2, 3)]).astype(np.int32)
    bt = torch.from_numpy(border.astype(np.float32))

    def model(x):
        depth = 10 ** (torch.softmax(x, 1).permute(0, 2, 3, 1) * bt).sum(3, dtype=torch.float32, keepdim=True)
        return ml(depth.permute(0, 3, 1, 2), x, bins, dgt)
    np.random.seed(99)
    l, g = _loss_and_grad(model, logit)
    out["g3_model"], out["g3_model_grad_sample"] = l, g[:, ::7, ::3, ::5].copy()
    out["g3_model_grad_abs"] = np.abs(g).sum((2, 3))
    np.savez_compressed(os.path.join(HERE, "vnl.npz"), **out)
    print("vnl.npz", len(out), "arrays; vnl", float(out["g3_vnl_1"]), float(out["g3_vnl_0"]), "wcel",
          float(out["g3_wcel"]), "model", float(out["g3_model"]))


def _reference_stdepth_loss(criteria, loss, single_layer):
    """The reference's composite criterion is a closure built by BaseModule.setup_criterion
    (modules/base_module.py:124-208).  That file cannot be imported here (top-level pytorch_lightning, wandb,
    cv2-backed datasets, and modules/__init__ runs a torch.hub download), so the ONE method is lifted out of the
    reference file with `ast` at generation time and executed unchanged in a namespace holding exactly the names
    it uses (torch, F, criteria, and stdepth_utils' depth_sort / composite_layers / dssim2d, all imported from
    the reference).  Nothing of it is stored in this repo."""
    import ast
    import torch.nn.functional as F
    import stdepth_utils
    src = open(os.path.join(REF, "modules", "base_module.py")).read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "BaseModule")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "setup_criterion")
    ns = {"torch": torch, "F": F, "criteria": criteria, "depth_sort": stdepth_utils.depth_sort,
          "composite_layers": stdepth_utils.composite_layers, "dssim2d": stdepth_utils.dssim2d}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "base_module.py:setup_criterion", "exec"), ns)
    me = types.SimpleNamespace(single_layer=single_layer, method=types.SimpleNamespace(
        loss=loss, variance_focus=0.85, depth_loss_weight=10.0, comp_loss_weight=2.0, fbdiv_loss_weight=0.2,
        ssim_loss_weight=2.0))
    return ns["setup_criterion"](me)


def stdepth_batch(seed, C, shape=(2, 20, 28), single=None):
    """pred / targ [N, C, H, W] in roughly [-0.1, 1.1], rgba with ~30 % transparent pixels, depth channels with holes."""
    N, H, Wd = shape
    pred = W.uniform(seed, "pred", (N, C, H, Wd), -0.1, 1.1)
    targ = W.uniform(seed, "targ", (N, C, H, Wd), 0.0, 1.0)
    rgba = W.uniform(seed, "rgba", (N, 4, H, Wd), 0.0, 1.0)
    rgba[:, 3] = rgba[:, 3].masked_fill(W.uniform(seed, "hole", (N, H, Wd)) < 0.3, 0.0)
    single = (C == 10) if single is None else single
    d = slice(8, 10) if single else slice(16, 20)
    targ[:, d] = targ[:, d].masked_fill(W.uniform(seed, "dhole", targ[:, d].shape) < 0.2, 0.0)
    pred[:, d] = pred[:, d].abs() + 0.05             # silog takes log(pred)
    return pred, targ, rgba


STDEPTH_CASES = [("mae+composite", True), ("silma", True), ("silms+fbdivergence", True), ("mse", True),
                 ("mae+composite+ssim", True), ("allssim+colorssim", True), ("silma+mse+fbdivergence", False),
                 ("mae+composite", True, 20), ("silma+allssim+composite+ssim", True, 20)]   # laina's default: 20 channels, single_layer


def gen_stdepth(criteria):
    """G6: the composite criterion (reference defaults: laina 'mae+composite', bts 'silma'; weights 10 / 2 / 0.2 / 2)."""
    out = {}
    for C in (10, 20):
        pred, targ, rgba = stdepth_batch(61 + C, C)
        out["c%d_pred" % C], out["c%d_targ" % C], out["c%d_rgba" % C] = _np(pred), _np(targ), _np(rgba)
    pred, targ, rgba = stdepth_batch(90, 20, single=True)      # 20 channels in the single-layer layout (laina's default)
    out["c20s_pred"], out["c20s_targ"], out["c20s_rgba"] = _np(pred), _np(targ), _np(rgba)
    for i, case in enumerate(STDEPTH_CASES):
        loss, single = case[:2]
        C = case[2] if len(case) > 2 else (10 if single else 20)
        fn = _reference_stdepth_loss(criteria, loss, single)
        key = "c20s" if (single and C == 20) else "c%d" % C
        pred, targ, rgba = [torch.from_numpy(out["%s_%s" % (key, k)]) for k in ("pred", "targ", "rgba")]
        p = pred.clone().requires_grad_(True)
        total, full, terms = fn(p, targ, rgba, return_composited=True, return_loss_dict=True)
        total.backward()
        out["k%d_loss" % i], out["k%d_grad" % i], out["k%d_full" % i] = _np(total), _np(p.grad), _np(full)
        out["k%d_terms" % i] = np.array([float(v) for v in terms.values()], dtype=np.float32)
        out["k%d_names" % i] = np.array(list(terms.keys()))
        print("stdepth %-24s single=%d  loss %.6f  %s" % (loss, single, float(total), {k: round(float(v), 5) for k, v in terms.items()}))
    np.savez_compressed(os.path.join(HERE, "stdepth.npz"), **out)


def gen_metrics(metrics):
    pred, tgt = depth_pair(13, (4, 1, 48, 64))
    mc = metrics.MetricComputation(["absrel", "rmse", "delta1", "delta2", "delta3", "log10", "sqrel"])
    vals = mc.compute(pred, tgt)
    out = {"pred": _np(pred), "tgt": _np(tgt)}
    for n, v in zip(mc.names, vals):
        out[n] = _np(v)
    # (mae / mse / msle are torchmetrics functions in the reference, metrics.py:116-121; torchmetrics is not in this
    #  image, so those three are pinned by their published definitions in the oracle, not by a reference run)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print("metrics.npz", {k: float(out[k]) for k in mc.names})


def gen_ssim():
    """Anchor for the 'ssim' metric.  The reference maps that name to torchmetrics' structural_similarity_index_measure
    (metrics.py:123; torchmetrics is absent from the image), so its value cannot be minted here.  What CAN be taken from the
    reference is its own in-tree SSIM (stdepth_utils.py:90-121, imports as is): the same 11-tap sigma-1.5 Gaussian window
    and the same index.  It differs from torchmetrics' in three conventions only -- zero 'same' padding instead of reflect
    padding + crop, a fixed data_range, the clamp of the contrast term -- none of which touches a pixel whose window lies
    inside the image when data_range is passed in and the clamp is off.  So: the reference's SSIM map with the tensors'
    own data_range, unclamped, averaged over the interior [5, H-5) x [5, W-5) = what torchmetrics' definition averages."""
    import stdepth_utils
    noise, tgt = depth_pair(17, (3, 2, 40, 56))
    pred = 0.8 * tgt + 0.6 * (noise - 0.5)            # a prediction correlated with its target (SSIM around one half), some of it < 0
    p = torch.clamp_min(pred, 1e-7)
    R = max(float(p.max() - p.min()), float(tgt.max() - tgt.min()))
    m = stdepth_utils.ssim(p, tgt, dim=2, data_range=R, nonnegative_ssim=False, reduction="none")
    out = {"pred": _np(pred), "tgt": _np(tgt), "data_range": np.float64(R), "ssim_interior": _np(m[..., 5:-5, 5:-5].mean())}
    np.savez_compressed(os.path.join(HERE, "ssim.npz"), **out)
    print("ssim.npz", float(out["ssim_interior"]), "data_range", R)


def gen_upproj(FCRN):
    """G4: reference Unpool and UpProjModule(16), fwd + input/weight grads, train-mode BN."""
    torch.manual_seed(0)
    x = W.normal(14, "x", (2, 16, 6, 8))
    out = {"x": _np(x), "unpool": _np(FCRN.Unpool(16)(x))}
    m = FCRN.UpProj.UpProjModule(16)
    sd = W.fill_state_dict(m, 14)
    m.train()
    xi = x.clone().requires_grad_(True)
    y = m(xi)
    gy = W.normal(14, "gy", tuple(y.shape))
    y.backward(gy)
    out["y"], out["gy"], out["gx"] = _np(y), _np(gy), _np(xi.grad)
    for k, p in m.named_parameters():
        out["grad." + k] = _np(p.grad)
    for k, v in m.state_dict().items():  # running stats after one train step
        if "running" in k:
            out["after." + k] = _np(v)
    for k, v in sd.items():
        out["sd." + k] = _np(v)
    np.savez_compressed(os.path.join(HERE, "upproj.npz"), **out)
    print("upproj.npz y", tuple(y.shape))


def gen_fcrn(criteria, metrics, FCRN):
    """G5: reference FCRN.ResNet(50) on 2x3x96x128, out_channels 1, eval + train BN;
    silog + metrics against a seeded target; backward pinned by per-parameter grad stats."""
    size = (96, 128)
    ref = FCRN.ResNet(layers=50, decoder="upproj", output_size=size, in_channels=3,
                      out_channels=1, pretrained=False)
    W.fcrn_fixture_state(ref, 5)
    rgb, tgt = W.synthetic_batch(5, 2, *size)
    W.calibrate_running_stats(ref, rgb)
    out = {}
    ref.eval()
    with torch.no_grad():
        y = ref(rgb)
    out["eval_out"] = _np(y)
    out["eval_silog"] = _np(criteria.silog_loss(0.85)(y, tgt))
    mc = metrics.MetricComputation(["absrel", "rmse", "delta1"])
    for n, v in zip(mc.names, mc.compute(y, tgt)):
        out["eval_" + n] = _np(v)
    ref.train()
    y = ref(rgb)
    loss = criteria.silog_loss(0.85)(y, tgt)
    loss.backward()
    out["train_out"], out["train_silog"] = _np(y), _np(loss)
    names, gnorm, gsum = [], [], []
    for k, p in ref.named_parameters():
        names.append(k)
        gnorm.append(float(p.grad.double().norm()))
        gsum.append(float(p.grad.double().sum()))
    out["grad_names"] = np.array(names)
    out["grad_norm"], out["grad_sum"] = np.array(gnorm), np.array(gsum)
    out["grad_conv3"] = _np(ref.conv3.weight.grad)
    out["grad_conv1_slice"] = _np(ref.conv1.weight.grad[:8])
    out["after_bn1_running_mean"] = _np(ref.bn1.running_mean)
    out["after_bn1_running_var"] = _np(ref.bn1.running_var)
    out["eval_out_minmax"] = np.array([float(out["eval_out"].min()), float(out["eval_out"].max())])
    out["n_state_keys"] = np.array(len(ref.state_dict()))
    out["n_params"] = np.array(sum(p.numel() for p in ref.parameters()))
    np.savez_compressed(os.path.join(HERE, "fcrn50.npz"), **out)
    print("fcrn50.npz eval_silog", float(out["eval_silog"]), "train_silog", float(out["train_silog"]),
          "keys", int(out["n_state_keys"]), "params", int(out["n_params"]))


def gen_fcrn_decoders(criteria, metrics, FCRN):
    """G5c: the reference network with its other decoders (FCRN.py:68-110: deconv2, deconv3, upconv) on the
    conditioned state, 2x3x64x96: eval output + metrics, train-mode SILog, per-parameter gradient norms."""
    size = (64, 96)
    out = {}
    for dec in ("upconv", "deconv2", "deconv3", "fasterupproj", "fasterupconv"):
        if dec == "fasterupconv":     # the reference's choose_decoder never returns this class (FCRN.py:113, :282-294)
            ref = FCRN.ResNet(layers=50, decoder="upconv", output_size=size, in_channels=3, out_channels=1, pretrained=False)
            ref.upSample = FCRN.FasterUpConv(1024)
        else:
            ref = FCRN.ResNet(layers=50, decoder=dec, output_size=size, in_channels=3, out_channels=1, pretrained=False)
        W.fcrn_conditioned_state(ref, 8)
        rgb, tgt = W.synthetic_batch(8, 2, *size)
        W.calibrate_running_stats(ref, rgb)
        ref.eval()
        with torch.no_grad():
            y = ref(rgb)
        out[dec + "_eval_out"] = _np(y)
        for n, v in zip(metrics.MetricComputation(["absrel", "rmse", "delta1"]).names,
                        metrics.MetricComputation(["absrel", "rmse", "delta1"]).compute(y, tgt)):
            out[dec + "_eval_" + n] = _np(v)
        ref.train()
        loss = criteria.silog_loss(0.85)(ref(rgb), tgt)
        loss.backward()
        out[dec + "_train_silog"] = _np(loss)
        out[dec + "_names"] = np.array([k for k, _ in ref.named_parameters()])
        out[dec + "_grad_norm"] = np.array([float(p.grad.double().norm()) for _, p in ref.named_parameters()])
        out[dec + "_state_keys"] = np.array(list(ref.state_dict().keys()))
        print("fcrn_decoders %-8s absrel %.6f train_silog %.5f params %d range %.3f..%.3f" % (
            dec, float(out[dec + "_eval_absrel"]), float(loss), sum(p.numel() for p in ref.parameters()),
            float(y.min()), float(y.max())))
    np.savez_compressed(os.path.join(HERE, "fcrn_decoders.npz"), **out)


def gen_fcrn_basic_trunks(criteria, metrics, FCRN):
    """G5d: the reference network on the BasicBlock trunks (`layers=18 / 34`, FCRN.py:297-332: num_channels = 512, so
    the UpProj decoder runs 256 -> 16 channels), 2x3x64x96: eval output + metrics, train SILog, gradient norms."""
    size = (64, 96)
    out = {}
    for layers in (18, 34):
        tag = "r%d" % layers
        ref = FCRN.ResNet(layers=layers, decoder="upproj", output_size=size, in_channels=3, out_channels=1, pretrained=False)
        W.fcrn_conditioned_state(ref, 10 + layers, basic=True)
        rgb, tgt = W.synthetic_batch(10 + layers, 2, *size)
        W.calibrate_running_stats(ref, rgb)
        ref.eval()
        with torch.no_grad():
            y = ref(rgb)
        out[tag + "_eval_out"] = _np(y)
        mc = metrics.MetricComputation(["absrel", "rmse", "delta1"])
        for n, v in zip(mc.names, mc.compute(y, tgt)):
            out[tag + "_eval_" + n] = _np(v)
        ref.train()
        loss = criteria.silog_loss(0.85)(ref(rgb), tgt)
        loss.backward()
        out[tag + "_train_silog"] = _np(loss)
        out[tag + "_names"] = np.array([k for k, _ in ref.named_parameters()])
        out[tag + "_grad_norm"] = np.array([float(p.grad.double().norm()) for _, p in ref.named_parameters()])
        out[tag + "_state_keys"] = np.array(list(ref.state_dict().keys()))
        print("fcrn_basic_trunks resnet%d absrel %.6f train_silog %.5f params %d keys %d range %.3f..%.3f" % (
            layers, float(out[tag + "_eval_absrel"]), float(loss), sum(p.numel() for p in ref.parameters()),
            len(ref.state_dict()), float(y.min()), float(y.max())))
    np.savez_compressed(os.path.join(HERE, "fcrn_basic_trunks.npz"), **out)


def gen_fcrn_in_channels(criteria, metrics, FCRN):
    """G5e: `in_channels != 3` (FCRN.py:307-313: a fresh conv1 / bn1), here 4 (RGB-D) and 1, ResNet-50, 2xCx64x96."""
    size = (64, 96)
    out = {}
    for cin in (4, 1):
        tag = "c%d" % cin
        ref = FCRN.ResNet(layers=50, decoder="upproj", output_size=size, in_channels=cin, out_channels=1, pretrained=False)
        W.fcrn_conditioned_state(ref, 40 + cin)
        x = W.uniform(40 + cin, "x", (2, cin) + size)
        _, tgt = W.synthetic_batch(40 + cin, 2, *size)
        W.calibrate_running_stats(ref, x)
        ref.eval()
        with torch.no_grad():
            y = ref(x)
        out[tag + "_eval_out"] = _np(y)
        mc = metrics.MetricComputation(["absrel", "rmse", "delta1"])
        for n, v in zip(mc.names, mc.compute(y, tgt)):
            out[tag + "_eval_" + n] = _np(v)
        ref.train()
        loss = criteria.silog_loss(0.85)(ref(x), tgt)
        loss.backward()
        out[tag + "_train_silog"] = _np(loss)
        out[tag + "_conv1_grad"] = _np(ref.conv1.weight.grad)
        out[tag + "_grad_norm"] = np.array([float(p.grad.double().norm()) for _, p in ref.named_parameters()])
        print("fcrn_in_channels %d absrel %.6f train_silog %.5f range %.3f..%.3f" % (
            cin, float(out[tag + "_eval_absrel"]), float(loss), float(y.min()), float(y.max())))
    np.savez_compressed(os.path.join(HERE, "fcrn_in_channels.npz"), **out)


def gen_fcrn_conditioned(criteria, metrics, FCRN):
    """G5b: the same reference network on the well-conditioned state (oracle/weights.py:
    fcrn_conditioned_state) — the fixture on which the 1e-4 AbsRel bound is asserted."""
    size = (96, 128)
    ref = FCRN.ResNet(layers=50, decoder="upproj", output_size=size, in_channels=3,
                      out_channels=1, pretrained=False)
    W.fcrn_conditioned_state(ref, 7)
    rgb, tgt = W.synthetic_batch(7, 2, *size)
    W.calibrate_running_stats(ref, rgb)
    out = {}
    ref.eval()
    with torch.no_grad():
        y = ref(rgb)
    out["eval_out"] = _np(y)
    out["eval_silog"] = _np(criteria.silog_loss(0.85)(y, tgt))
    mc = metrics.MetricComputation(["absrel", "rmse", "delta1", "delta2", "delta3", "log10"])
    for n, v in zip(mc.names, mc.compute(y, tgt)):
        out["eval_" + n] = _np(v)
    ref.train()
    y = ref(rgb)
    loss = criteria.silog_loss(0.85)(y, tgt)
    out["train_out"], out["train_silog"] = _np(y), _np(loss)
    np.savez_compressed(os.path.join(HERE, "fcrn50_cond.npz"), **out)
    print("fcrn50_cond.npz eval_absrel", float(out["eval_absrel"]), "eval_silog", float(out["eval_silog"]),
          "range", float(out["eval_out"].min()), float(out["eval_out"].max()))


def _grad_norms(model):
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    return np.array(names), np.array([float(dict(model.named_parameters())[n].grad.norm()) for n in names], dtype=np.float32)


VNL_SIZE = (64, 96)


def vnl_bins(depth, params):
    """modules/vnl.py:202-217 depth_to_bins, restated (the `modules` package does not import here, see above)."""
    C = params.dec_out_c
    invalid = depth < 0.
    d = depth.clamp(params.depth_min, 1.1)
    bins = ((torch.log10(d) - params.depth_min_log) / params.depth_bin_interval).to(torch.int)
    bins[invalid] = C + 1
    bins[bins == C] = C - 1
    return bins


def gen_vnl_net(criteria):
    """C4: the reference's own network/VNL.py (MetricDepthModel, resnext50_32x4d_body_stride16) on a deterministic,
    well-conditioned state: eval and train outputs, criteria.ModelLoss on them (the index draw stored), gradient norms of
    every parameter.  VNL.py only needs `import torchvision` to succeed (it touches torchvision under __main__ only)."""
    from network import VNL
    from oracle import nets
    if not hasattr(np, "int"):
        np.int = int
    params = nets.vnl_params()
    torch.manual_seed(0)
    ref = VNL.MetricDepthModel(params)
    W.vnl_fixture_state(ref, 41)
    H, Wd = VNL_SIZE
    rgb, tgt = W.synthetic_batch(41, 2, H, Wd)
    W.calibrate_running_stats(ref, rgb)
    border = torch.tensor(params.depth_bin_border, dtype=torch.float32)
    to_depth = lambda prob: 10 ** (prob.permute(0, 2, 3, 1) * border).sum(3, dtype=torch.float32, keepdim=True).permute(0, 3, 1, 2)
    out = {"keys": np.array(list(ref.state_dict().keys()))}
    ref.eval()
    with torch.no_grad():
        logit, prob = ref(rgb)
    out["eval_depth"], out["eval_logit_s"], out["eval_prob_s"] = _np(to_depth(prob)), _np(logit[:, ::5, ::4, ::4]), _np(prob[:, ::5, ::4, ::4])
    out["eval_logit_csum"] = _np(logit.sum((2, 3)))
    ref.train()
    params.crop_size = (H, Wd)
    crit = criteria.ModelLoss(params)
    np.random.seed(77)
    p123 = crit.virtual_normal_loss.select_index()
    out["p123"] = np.stack([p123["p%d_y" % i] * Wd + p123["p%d_x" % i] for i in (1, 2, 3)]).astype(np.int32)
    gt = tgt.clone()
    gt[:, :, :, :4] = -1.0                                # invalid side padding (vnl.py:209-216)
    bins = vnl_bins(gt.clone(), params)
    gt_loss = gt.clone()
    gt_loss[gt_loss < params.depth_min] = torch.where(gt_loss[gt_loss < params.depth_min] < 0, torch.tensor(-1.0), torch.tensor(params.depth_min))
    logit, prob = ref(rgb)
    np.random.seed(77)
    loss = crit(to_depth(prob), logit, bins, gt_loss)
    loss.backward()
    out["gt"], out["bins"] = _np(gt_loss), _np(bins)
    out["train_depth"], out["train_logit_s"], out["train_loss"] = _np(to_depth(prob)), _np(logit[:, ::5, ::4, ::4]), _np(loss)
    out["grad_names"], out["grad_norms"] = _grad_norms(ref)
    sd = ref.state_dict()
    out["rm_res5"] = _np(sd["depth_model.encoder_modules.bottomup.res5.2.bn3.running_mean"])
    out["rv_aspp"] = _np(sd["depth_model.encoder_modules.bottomup_top.globalpool_bn.running_var"])
    np.savez_compressed(os.path.join(HERE, "vnl_net.npz"), **out)
    print("vnl_net.npz: %d keys, %d params, eval depth range %.4f..%.4f, train loss %.5f" % (
        len(out["keys"]), sum(p.numel() for p in ref.parameters()), out["eval_depth"].min(), out["eval_depth"].max(), float(loss)))


def gen_vnl_mobilenet(criteria):
    """C4 with `--encoder mobilenetv2_body_stride8`: the reference's own MobileNetV2 / Global_pool_block (VNL.py:389-537,
    :172-187) under MetricDepthModel, same protocol as gen_vnl_net."""
    from network import VNL
    from oracle import nets
    if not hasattr(np, "int"):
        np.int = int
    H, Wd = VNL_SIZE
    params = nets.vnl_mobilenet_params((H, Wd))
    torch.manual_seed(0)
    ref = VNL.MetricDepthModel(params)
    W.vnl_mobilenet_fixture_state(ref, 45)
    rgb, tgt = W.synthetic_batch(45, 2, H, Wd)
    W.calibrate_running_stats(ref, rgb)
    border = torch.tensor(params.depth_bin_border, dtype=torch.float32)
    to_depth = lambda prob: 10 ** (prob.permute(0, 2, 3, 1) * border).sum(3, dtype=torch.float32, keepdim=True).permute(0, 3, 1, 2)
    out = {"keys": np.array(list(ref.state_dict().keys()))}
    ref.eval()
    with torch.no_grad():
        logit, prob = ref(rgb)
    out["eval_depth"], out["eval_logit_s"], out["eval_prob_s"] = _np(to_depth(prob)), _np(logit[:, ::5, ::4, ::4]), _np(prob[:, ::5, ::4, ::4])
    out["eval_logit_csum"] = _np(logit.sum((2, 3)))
    ref.train()
    crit = criteria.ModelLoss(params)
    np.random.seed(77)
    p123 = crit.virtual_normal_loss.select_index()
    out["p123"] = np.stack([p123["p%d_y" % i] * Wd + p123["p%d_x" % i] for i in (1, 2, 3)]).astype(np.int32)
    gt = tgt.clone()
    gt[:, :, :, :4] = -1.0
    bins = vnl_bins(gt.clone(), params)
    gt_loss = gt.clone()
    gt_loss[gt_loss < params.depth_min] = torch.where(gt_loss[gt_loss < params.depth_min] < 0, torch.tensor(-1.0), torch.tensor(params.depth_min))
    logit, prob = ref(rgb)
    np.random.seed(77)
    loss = crit(to_depth(prob), logit, bins, gt_loss)
    loss.backward()
    out["gt"], out["bins"] = _np(gt_loss), _np(bins)
    out["train_depth"], out["train_logit_s"], out["train_loss"] = _np(to_depth(prob)), _np(logit[:, ::5, ::4, ::4]), _np(loss)
    out["grad_names"], out["grad_norms"] = _grad_norms(ref)
    sd = ref.state_dict()
    out["rm_res5"] = _np(sd["depth_model.encoder_modules.bottomup.res5.3.conv.7.running_mean"])
    out["rv_top"] = _np(sd["depth_model.encoder_modules.bottomup_top.globalpool_bn.running_var"])
    np.savez_compressed(os.path.join(HERE, "vnl_mbv2.npz"), **out)
    print("vnl_mbv2.npz: %d keys, %d params, eval depth range %.4f..%.4f, train loss %.5f" % (
        len(out["keys"]), sum(p.numel() for p in ref.parameters()), out["eval_depth"].min(), out["eval_depth"].max(), float(loss)))


MIDAS_SIZE = (64, 96)


def gen_midas_net(criteria):
    """C3: the reference's own network/MiDaS.py (MidasNet: _make_scratch, FeatureFusionBlock, ResidualConvUnit, Interpolate,
    output_conv, forward).  Its trunk comes from torch.hub (MiDaS.py:110, never executed): the one function that would
    download it is replaced by the local resnext101_32x8d stand-in (oracle/trunks.py), wired through the reference's own
    _make_resnet_backbone.  Loss: criteria.MidasLoss(0.5, 'ssimse') on channel 0."""
    from network import MiDaS
    from oracle import trunks
    MiDaS._make_pretrained_resnext101_wsl = lambda use_pretrained: MiDaS._make_resnet_backbone(trunks.resnext101_32x8d())
    torch.manual_seed(0)
    ref = MiDaS.MidasNet(features=256)
    W.midas_fixture_state(ref, 43)
    H, Wd = MIDAS_SIZE
    rgb, tgt = W.synthetic_batch(43, 2, H, Wd)
    W.calibrate_running_stats(ref, rgb)
    out = {"keys": np.array(list(ref.state_dict().keys()))}
    ref.eval()
    with torch.no_grad():
        y = ref(rgb)
    out["eval_out"] = _np(y)
    ref.train()
    y = ref(rgb)
    loss = criteria.MidasLoss(alpha=0.5, loss="ssimse")(y[:, :1], tgt)
    loss.backward()
    out["train_out"], out["train_loss"] = _np(y), _np(loss)
    out["grad_names"], out["grad_norms"] = _grad_norms(ref)
    sd = ref.state_dict()
    out["rm_l4"] = _np(sd["pretrained.layer4.2.bn3.running_mean"])
    np.savez_compressed(os.path.join(HERE, "midas_net.npz"), **out)
    print("midas_net.npz: %d keys, %d params, eval range %.4f..%.4f, train loss %.5f" % (
        len(out["keys"]), sum(p.numel() for p in ref.parameters()), out["eval_out"].min(), out["eval_out"].max(), float(loss)))


BTS_SIZE = (64, 96)


def gen_bts_net(criteria):
    """C2: the reference's own network/Bts.py (BtsModel: encoder's feature walk, bts decoder with upconv / atrous_conv /
    reduction_1x1 / local_planar_guidance) over the densenet161 stand-in (oracle/trunks.py, handed over as
    torchvision.models.densenet161).  Loss: criteria.silog_loss(0.85) on the final depth."""
    from network import Bts
    torch.manual_seed(0)
    ref = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=1, encoder_version="densenet161_bts")
    W.bts_fixture_state(ref, 47)
    H, Wd = BTS_SIZE
    rgb, tgt = W.synthetic_batch(47, 2, H, Wd)
    W.calibrate_running_stats(ref, rgb)
    out = {"keys": np.array(list(ref.state_dict().keys()))}
    names = ("d8", "d4", "d2", "r1", "final")
    ref.eval()
    with torch.no_grad():
        ys = ref(rgb)
    for nme, y in zip(names, ys):
        out["eval_" + nme] = _np(y)
    ref.train()
    ys = ref(rgb)
    loss = criteria.silog_loss(0.85)(ys[4], tgt * 10.0)
    loss.backward()
    for nme, y in zip(names, ys):
        out["train_" + nme] = _np(y)
    out["train_loss"] = _np(loss)
    out["grad_names"], out["grad_norms"] = _grad_norms(ref)
    sd = ref.state_dict()
    out["rm_norm5"] = _np(sd["encoder.base_model.norm5.running_mean"])
    out["rv_bn4_2"] = _np(sd["decoder.bn4_2.running_var"])
    np.savez_compressed(os.path.join(HERE, "bts_net.npz"), **out)
    print("bts_net.npz: %d keys, %d params, eval final range %.4f..%.4f, train loss %.5f" % (
        len(out["keys"]), sum(p.numel() for p in ref.parameters()), out["eval_final"].min(), out["eval_final"].max(), float(loss)))


def gen_bts_resnet(criteria):
    """C2's other encoders (Bts.py:293-307): the reference's BtsModel over a ResNet-50 and a ResNeXt-50 32x4d.  Bts.py keeps the
    WHOLE torchvision model as `encoder.base_model` (fc included), so for this generator `torchvision.models.resnet50` /
    `resnext50_32x4d` are the full stand-ins of oracle/trunks.py (the FCRN generators keep the trunk-only stand-in they were
    minted with)."""
    import torchvision.models as tvm
    from network import Bts
    from oracle import trunks
    saved = {k: tvm.__dict__.get(k) for k in ("resnet50", "resnext50_32x4d", "resnet101", "resnext101_32x8d")}
    tvm.resnet50, tvm.resnext50_32x4d = trunks.resnet50_full, trunks.resnext50_32x4d_full
    tvm.resnet101, tvm.resnext101_32x8d = trunks.resnet101_full, trunks.resnext101_32x8d_full
    try:
        for version, seed in (("resnet50_bts", 57), ("resnext50_bts", 59), ("resnet101_bts", 63), ("resnext101_bts", 65)):
            torch.manual_seed(0)
            ref = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=1, encoder_version=version)
            W.bts_resnet_fixture_state(ref, seed)
            H, Wd = BTS_SIZE
            rgb, tgt = W.synthetic_batch(seed, 2, H, Wd)
            W.calibrate_running_stats(ref, rgb)
            out = {"keys": np.array(list(ref.state_dict().keys())), "n_params": np.int64(sum(p.numel() for p in ref.parameters()))}
            ref.eval()
            with torch.no_grad():
                ys = ref(rgb)
            for nme, y in zip(("d8", "d4", "d2", "r1", "final"), ys):
                out["eval_" + nme] = _np(y).astype(np.float16) if nme != "final" else _np(y)
            ref.train()
            ys = ref(rgb)
            loss = criteria.silog_loss(0.85)(ys[4], tgt * 10.0)
            loss.backward()
            out["train_loss"] = _np(loss)
            out["grad_names"], out["grad_norms"] = _grad_norms(ref)
            np.savez_compressed(os.path.join(HERE, "bts_%s.npz" % version[:-4]), **out)
            print("bts_%s.npz: %d keys, %d params, eval final range %.4f..%.4f, train loss %.5f" % (
                version[:-4], len(out["keys"]), int(out["n_params"]), out["eval_final"].min(), out["eval_final"].max(), float(loss)))
    finally:
        for k, v in saved.items():
            if v is None:
                tvm.__dict__.pop(k, None)
            else:
                tvm.__dict__[k] = v


def gen_bts_image_residuals(criteria):
    """Bts.py:264-271: BtsModel(out_channels=10, image_residuals=True) -- final_depth = two RGBA layers as residuals on the input
    image + two depths.  Eval output and, in train mode, a masked-L1-like scalar (mean |final - target|) with its gradient norms."""
    from network import Bts
    torch.manual_seed(0)
    ref = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=10, image_residuals=True, encoder_version="densenet161_bts")
    W.bts_conditioned_state(ref, 61)
    H, Wd = BTS_SIZE
    rgb, _ = W.synthetic_batch(61, 2, H, Wd)
    target = W.uniform(61, "layers", (2, 10, H, Wd), 0.0, 1.0)
    W.calibrate_running_stats(ref, rgb)
    ref.eval()
    with torch.no_grad():
        final = ref(rgb)[4]
    out = {"eval_final": _np(final).astype(np.float16), "clamped_share": np.float64(((final[:, :8] <= 0) | (final[:, :8] >= 1)).float().mean())}
    ref.train()
    loss = (ref(rgb)[4] - target).abs().mean()
    loss.backward()
    out["train_loss"] = _np(loss)
    out["grad_names"], out["grad_norms"] = _grad_norms(ref)
    np.savez_compressed(os.path.join(HERE, "bts_imgres.npz"), **out)
    print("bts_imgres.npz: final %s range %.3f..%.3f, %.1f %% of the colour values clamped, train loss %.5f" % (
        tuple(final.shape), float(final.min()), float(final.max()), 100 * float(out["clamped_share"]), float(loss)))


def gen_bts_conditioned(criteria, metrics):
    """C2 on a WELL-CONDITIONED state (oracle/weights.bts_conditioned_state: the north-star bound |dAbsRel| <= 1e-4 is only
    meaningful where the fp32 reference itself is stable under bf16 storage): the reference's own network/Bts.py, eval
    outputs and the metrics of the final depth from the reference's metrics.py on the 8-image eval batch
    (weights.BTS_COND_BATCH; the five outputs are kept for its first two images), and the SILog of one train-mode forward on
    the 2-image batch the convergence test trains on."""
    from network import Bts
    H, Wd = BTS_SIZE
    torch.manual_seed(0)
    ref = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=1, encoder_version="densenet161_bts")
    W.bts_conditioned_state(ref, 53)
    rgb, tgt = W.synthetic_batch(53, W.BTS_COND_BATCH, H, Wd)
    W.calibrate_running_stats(ref, rgb)
    ref.eval()
    with torch.no_grad():
        ys = ref(rgb)
    out = {"eval_" + nme: _np(y[:2]) for nme, y in zip(("d8", "d4", "d2", "r1", "final"), ys)}
    mc = metrics.MetricComputation(["absrel", "rmse", "delta1", "log10"])
    for n, v in zip(mc.names, mc.compute(ys[4], tgt * 10.0)):
        out["eval_" + n] = _np(v)
    W.bts_conditioned_state(ref, 53)
    rgb2, tgt2 = W.synthetic_batch(53, 2, H, Wd)
    W.calibrate_running_stats(ref, rgb2)
    ref.train()
    out["train_loss"] = _np(criteria.silog_loss(0.85)(ref(rgb2)[4], tgt2 * 10.0))
    np.savez_compressed(os.path.join(HERE, "bts_cond.npz"), **out)
    print("bts_cond.npz: eval final range %.4f..%.4f, AbsRel %.6f, train SILog %.5f" % (
        float(ys[4].min()), float(ys[4].max()), float(out["eval_absrel"]), float(out["train_loss"])))


def gen_bts_curve(criteria):
    """Convergence parity pinned to the REFERENCE's own training run: network/Bts.py's BtsModel on the conditioned state,
    20 torch.optim.AdamW steps as modules/bts.py:139-152 configures them (eps 1e-3, weight decay 1e-2 on the encoder / 0 on the
    decoder, lr 1e-4) on one 2-image batch, criteria.silog_loss(0.85) on the final depth: the loss of every step.  (Round 3's
    test trained the CPU oracle beside the HIP path at test time: 129 s of the GPU suite.)"""
    from network import Bts
    torch.manual_seed(0)
    ref = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=1, encoder_version="densenet161_bts")
    W.bts_conditioned_state(ref, 53)
    rgb, tgt = W.synthetic_batch(53, 2, *BTS_SIZE)
    W.calibrate_running_stats(ref, rgb)
    ref.train()
    enc = [p for n, p in ref.named_parameters() if n.startswith("encoder.") and p.requires_grad]
    dec = [p for n, p in ref.named_parameters() if not n.startswith("encoder.") and p.requires_grad]
    opt = torch.optim.AdamW([{"params": enc, "weight_decay": 1e-2}, {"params": dec, "weight_decay": 0.0}], lr=1e-4, eps=1e-3)
    crit = criteria.silog_loss(0.85)
    curve = []
    for _ in range(20):
        opt.zero_grad()
        loss = crit(ref(rgb)[4], tgt * 10.0)
        loss.backward()
        opt.step()
        curve.append(float(loss))
    np.savez_compressed(os.path.join(HERE, "bts_curve.npz"), silog=np.array(curve, dtype=np.float64), lr=1e-4, steps=20)
    print("bts_curve.npz: SILog %.5f -> %.5f over 20 AdamW steps" % (curve[0], curve[-1]))


OFFGRID_FCRN_SEEDS = (7, 21, 22)


def gen_offgrid(criteria, metrics, FCRN):
    """The north-star bound "AbsRel within 1e-4 of the CPU reference on identical weights" for weights that are NOT on the
    16-bit grid (every other conditioned fixture rounds its conv weights to bf16 first): the four networks of BASELINE
    configurations 2-5 -- the reference's own network/FCRN.py (three seeds), Bts.py, VNL.py, MiDaS.py -- on their conditioned
    states with every conv / linear weight moved off the grid (oracle/weights.off_grid), eval-mode outputs and the
    reference's metrics.py on them.  Also the ON-grid golden of the conditioned MiDaS state (weights.midas_conditioned_state)."""
    from network import Bts, MiDaS, VNL
    from oracle import nets, trunks
    names = ["absrel", "rmse", "delta1", "log10"]
    out = {}

    def record(tag, y, tgt, keep_output):
        mc = metrics.MetricComputation(names)
        for n, v in zip(mc.names, mc.compute(y, tgt)):
            out["%s_%s" % (tag, n)] = _np(v)
        if keep_output:
            out[tag + "_out"] = _np(y)
        print("offgrid.npz %-12s range %.4f..%.4f AbsRel %.6f" % (tag, float(y.min()), float(y.max()), float(out[tag + "_absrel"])))

    size = (96, 128)
    for seed in OFFGRID_FCRN_SEEDS:
        ref = FCRN.ResNet(layers=50, decoder="upproj", output_size=size, in_channels=3, out_channels=1, pretrained=False)
        W.off_grid(ref, W.fcrn_conditioned_state(ref, seed), seed)
        rgb, tgt = W.synthetic_batch(seed, W.OFFGRID_FCRN_BATCH, *size)
        W.calibrate_running_stats(ref, rgb)
        ref.eval()
        with torch.no_grad():
            y = ref(rgb)
            record("fcrn_s%d" % seed, y, tgt, False)
            if seed == OFFGRID_FCRN_SEEDS[0]:
                out["fcrn_s%d_out" % seed] = _np(y[:2])

    torch.manual_seed(0)
    ref = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=1, encoder_version="densenet161_bts")
    W.off_grid(ref, W.bts_conditioned_state(ref, 53), 53)
    rgb, tgt = W.synthetic_batch(53, W.BTS_COND_BATCH, *BTS_SIZE)
    W.calibrate_running_stats(ref, rgb)
    ref.eval()
    with torch.no_grad():
        y = ref(rgb)[4]
        record("bts", y, tgt * 10.0, False)
        out["bts_out"] = _np(y[:2])

    if not hasattr(np, "int"):
        np.int = int
    params = nets.vnl_params()
    torch.manual_seed(0)
    ref = VNL.MetricDepthModel(params)
    W.off_grid(ref, W.vnl_fixture_state(ref, 41), 41)
    rgb, tgt = W.synthetic_batch(41, W.OFFGRID_BATCH, *VNL_SIZE)
    W.calibrate_running_stats(ref, rgb)
    border = torch.tensor(params.depth_bin_border, dtype=torch.float32)
    ref.eval()
    with torch.no_grad():
        prob = ref(rgb)[1]
        depth = 10 ** (prob.permute(0, 2, 3, 1) * border).sum(3, dtype=torch.float32, keepdim=True).permute(0, 3, 1, 2)
        record("vnl", depth, tgt, False)
        out["vnl_out"] = _np(depth[:2])

    MiDaS._make_pretrained_resnext101_wsl = lambda use_pretrained: MiDaS._make_resnet_backbone(trunks.resnext101_32x8d())
    for tag, off in (("midas_ongrid", False), ("midas", True)):
        torch.manual_seed(0)
        ref = MiDaS.MidasNet(features=256)
        sd = W.midas_conditioned_state(ref, 43)
        if off:
            W.off_grid(ref, sd, 43)
        rgb, tgt = W.synthetic_batch(43, W.OFFGRID_BATCH, *MIDAS_SIZE)
        W.calibrate_running_stats(ref, rgb)
        ref.eval()
        with torch.no_grad():
            y = ref(rgb)[:, :1]
            record(tag, y, tgt, False)
            out[tag + "_out"] = _np(y[:2])
    np.savez_compressed(os.path.join(HERE, "offgrid.npz"), **out)


def gen_eigen(criteria):
    """C1 / BASELINE configuration 1 (CPU plumbing): the reference's own network/Eigen.py (Eigen, Scale2, Scale3, VGG) over
    the vgg19_bn stand-in, 4 x 3 x 240 x 320 (the two Linear layers fix that input size; 64 x 64 is rejected by the
    reference itself), with SILog (what BASELINE.json pairs it with) and MaskedDepthLoss (what
    modules/eigen.py pairs it with) on the output resized to the target as eigen.py:30 does."""
    from network import Eigen
    torch.manual_seed(0)
    ref = Eigen.Eigen(scale1="vgg", pretrained=False)
    sd = W.fill_state_dict(ref, 53)
    for k in sd:                                           # keep the 237 M-parameter Linear layers from saturating anything
        if k.startswith("scale1.mlp"):
            sd[k] = sd[k] * 0.3
    ref.load_state_dict(sd)
    rgb, tgt = W.synthetic_batch(53, 4, 240, 320)
    W.calibrate_running_stats(ref, rgb)
    out = {"n_params": np.int64(sum(p.numel() for p in ref.parameters())), "n_keys": np.int64(len(ref.state_dict()))}
    ref.eval()
    with torch.no_grad():
        y = ref(rgb)
    out["eval_out"] = _np(y)
    ref.train()
    y = ref(rgb)
    up = torch.nn.functional.interpolate(y, (240, 320), mode="bilinear")
    silog = criteria.silog_loss(0.85)(up + 0.1, tgt)       # (+0.1: the head ends in a ReLU, SILog needs a positive estimate)
    md = criteria.MaskedDepthLoss()(up, tgt)
    (silog + md).backward()
    out["train_out"], out["train_silog"], out["train_masked_depth"] = _np(y), _np(silog), _np(md)
    names = ["scale1.feature_extractor.0.weight", "scale1.feature_extractor.49.bias", "scale1.mlp1.bias", "scale1.mlp2.bias",
             "scale1.upsample.weight", "scale2.conv.weight", "scale2.scale2_onestack.6.weight", "scale3.conv.bias",
             "scale3.scale3_onestack.0.weight", "scale3.scale3_onestack.6.weight"]
    pd = dict(ref.named_parameters())
    out["grad_names"], out["grad_norms"] = np.array(names), np.array([float(pd[k].grad.norm()) for k in names], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "eigen.npz"), **out)
    print("eigen.npz: %d params, out %s range %.4f..%.4f, silog %.5f masked_depth %.5f" % (
        int(out["n_params"]), out["eval_out"].shape, out["eval_out"].min(), out["eval_out"].max(), float(silog), float(md)))


DORN_ARGS = dict(input_size=(65, 81), kernel_size=4, ord_num=12, alpha=0.02, beta=10.0, discretization="SID", pretrained=0,
                 pyramid=[2, 3, 4], batch_norm=0, dropout=0.5)


def gen_dorn_net(criteria):
    """N4: the reference's own network/Dorn.py (DORN: dilated ResNet-101, SceneUnderstandingModule with its three Dropout2d,
    OrdinalRegressionLayer) at 2 x 3 x 65 x 81 with criteria.ordLoss on the SID label map modules/dorn.py:102-107 computes
    (restated inline: modules/ does not import here), both scene-module variants (batch_norm 0 / 1), and ordLoss alone on
    random probabilities with targets that include 0, a negative, -inf and NaN (depth 0 / negative under the log)."""
    from network import Dorn
    out = {}
    for bn in (0, 1):
        args = types.SimpleNamespace(**dict(DORN_ARGS, batch_norm=bn))
        torch.manual_seed(0)
        ref = Dorn.DORN(args)
        W.dorn_fixture_state(ref, 59 + bn)
        H, Wd = args.input_size
        rgb, tgt = W.synthetic_batch(59, 2, H, Wd)
        torch.manual_seed(7)                                   # (the calibration pass runs in train mode: its Dropout2d draw too)
        W.calibrate_running_stats(ref, rgb)
        pre = "bn%d_" % bn
        out[pre + "keys"] = np.array(list(ref.state_dict().keys()))
        ref.eval()
        with torch.no_grad():
            label, prob = ref(rgb)
        out[pre + "eval_label"], out[pre + "eval_prob"] = label.numpy().astype(np.int16), _np(prob)
        ref.train()
        torch.manual_seed(1234)                                # the three Dropout2d draw from the global generator
        label, prob = ref(rgb)
        a, b, k = torch.tensor(args.alpha).float(), torch.tensor(args.beta).float(), torch.tensor(args.ord_num).int()
        y_sid = k * torch.log((tgt * 10.0) / a) / torch.log(b / a)          # modules/dorn.py:102-104
        loss = criteria.ordLoss()(prob, y_sid)
        loss.backward()
        out[pre + "train_label"], out[pre + "train_prob"], out[pre + "train_loss"] = label.numpy().astype(np.int16), _np(prob), _np(loss)
        out[pre + "grad_names"], out[pre + "grad_norms"] = _grad_norms(ref)
        sd = ref.state_dict()
        out[pre + "rm_l4"] = _np(sd["backbone.backbone.layer4.2.bn3.running_mean"])
        print("dorn_net bn=%d: %d keys, %d params, eval labels %d..%d (mean %.2f), train loss %.5f" % (
            bn, len(sd), sum(p.numel() for p in ref.parameters()), int(label.min()), int(label.max()), float(label.float().mean()), float(loss)))
    g = torch.Generator().manual_seed(61)
    prob = torch.rand(2, 12, 9, 11, generator=g)
    prob[0, 3, 2, 2], prob[1, 5, 4, 4], prob[0, 0, 0, 0] = 0.0, 1.0, 1e-9      # the clamps
    t = torch.rand(2, 1, 9, 11, generator=g) * 14.0 - 1.0
    t[0, 0, 0, 1], t[0, 0, 0, 2], t[1, 0, 3, 3], t[1, 0, 5, 5] = 3.0, float("-inf"), float("nan"), 0.0
    prob.requires_grad_(True)
    loss = criteria.ordLoss()(prob, t)
    loss.backward()
    out["ord_prob"], out["ord_target"], out["ord_loss"], out["ord_grad"] = _np(prob), _np(t), _np(loss), _np(prob.grad)
    np.savez_compressed(os.path.join(HERE, "dorn_net.npz"), **out)


MYNET_SIZE = (64, 96)


def gen_mynet(criteria):
    """N4: the reference's own network/MyNet.py (MyModel: encoder's feature walk, my_decoder with its residual units, the three
    branches, the shared depth head and the weighter) over the densenet161 stand-in at 2 x 3 x 64 x 96 = the model's
    input_size, with criteria.MidasLoss(alpha=0.5, loss='mse', reduction='batch-based') (modules/my.py:36)."""
    from network import MyNet
    torch.manual_seed(0)
    ref = MyNet.MyModel(input_size=MYNET_SIZE, encoder_version="densenet161_bts")
    W.mynet_fixture_state(ref, 71)
    rgb, tgt = W.synthetic_batch(71, 2, *MYNET_SIZE)
    W.calibrate_running_stats(ref, rgb)
    out = {"keys": np.array(list(ref.state_dict().keys()))}
    ref.eval()
    with torch.no_grad():
        out["eval_out"] = _np(ref(rgb))
    ref.train()
    y = ref(rgb)
    loss = criteria.MidasLoss(alpha=0.5, loss="mse", reduction="batch-based")(y, tgt * 10.0)
    loss.backward()
    out["train_out"], out["train_loss"] = _np(y), _np(loss)
    names, norms = [], []
    for k, p in ref.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(float(p.grad.norm()))
    out["grad_names"], out["grad_norms"] = np.array(names), np.array(norms, dtype=np.float32)
    out["no_grad"] = np.array([k for k, p in ref.named_parameters() if p.grad is None])
    out["rv_weighter"] = _np(ref.state_dict()["decoder.weighter.conv.bn.running_var"])
    np.savez_compressed(os.path.join(HERE, "mynet.npz"), **out)
    print("mynet.npz: %d keys, %d params, eval range %.4f..%.4f, train loss %.5f, %d parameters without gradient" % (
        len(out["keys"]), sum(p.numel() for p in ref.parameters()), out["eval_out"].min(), out["eval_out"].max(), float(loss), len(out["no_grad"])))


def gen_augment():
    """N2: the input pipeline of modules/base_module.py:234-284 as oracle/augment.py composes it over PIL ITSELF (torchvision's
    wrappers are absent and restated there; Pillow 12.2 does every image operation): a 120 x 160 sample with two depth layers,
    four seeded train draws and the val path, stored as uint8 (the outputs are k / 255)."""
    from oracle import augment as OA
    rng = np.random.RandomState(11)
    rgb = torch.from_numpy(rng.rand(3, 120, 160).astype(np.float32))
    depth = [torch.from_numpy(rng.rand(1, 120, 160).astype(np.float32)) for _ in range(2)]
    out = {"rgb": _np(rgb), "depth": _np(torch.cat(depth, 0))}
    for seed in range(4):
        np.random.seed(seed)
        r, d = OA.train_preprocess(rgb, depth, 64, (56, 72))
        out["train%d_rgb" % seed], out["train%d_depth" % seed] = np.round(_np(r) * 255).astype(np.uint8), np.round(_np(d) * 255).astype(np.uint8)
    r, d = OA.val_preprocess(rgb, depth, 64, (56, 72))
    out["val_rgb"], out["val_depth"] = np.round(_np(r) * 255).astype(np.uint8), np.round(_np(d) * 255).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)
    print("augment.npz: PIL", __import__("PIL").__version__)


def gen_vnl_keymap():
    """N3: the reference's own convert_state_dict_resnext (VNL.py:44-67) on the index-path keys of the shipped
    ResNeXt-ImageNet files, enumerated from the documented nn.Sequential structure -> {source key: body key} pairs."""
    import json
    from network import VNL
    src = ["0.weight", "1.weight", "1.bias", "1.running_mean", "1.running_var", "10.1.weight", "10.1.bias", "8.weight"]
    for stage, nblk in zip((4, 5, 6, 7), (3, 4, 6, 3)):
        for b in range(nblk):
            for idx, params in ((0, ("weight",)), (1, ("weight", "bias", "running_mean", "running_var")), (3, ("weight",)),
                                (4, ("weight", "bias", "running_mean", "running_var"))):
                src += ["%d.%d.0.0.0.%d.%s" % (stage, b, idx, p) for p in params]
            src += ["%d.%d.0.0.1.weight" % (stage, b)] + ["%d.%d.0.0.2.%s" % (stage, b, p) for p in ("weight", "bias", "running_mean", "running_var")]
            if b == 0:
                src += ["%d.%d.0.1.0.weight" % (stage, b)] + ["%d.%d.0.1.1.%s" % (stage, b, p) for p in ("weight", "bias", "running_mean", "running_var")]
    dst = VNL.convert_state_dict_resnext({k: i for i, k in enumerate(src)})
    inv = {v: k for k, v in dst.items()}
    pairs = {src[i]: inv[i] for i in sorted(inv)}
    json.dump({"pairs": pairs, "dropped": [k for i, k in enumerate(src) if i not in inv]}, open(os.path.join(HERE, "vnl_keymap.json"), "w"), indent=0)
    print("vnl_keymap.json: %d mapped, %d dropped" % (len(pairs), len(src) - len(pairs)))


def main():
    torch.set_num_threads(8)
    criteria, metrics, FCRN = _import_reference()
    only = set(sys.argv[1:])
    want = lambda name: not only or name in only
    if want("losses"):
        gen_losses(criteria)
    if want("vnl"):
        gen_vnl(criteria)
    if want("stdepth"):
        gen_stdepth(criteria)
    if want("metrics"):
        gen_metrics(metrics)
    if want("ssim"):
        gen_ssim()
    if want("upproj"):
        gen_upproj(FCRN)
    if want("fcrn"):
        gen_fcrn(criteria, metrics, FCRN)
        gen_fcrn_conditioned(criteria, metrics, FCRN)
        gen_fcrn_decoders(criteria, metrics, FCRN)
        gen_fcrn_basic_trunks(criteria, metrics, FCRN)
        gen_fcrn_in_channels(criteria, metrics, FCRN)
    if want("vnl_net"):
        gen_vnl_net(criteria)
    if want("vnl_mbv2"):
        gen_vnl_mobilenet(criteria)
    if want("midas_net"):
        gen_midas_net(criteria)
    if want("bts_net"):
        gen_bts_net(criteria)
    if want("bts_cond"):
        gen_bts_conditioned(criteria, metrics)
    if want("bts_resnet"):
        gen_bts_resnet(criteria)
    if want("bts_imgres"):
        gen_bts_image_residuals(criteria)
    if want("bts_curve"):
        gen_bts_curve(criteria)
    if want("offgrid"):
        gen_offgrid(criteria, metrics, FCRN)
    if want("eigen"):
        gen_eigen(criteria)
    if want("dorn_net"):
        gen_dorn_net(criteria)
    if want("mynet"):
        gen_mynet(criteria)
    if want("augment"):
        gen_augment()
    if want("vnl_keymap"):
        gen_vnl_keymap()


if __name__ == "__main__":
    main()
