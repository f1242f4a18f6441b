"""GPU parity of the streaming kernels (BatchNorm, maxpool, bilinear+sigmoid, SILog, metrics,
Adam, packing, stem/head convs) against the CPU oracle / plain torch fp32 ops.
bf16-output kernels: |hip - ref| <= 2^-8|ref| + 2^-8 rms(ref); fp32 kernels: rtol 1e-4/1e-5."""
import numpy as np
import pytest
import torch

from mono_depth_estimation_amd.ops import ACT_DTYPE as ACT        # the library's 16-bit storage type (bf16; fp16 under MDE_ACT_DTYPE=fp16)
import torch.nn.functional as F

from oracle import losses as OL
from oracle import metrics as OM
from oracle import weights as W

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(ACT).to(torch.float32)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(ACT).cuda()


def _nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def _close_bf16(got, ref, what, tol=2.0 ** -8):
    err = (got - ref).abs()
    bound = tol * ref.abs() + tol * ref.pow(2).mean().sqrt()
    bad = (~(err <= bound)).sum().item()          # NaN/Inf compare False: they count as bad
    assert bad == 0, "%s: %d/%d outside tolerance, max err %.4g" % (what, bad, ref.numel(), err.max().item())


# ------------------------------------------------------------------------------------ BatchNorm
@pytest.mark.parametrize("N,H,Wd,C,relu,res", [
    (2, 9, 11, 64, True, 0), (3, 7, 5, 256, False, 0), (2, 6, 6, 128, True, 1), (2, 5, 9, 64, True, 2),
    (1, 3, 4, 2048, True, 1),
])
def test_bn_train_forward_backward(N, H, Wd, C, relu, res):
    """stats -> finalize -> apply and the three backward passes vs nn.BatchNorm2d autograd.
    res: 0 none, 1 identity residual, 2 residual through a second BN (up-projection join)."""
    from mono_depth_estimation_amd import ops
    M = N * H * Wd
    x = _bf(W.normal(7, "x", (N, C, H, Wd), 1.5, 0.3)).requires_grad_(True)
    r = _bf(W.normal(7, "r", (N, C, H, Wd))).requires_grad_(True)
    bn, bn2 = torch.nn.BatchNorm2d(C), torch.nn.BatchNorm2d(C)
    for i, b in enumerate((bn, bn2)):
        b.weight.data = W.normal(7, "g%d" % i, (C,), 0.2, 1.0)
        b.bias.data = W.normal(7, "b%d" % i, (C,), 0.2)
        b.running_mean.data = W.normal(7, "rm%d" % i, (C,), 0.2)
        b.running_var.data = W.uniform(7, "rv%d" % i, (C,), 0.5, 1.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    y = bn(x)
    if res == 1:
        y = y + r
    elif res == 2:
        y = y + bn2(r)
    if relu:
        y = F.relu(y)
    dy = _bf(W.normal(7, "dy", tuple(y.shape)))
    y.backward(dy)

    dev = "cuda"
    xd, rd, dyd = _nhwc(x.detach()), _nhwc(r.detach()), _nhwc(dy)
    part = ops.new_stat_buffer(C)
    f = lambda t: t.detach().clone().to(dev)
    gamma, beta, rmean, rvar = f(bn.weight), f(bn.bias), rm0.to(dev), rv0.to(dev)
    scale, shift, smean, srstd = (torch.empty(C, device=dev) for _ in range(4))
    ops.bn_stats(xd, M, C, C, part)
    ops.bn_finalize(part, M, C, gamma, beta, rmean, rvar, 0.1, 1e-5, scale, shift, smean, srstd)
    assert float(part.abs().max()) == 0.0, "finalize must re-zero the partial buffer"
    out = torch.empty_like(xd)
    bits = torch.zeros(M * C // 8, dtype=torch.uint8, device=dev) if (relu and res) else None   # packed ReLU mask
    if res == 2:
        part2 = ops.new_stat_buffer(C)
        g2, b2 = f(bn2.weight), f(bn2.bias)
        sc2, sh2, sm2, sr2 = (torch.empty(C, device=dev) for _ in range(4))
        ops.bn_stats(rd, M, C, C, part2)
        ops.bn_finalize(part2, M, C, g2, b2, None, None, 0.1, 1e-5, sc2, sh2, sm2, sr2)
        ops.bn_apply(xd, C, scale, shift, out, C, M, C, relu, r=rd, ldr=C, rscale=sc2, rshift=sh2, relu_bits=bits)
    elif res == 1:
        ops.bn_apply(xd, C, scale, shift, out, C, M, C, relu, r=rd, ldr=C, relu_bits=bits)
    else:
        ops.bn_apply(xd, C, scale, shift, out, C, M, C, relu)
    torch.cuda.synchronize()
    _close_bf16(_nchw(out), y.detach(), "bn fwd")
    assert torch.allclose(rmean.cpu(), bn.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(rvar.cpu(), bn.running_var, rtol=1e-4, atol=1e-5)

    # backward
    coef = torch.empty(3, C, device=dev)
    dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx = torch.empty_like(xd)
    dres = torch.empty_like(xd) if res == 1 else None
    mk = dict(mask_scale=scale, mask_shift=shift) if (relu and res == 0) else {}   # recomputed mask: never reads `out`
    if bits is not None:
        want = (out.float() > 0).reshape(M, C // 8, 8).to(torch.int32)
        want = (want << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8).reshape(-1)
        assert torch.equal(bits, want), "packed ReLU mask differs from (out > 0)"
        mk = dict(relu_bits=bits)
    ops.bn_bwd_reduce(dyd, C, None if mk else out, C, xd, C, smean, srstd, M, C, relu, part, **mk)
    ops.bn_bwd_finalize(part, M, C, gamma, srstd, dgamma, dbeta, coef)
    ops.bn_bwd_apply(dyd, C, None if mk else out, C, xd, C, smean, srstd, coef, M, C, relu, dx, C, dres=dres, ldres=C, **mk)
    torch.cuda.synchronize()
    _close_bf16(_nchw(dx), x.grad, "bn dx", tol=2.0 ** -7)
    tolp = dict(rtol=2e-2, atol=2e-2 * float(bn.weight.grad.abs().max()))
    assert torch.allclose(dgamma.cpu(), bn.weight.grad, **tolp)
    assert torch.allclose(dbeta.cpu(), bn.bias.grad, **tolp)
    if res == 1:
        _close_bf16(_nchw(dres), r.grad, "bn dres")
    if res == 2:
        dr = torch.empty_like(xd)
        ops.bn_bwd_reduce(dyd, C, None, C, rd, C, sm2, sr2, M, C, relu, part2, relu_bits=bits)
        ops.bn_bwd_finalize(part2, M, C, g2, sr2, None, None, coef)
        ops.bn_bwd_apply(dyd, C, None, C, rd, C, sm2, sr2, coef, M, C, relu, dr, C, relu_bits=bits)
        torch.cuda.synchronize()
        _close_bf16(_nchw(dr), r.grad, "bn2 dx", tol=2.0 ** -7)


def test_bn_eval_and_channel_slices():
    """Eval-mode scale/shift, and a BN site living in a channel slice of a wider tensor."""
    from mono_depth_estimation_amd import ops
    N, H, Wd, C, LD = 2, 5, 7, 64, 128
    M = N * H * Wd
    x = _bf(W.normal(8, "x", (N, LD, H, Wd)))
    bn = torch.nn.BatchNorm2d(C).eval()
    bn.weight.data = W.normal(8, "g", (C,), 0.2, 1.0)
    bn.bias.data = W.normal(8, "b", (C,), 0.2)
    bn.running_mean.data = W.normal(8, "rm", (C,), 0.2)
    bn.running_var.data = W.uniform(8, "rv", (C,), 0.5, 1.5)
    ref = F.relu(bn(x[:, C:]))
    xd = _nhwc(x)
    scale, shift = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_eval_scale_shift(bn.weight.cuda(), bn.bias.cuda(), bn.running_mean.cuda(), bn.running_var.cuda(), 1e-5, C,
                            scale, shift)
    out = torch.zeros(N, H, Wd, C, dtype=ACT, device="cuda")
    ops.bn_apply(xd[..., C:], LD, scale, shift, out, C, M, C, True)
    torch.cuda.synchronize()
    _close_bf16(_nchw(out), ref, "bn eval slice")


# ------------------------------------------------------------------------------------ maxpool
@pytest.mark.parametrize("N,H,Wd,C", [(2, 12, 16, 64), (1, 9, 11, 64)])
def test_maxpool(N, H, Wd, C):
    from mono_depth_estimation_amd import ops
    x = F.relu(_bf(W.normal(9, "x", (N, C, H, Wd)))).requires_grad_(True)   # post-ReLU: many ties at 0
    y = F.max_pool2d(x, 3, 2, 1)
    dy = _bf(W.normal(9, "dy", tuple(y.shape)))
    y.backward(dy)
    OH, OW = y.shape[2:]
    xd, dyd = _nhwc(x.detach()), _nhwc(dy)
    out = torch.empty(N, OH, OW, C, dtype=ACT, device="cuda")
    idx = torch.empty(N, OH, OW, C, dtype=torch.uint8, device="cuda")
    dx = torch.empty_like(xd)
    ops.maxpool_fwd(xd, out, idx, N, H, Wd, C)
    ops.maxpool_bwd(dyd, idx, dx, N, H, Wd, C)
    torch.cuda.synchronize()
    assert torch.equal(_nchw(out), y.detach())
    _close_bf16(_nchw(dx), x.grad, "maxpool bwd")


# ------------------------------------------------------------------------------------ bilinear + sigmoid
@pytest.mark.parametrize("N,C,H,Wd,OH,OW", [(2, 1, 12, 16, 24, 32), (1, 3, 7, 9, 20, 13), (2, 1, 6, 8, 6, 8)])
def test_upsample_sigmoid(N, C, H, Wd, OH, OW):
    from mono_depth_estimation_amd import ops
    x = W.normal(10, "x", (N, C, H, Wd)).requires_grad_(True)
    y = torch.sigmoid(F.interpolate(x, size=(OH, OW), mode="bilinear", align_corners=True))
    dy = W.normal(10, "dy", tuple(y.shape))
    y.backward(dy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda()      # fp32 NHWC
    out = torch.empty(N, C, OH, OW, device="cuda")
    dx = torch.empty_like(xd)
    ops.upsample_sigmoid_fwd(xd, out, N, H, Wd, C, OH, OW)
    ops.upsample_sigmoid_bwd(dy.cuda(), out, dx, N, H, Wd, C, OH, OW)
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), y.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(dx.cpu().permute(0, 3, 1, 2), x.grad, rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------ SILog / metrics
@pytest.mark.parametrize("lam", [0.85, 0.5])
def test_silog_golden(golden, lam):
    """HIP SILog vs the reference's own output (tests/golden/losses.npz) and the oracle."""
    from mono_depth_estimation_amd import ops
    g = golden("losses")
    est, gt = torch.from_numpy(g["g1_est"]).cuda(), torch.from_numpy(g["g1_gt"]).cuda()
    ws, loss, grad = ops.silog_ws(), torch.empty(1, device="cuda"), torch.empty_like(est)
    ops.silog_fwd(est, gt, lam, ws, loss)
    ops.silog_bwd(est, gt, lam, ws, None, grad)
    torch.cuda.synchronize()
    assert np.allclose(loss.item(), g["g1_silog_%g" % lam], rtol=1e-5)
    assert np.allclose(grad.cpu().numpy(), g["g1_silog_%g_grad" % lam], rtol=1e-4, atol=1e-8)
    gs = torch.full((1,), 0.25, device="cuda")
    ops.silog_bwd(est, gt, lam, ws, gs, grad)
    assert np.allclose(grad.cpu().numpy(), 0.25 * g["g1_silog_%g_grad" % lam], rtol=1e-4, atol=1e-8)


def test_silog_large_matches_oracle():
    from mono_depth_estimation_amd import ops
    est = W.uniform(11, "est", (4, 1, 480, 640), 0.05, 1.0)
    _, gt = W.synthetic_batch(11, 4, 480, 640)
    e = est.clone().requires_grad_(True)
    ref = OL.silog(e, gt, 0.85)
    ref.backward()
    ed, gd = est.cuda(), gt.cuda()
    ws, loss, grad = ops.silog_ws(), torch.empty(1, device="cuda"), torch.empty_like(ed)
    ops.silog_fwd(ed, gd, 0.85, ws, loss)
    ops.silog_bwd(ed, gd, 0.85, ws, None, grad)
    assert np.allclose(loss.item(), ref.item(), rtol=1e-5)
    assert torch.allclose(grad.cpu(), e.grad, rtol=2e-3, atol=1e-9)


_MASKED = {"masked_l1": ("MaskedL1Loss", OL.masked_l1), "masked_mse": ("MaskedMSELoss", OL.masked_mse),
           "berhu": ("berHuLoss", OL.berhu), "masked_depth": ("MaskedDepthLoss", OL.masked_depth)}


@pytest.mark.parametrize("name", sorted(_MASKED))
def test_masked_losses_golden(golden, name):
    """criteria.MaskedL1Loss / MaskedMSELoss / berHuLoss / MaskedDepthLoss (HIP) against the values and
    gradients the reference's own classes produced (tests/golden/losses.npz, minted by gen_golden.py)."""
    from mono_depth_estimation_amd import criteria
    g = golden("losses")
    pred = torch.from_numpy(g["g2_pred"]).cuda().requires_grad_(True)
    tgt = torch.from_numpy(g["g2_tgt"]).cuda()
    crit = getattr(criteria, _MASKED[name][0])()
    loss = crit(pred, tgt)
    (0.5 * loss).backward()                                   # upstream gradient != 1 on purpose
    assert loss.dim() == 0 and crit.loss is loss
    assert np.allclose(loss.item(), g["g2_" + name], rtol=2e-5), (loss.item(), g["g2_" + name])
    ref = 0.5 * g["g2_" + name + "_grad"]
    assert np.allclose(pred.grad.cpu().numpy(), ref, rtol=2e-4, atol=1e-6 * np.abs(ref).max())


@pytest.mark.parametrize("name", sorted(_MASKED))
def test_masked_losses_large_match_oracle(name):
    """Bench-sized maps (8 x 1 x 480 x 640, 10 % invalid pixels) against the oracle's autograd."""
    from mono_depth_estimation_amd import criteria
    pred = W.uniform(31, "pred", (8, 1, 480, 640), 0.05, 1.0)
    _, tgt = W.synthetic_batch(31, 8, 480, 640)
    p = pred.clone().requires_grad_(True)
    ref = _MASKED[name][1](p, tgt)
    ref.backward()
    pd = pred.cuda().requires_grad_(True)
    loss = getattr(criteria, _MASKED[name][0])()(pd, tgt.cuda())
    loss.backward()
    assert np.allclose(loss.item(), ref.item(), rtol=2e-5), (loss.item(), ref.item())
    assert torch.allclose(pd.grad.cpu(), p.grad, rtol=2e-3, atol=1e-6 * float(p.grad.abs().max()))


def test_masked_losses_edge_cases():
    from mono_depth_estimation_amd import criteria
    pred = W.uniform(32, "pred", (2, 5, 7), 0.1, 1.0).cuda().requires_grad_(True)      # 3-D input, odd sizes
    tgt = W.uniform(32, "tgt", (2, 5, 7), 0.1, 1.0)
    tgt[0, :2] = 0.0
    ref = OL.masked_depth(pred.detach().cpu().requires_grad_(True), tgt)
    loss = criteria.MaskedDepthLoss()(pred, tgt.cuda())
    assert np.allclose(loss.item(), ref.item(), rtol=2e-5)
    empty = torch.zeros(2, 1, 4, 4, device="cuda")                                       # no valid pixel: NaN, as the reference
    assert torch.isnan(criteria.MaskedL1Loss()(pred.detach().reshape(2, 1, 5, 7)[:, :, :4, :4].contiguous(), empty))
    with pytest.raises(RuntimeError):
        criteria.berHuLoss()(torch.ones(1, 1, 2, 2), torch.ones(1, 1, 2, 2))           # CPU tensors: no fallback
    # half / bf16 / non-contiguous predictions (AMP users, channel slices as in base_module.py:156): gradient comes
    # back in the prediction's dtype and layout
    base = W.uniform(32, "p16", (2, 3, 6, 8), 0.1, 1.0).cuda()
    tg = W.uniform(32, "t16", (2, 1, 6, 8), 0.1, 1.0).cuda()
    for mk in (lambda: criteria.silog_loss(0.85), criteria.MaskedL1Loss, criteria.MaskedDepthLoss, criteria.MidasLoss,
               criteria.TrimmedProcrustesLoss):
        for dt in (torch.float16, torch.bfloat16, torch.float32):
            p = base.detach().to(dt).clone().requires_grad_(True)
            loss = mk()(p[:, 1:2], tg)                                                     # a non-contiguous channel slice
            loss.backward()
            assert p.grad.dtype == dt and torch.isfinite(p.grad.float()).all()
            assert float(p.grad[:, 0].abs().sum()) == 0 and float(p.grad[:, 1].abs().sum()) > 0


@pytest.mark.parametrize("loss", ["mse", "l1", "trim", "ssimse", "ssil1", "ssitrim"])
def test_midas_loss_golden(golden, loss):
    """criteria.MidasLoss (HIP) against the reference's own values and gradients (losses.npz, alpha 0.5, 4 scales)."""
    from mono_depth_estimation_amd import criteria
    g = golden("losses")
    pred = torch.from_numpy(g["g2_pred"]).cuda().requires_grad_(True)
    tgt = torch.from_numpy(g["g2_tgt"]).cuda()
    out = criteria.MidasLoss(alpha=0.5, loss=loss)(pred, tgt)
    (2.0 * out).backward()
    assert np.allclose(out.item(), g["g2_midas_" + loss], rtol=5e-5), (out.item(), g["g2_midas_" + loss])
    ref = 2.0 * g["g2_midas_" + loss + "_grad"]
    err = np.abs(pred.grad.cpu().numpy() - ref)
    assert err.max() <= 2e-3 * np.abs(ref).max() and err.mean() <= 2e-5 * np.abs(ref).max(), (err.max(), err.mean(), np.abs(ref).max())


def test_scale_shift_and_gradient_loss_golden(golden):
    from mono_depth_estimation_amd import criteria
    g = golden("losses")
    pred, tgt = torch.from_numpy(g["g2_pred"]).cuda(), torch.from_numpy(g["g2_tgt"]).cuda()
    s, h = criteria.compute_scale_and_shift(pred, tgt)
    assert np.allclose(s.cpu().numpy(), g["g2_scale"], rtol=1e-4) and np.allclose(h.cpu().numpy(), g["g2_shift"], rtol=1e-4, atol=1e-6)
    mask = (tgt > 0).float()
    for red, key in (("batch-based", "g2_gradient_batch"), ("image-based", "g2_gradient_image")):
        p = pred.clone().requires_grad_(True)
        out = criteria.GradientLoss(scales=4, reduction=red)(p, tgt, mask)
        out.backward()
        assert np.allclose(out.item(), g[key], rtol=5e-5), (red, out.item(), g[key])
        ref = g[key + "_grad"]
        assert np.abs(p.grad.cpu().numpy().reshape(ref.shape) - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-9


@pytest.mark.parametrize("loss", ["ssimse", "ssitrim", "mse"])
def test_midas_loss_large_matches_oracle(loss):
    """MiDaS-sized maps (4 x 384 x 384, 10 % invalid) against the oracle's autograd."""
    from mono_depth_estimation_amd import criteria
    pred = W.uniform(41, "pred", (4, 1, 384, 384), 0.05, 1.0)
    _, tgt = W.synthetic_batch(41, 4, 384, 384)
    tgt = tgt * (1.0 + 0.5 * W.uniform(41, "ramp", (4, 1, 384, 384)))       # decorrelate scale from 1
    p = pred.clone().requires_grad_(True)
    ref = OL.midas_loss(p, tgt, alpha=0.5, loss=loss)
    ref.backward()
    pd = pred.cuda().requires_grad_(True)
    out = criteria.MidasLoss(alpha=0.5, loss=loss)(pd, tgt.cuda())
    out.backward()
    assert np.allclose(out.item(), ref.item(), rtol=1e-4), (out.item(), ref.item())
    err = (pd.grad.cpu() - p.grad).abs()
    scale = float(p.grad.abs().max())
    # sign() terms flip where two neighbouring residuals differ by rounding noise: a few isolated pixels
    assert float(err.mean()) <= 2e-5 * scale and float((err > 1e-3 * scale).float().mean()) < 1e-3, (float(err.max()), float(err.mean()), scale)


def test_trimmed_procrustes_golden_and_large(golden):
    """criteria.TrimmedProcrustesLoss (HIP: radix-select median, robust scale, chain rule through both) against the
    reference's golden value and gradient, then on MiDaS-sized maps against the oracle's autograd."""
    from mono_depth_estimation_amd import criteria
    g = golden("losses")
    pred = torch.from_numpy(g["g2_pred"]).cuda().requires_grad_(True)
    tgt = torch.from_numpy(g["g2_tgt"]).cuda()
    crit = criteria.TrimmedProcrustesLoss(alpha=0.5)
    out = crit(pred, tgt)
    out.backward()
    assert np.allclose(out.item(), g["g2_procrustes"], rtol=5e-5), (out.item(), g["g2_procrustes"])
    ref = g["g2_procrustes_grad"]
    err = np.abs(pred.grad.cpu().numpy() - ref)
    assert err.max() <= 2e-3 * np.abs(ref).max() and err.mean() <= 2e-5 * np.abs(ref).max(), (err.max(), err.mean(), np.abs(ref).max())
    # the normalised prediction the reference exposes as .prediction_ssi
    p3, t3 = torch.from_numpy(g["g2_pred"]).squeeze(1), torch.from_numpy(g["g2_tgt"]).squeeze(1)
    pn = OL.normalize_robust(p3, (t3 > 0).float())
    assert torch.allclose(crit.prediction_ssi.cpu(), pn, rtol=1e-4, atol=1e-5)
    # large
    pred = W.uniform(43, "pred", (3, 1, 384, 384), 0.05, 1.0)
    _, tgt = W.synthetic_batch(43, 3, 384, 384)
    p = pred.clone().requires_grad_(True)
    refl = OL.trimmed_procrustes(p, tgt, alpha=0.5)
    refl.backward()
    pd = pred.cuda().requires_grad_(True)
    outl = criteria.TrimmedProcrustesLoss(alpha=0.5)(pd, tgt.cuda())
    outl.backward()
    assert np.allclose(outl.item(), refl.item(), rtol=1e-4), (outl.item(), refl.item())
    err = (pd.grad.cpu() - p.grad).abs()
    scale = float(p.grad.abs().max())
    assert float(err.mean()) <= 2e-5 * scale and float((err > 1e-3 * scale).float().mean()) < 1e-3, (float(err.max()), float(err.mean()), scale)


def test_metrics_golden(golden):
    from mono_depth_estimation_amd import ops
    g = golden("metrics")
    pred, tgt = torch.from_numpy(g["pred"]).cuda(), torch.from_numpy(g["tgt"]).cuda()
    ws, out = ops.metrics_ws(), torch.empty(len(OM.NAMES), device="cuda")
    ops.depth_metrics(pred, tgt, ws, out)
    got = out.cpu().numpy()
    ora = OM.compute(torch.from_numpy(g["pred"]), torch.from_numpy(g["tgt"]))
    for i, k in enumerate(OM.NAMES):
        if k in OM.PINNED:
            assert np.allclose(got[i], g[k], rtol=1e-5), k              # the reference's own value
        assert np.allclose(got[i], float(ora[k]), rtol=2e-5), k        # mae / mse / msle: by definition (oracle)


def test_metric_computation_accepts_the_reference_default_names():
    """train.py:67 asks for ['delta1', 'delta2', 'delta3', 'mse', 'mae', 'log10', 'rmse', 'ssim'] by default: all but
    'ssim' (a torchmetrics CPU call in the reference, metrics.py:63,123) are computed in the one device pass."""
    from mono_depth_estimation_amd import metrics
    pred, tgt = W.uniform(3, "p", (2, 1, 40, 56), 0.0, 1.2).cuda(), W.uniform(3, "t", (2, 1, 40, 56), -0.2, 1.0).cuda()
    names = ['delta1', 'delta2', 'delta3', 'mse', 'mae', 'log10', 'rmse', 'msle', 'sqrel', 'absrel']
    mc = metrics.MetricComputation(names)
    vals = mc.compute(pred, tgt)
    ora = OM.compute(pred.cpu(), tgt.cpu())
    for n, v in zip(names, vals):
        assert np.allclose(float(v), float(ora[n]), rtol=2e-5), n
    assert np.allclose(float(mc.avg("mae")), float(ora["mae"]), rtol=2e-5)
    with pytest.raises(NotImplementedError):
        metrics.MetricComputation(["psnr"])


def test_ssim_metric(golden):
    """'ssim' of train.py's default metric list (metrics.py:63,123): the device kernel against the oracle's restatement of
    torchmetrics 0.7.3's definition and against the anchor minted from the reference's own SSIM window (tests/golden/ssim.npz);
    several planes, sizes that are not multiples of the 16 x 64 tile, a prediction with values below the 1e-7 clamp."""
    from mono_depth_estimation_amd import metrics
    g = golden("ssim")
    pred, tgt = torch.from_numpy(g["pred"]).cuda(), torch.from_numpy(g["tgt"]).cuda()
    mc = metrics.MetricComputation(["delta1", "ssim", "absrel"])
    vals = mc.compute(pred, tgt)
    assert np.allclose(float(vals[1]), float(g["ssim_interior"]), rtol=1e-4), (float(vals[1]), float(g["ssim_interior"]))
    assert np.allclose(float(vals[0]), float(OM.compute(pred.cpu(), tgt.cpu())["delta1"]), rtol=2e-5)
    for shape, seed in (((2, 3, 33, 150), 5), ((1, 1, 11, 11), 6), ((4, 1, 96, 128), 7)):
        t = W.uniform(seed, "t", shape, 0.0, 1.0)
        p = 0.7 * t + 0.3 * W.uniform(seed, "p", shape, -0.2, 1.0)
        got = metrics.MetricComputation(["ssim"]).compute(p.cuda(), t.cuda())[0]
        ref = float(OM.ssim(p, t))
        assert abs(float(got) - ref) <= 1e-4 * max(abs(ref), 0.1), (shape, float(got), ref)
    with pytest.raises(RuntimeError, match="10 x 10"):
        metrics.MetricComputation(["ssim"]).compute(torch.rand(1, 1, 8, 40).cuda(), torch.rand(1, 1, 8, 40).cuda())


# ------------------------------------------------------------------------------------ optimiser plumbing
def test_adam_matches_torch():
    from mono_depth_estimation_amd import ops
    n = 100003
    p0, g1, g2 = W.normal(12, "p", (n,)), W.normal(12, "g1", (n,)), W.normal(12, "g2", (n,), 0.3)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    pd, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb = torch.empty(n, dtype=ACT, device="cuda")
    for step, g in enumerate((g1, g2), 1):
        p.grad = g.clone()
        opt.step()
        ops.adam_step(pd, (2.0 * g).cuda(), m, v, pb, n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.5, step)
    torch.cuda.synchronize()
    assert torch.allclose(pd.cpu(), p.detach(), rtol=1e-5, atol=1e-7)
    assert torch.equal(pb.cpu(), pd.cpu().to(ACT))


def test_adamw_and_sgd_match_torch():
    """The other optimiser configurations of SURVEY §8 row O: AdamW eps 1e-3 wd 1e-2 (bts.py:139-152) and SGD momentum
    0.9 wd 5e-4 (vnl.py:289-326), three steps each against torch.optim on the CPU."""
    from mono_depth_estimation_amd import ops
    n = 100003
    p0 = W.normal(15, "p", (n,))
    gs = [W.normal(15, "g%d" % i, (n,), 0.5) for i in range(3)]
    pa, ps = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    adamw = torch.optim.AdamW([pa], lr=1e-3, betas=(0.9, 0.999), eps=1e-3, weight_decay=1e-2)
    sgd = torch.optim.SGD([ps], lr=1e-2, momentum=0.9, weight_decay=5e-4)
    da, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ds, buf = p0.clone().cuda(), torch.zeros(n, device="cuda")
    ba, bs = torch.empty(n, dtype=ACT, device="cuda"), torch.empty(n, dtype=ACT, device="cuda")
    for step, g in enumerate(gs, 1):
        pa.grad, ps.grad = g.clone(), g.clone()
        adamw.step()
        sgd.step()
        ops.adamw_step(da, (4.0 * g).cuda(), m, v, ba, n, 1e-3, 0.9, 0.999, 1e-3, 1e-2, 0.25, step)
        ops.sgd_step(ds, (4.0 * g).cuda(), buf, bs, n, 1e-2, 0.9, 5e-4, 0.25)
    torch.cuda.synchronize()
    assert torch.allclose(da.cpu(), pa.detach(), rtol=1e-5, atol=1e-7)
    assert torch.allclose(ds.cpu(), ps.detach(), rtol=1e-5, atol=1e-7)
    assert torch.equal(ba.cpu(), da.cpu().to(ACT)) and torch.equal(bs.cpu(), ds.cpu().to(ACT))


def test_cast_and_pack():
    from mono_depth_estimation_amd import ops
    O, T, I = 70, 9, 130
    w = W.normal(13, "w", (O, T, I))
    wd = w.cuda()
    c = torch.empty(O, T, I, dtype=ACT, device="cuda")
    t = torch.empty(I, T, O, dtype=ACT, device="cuda")
    ops.cast_bf16(wd, c)
    ops.pack_wt(wd, t, O, T, I)
    torch.cuda.synchronize()
    assert torch.equal(c.cpu(), w.to(ACT))
    assert torch.equal(t.cpu(), w.permute(2, 1, 0).contiguous().to(ACT))
    # batched form: several weights of one flat buffer (with gaps between them) in one launch
    shapes = [(70, 9, 130), (64, 1, 64), (33, 25, 40), (256, 1, 1024)]
    offs, total = [], 8
    for o, t_, i in shapes:
        offs.append(total)
        total += o * t_ * i + 24
    flat = W.normal(13, "flat", (total,)).cuda()
    dst = torch.full((total,), 7.0, dtype=ACT, device="cuda")
    jobs, nblocks = ops.pack_jobs([(off, o, t_, i) for off, (o, t_, i) in zip(offs, shapes)], "cuda")
    ops.pack_wt_batch(flat, dst, jobs, nblocks)
    torch.cuda.synchronize()
    covered = torch.zeros(total, dtype=torch.bool)
    for off, (o, t_, i) in zip(offs, shapes):
        n = o * t_ * i
        ref = flat[off:off + n].cpu().view(o, t_, i).permute(2, 1, 0).contiguous().to(ACT)
        assert torch.equal(dst[off:off + n].cpu().view(i, t_, o), ref)
        covered[off:off + n] = True
    assert (dst.cpu()[~covered].float() == 7.0).all(), "wrote outside the weights"
    x = W.normal(13, "x", (2, 5, 6, 7))
    xd = torch.empty(2, 6, 7, 5, dtype=ACT, device="cuda")
    ops.nchw_to_nhwc_bf16(x.cuda(), xd)
    back = torch.empty(2, 5, 6, 7, device="cuda")
    ops.nhwc_bf16_to_nchw(xd, back)
    assert torch.equal(back.cpu(), _bf(x))


@pytest.mark.parametrize("N,H,Wd,C", [(2, 9, 11, 64), (1, 5, 7, 2048), (3, 16, 20, 256)])
def test_bn_join_backward_two_sites(N, H, Wd, C):
    """mde_bn_bwd_reduce2 / apply2 (a residual join of two BN outputs, one masked gradient) against the
    single-site kernels run twice: sums to 1e-4 relative (fp32 atomics, order not fixed), dx to one bf16 ulp
    of their scale."""
    from mono_depth_estimation_amd import ops
    M = N * H * Wd
    ld_b = C + 24                                             # second site is a channel slice of a wider tensor
    dout = _bf(W.normal(21, "dout", (M, C))).to(ACT).cuda()
    xa = _bf(W.normal(21, "xa", (M, C)) * 1.5 + 0.2).to(ACT).cuda()
    xb_full = _bf(W.normal(21, "xb", (M, ld_b)) * 0.7 - 0.1).to(ACT).cuda()
    xb = xb_full[:, 8:8 + C]
    bits = (W.uniform(21, "bits", (M, C // 8)) * 256).to(torch.uint8).cuda()
    stats = [[(W.normal(21, "m%d" % i, (C,)) * 0.3).cuda(), (W.uniform(21, "r%d" % i, (C,)) + 0.5).cuda(),
              (W.normal(21, "g%d" % i, (C,)) * 0.5 + 1.0).cuda()] for i in range(2)]
    xs, lds = [xa, xb], [C, ld_b]
    # reference: the single-site kernels, once per site
    ref_dx, ref_coef, ref_dg, ref_db = [], [], [], []
    for i in range(2):
        part = ops.new_stat_buffer(C)
        coef, dg, db = torch.empty(3, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dx = torch.empty(M, C, dtype=ACT, device="cuda")
        ops.bn_bwd_reduce(dout, C, None, 0, xs[i], lds[i], stats[i][0], stats[i][1], M, C, True, part, relu_bits=bits)
        ops.bn_bwd_finalize(part, M, C, stats[i][2], stats[i][1], dg, db, coef)
        ops.bn_bwd_apply(dout, C, None, 0, xs[i], lds[i], stats[i][0], stats[i][1], coef, M, C, True, dx, C, relu_bits=bits)
        ref_dx.append(dx); ref_coef.append(coef); ref_dg.append(dg); ref_db.append(db)
    pa, pb = ops.new_stat_buffer(C), ops.new_stat_buffer(C)
    ops.bn_bwd_reduce2(dout, C, xa, C, xb, ld_b, stats[0][0], stats[0][1], stats[1][0], stats[1][1], bits, M, C, pa, pb)
    got_dx = [torch.empty(M, C, dtype=ACT, device="cuda"), torch.full((M, C + 8), 5.0, dtype=ACT, device="cuda")]
    coefs = []
    for i, part in enumerate((pa, pb)):
        coef, dg, db = torch.empty(3, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        ops.bn_bwd_finalize(part, M, C, stats[i][2], stats[i][1], dg, db, coef)
        coefs.append(coef)
        scale = float(ref_dg[i].abs().max()) + float(ref_db[i].abs().max())
        assert torch.allclose(dg, ref_dg[i], rtol=1e-4, atol=1e-4 * scale) and torch.allclose(db, ref_db[i], rtol=1e-4, atol=1e-4 * scale)
        assert torch.allclose(coef, ref_coef[i], rtol=1e-4, atol=1e-6)
    ops.bn_bwd_apply2(dout, C, xa, C, xb, ld_b, stats[0][0], stats[0][1], stats[1][0], stats[1][1], bits, coefs[0], coefs[1],
                      M, C, got_dx[0], C, got_dx[1], C + 8)
    torch.cuda.synchronize()
    _close_bf16(got_dx[0].float().cpu(), ref_dx[0].float().cpu(), "join dx a")
    _close_bf16(got_dx[1][:, :C].float().cpu(), ref_dx[1].float().cpu(), "join dx b")
    assert (got_dx[1][:, C:].float() == 5.0).all(), "wrote outside its channel slice"


# ------------------------------------------------------------------------------------ stem / head convs
@pytest.mark.parametrize("N,H,Wd,CO", [(2, 32, 48, 64), (1, 30, 200, 64), (3, 62, 260, 64), (40, 64, 256, 64),
                                        (2, 32, 48, 96), (3, 62, 260, 96), (1, 30, 200, 96)])
def test_stem_conv(N, H, Wd, CO):
    """CO = 96: densenet161's conv0 (two launches of 48 channels into one 96-channel tensor, one statistics buffer)."""
    from mono_depth_estimation_amd import ops
    x = W.uniform(14, "x", (N, 3, H, Wd))
    # the kernel multiplies bf16 weights (as every MFMA conv here) with the image split into
    # bf16 hi+lo parts (~fp32): the reference uses bf16-representable weights and the fp32 image
    w = _bf(W.normal(14, "w", (CO, 3, 7, 7), (2.0 / (49 * 64)) ** 0.5)).requires_grad_(True)
    y = F.conv2d(x, w, stride=2, padding=3)
    dy = _bf(W.normal(14, "dy", tuple(y.shape)))
    y.backward(dy)
    w_ohwi = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    OH, OW = y.shape[2:]
    out = torch.empty(N, OH, OW, CO, dtype=ACT, device="cuda")
    dw = torch.zeros(CO, 7, 7, 3, device="cuda")
    part = ops.new_stat_buffer(CO)
    ops.stem_conv_fwd(x.cuda(), w_ohwi, out, part, CO)
    ops.stem_conv_wgrad(x.cuda(), _nhwc(dy), dw, CO)
    torch.cuda.synchronize()
    _close_bf16(_nchw(out), y.detach(), "stem fwd")
    # BatchNorm partial sums from the epilogue (fp32 results): sum and sum of squares per channel
    yd, st = y.detach().double(), part.sum(0).double().cpu()
    s1, s2 = yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))
    assert torch.allclose(st[0], s1, rtol=1e-3, atol=1e-3 * float(yd.abs().sum((0, 2, 3)).max()))
    assert torch.allclose(st[1], s2, rtol=2e-3)
    ref = w.grad.permute(0, 2, 3, 1)
    assert torch.allclose(dw.cpu(), ref, rtol=1e-3, atol=1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("N,H,Wd,Cout,Cin", [(2, 10, 12, 1, 64), (1, 7, 9, 1, 64), (3, 33, 41, 1, 64), (1, 7, 9, 20, 64), (1, 6, 6, 3, 64),
                                             (2, 10, 12, 1, 32), (3, 33, 41, 1, 32), (1, 7, 9, 1, 32), (1, 7, 9, 5, 32), (2, 9, 11, 1, 16)])
def test_head_conv(N, H, Wd, Cout, Cin):
    """(Cin = 32, Cout = 1: BTS' get_depth, the four-lanes-per-pixel instances of the FCRN head kernels.)"""
    from mono_depth_estimation_amd import ops
    x = _bf(W.normal(15, "x", (N, Cin, H, Wd))).requires_grad_(True)
    w = W.normal(15, "w", (Cout, Cin, 3, 3), 0.05).requires_grad_(True)
    y = F.conv2d(x, w, padding=1)
    dy = W.normal(15, "dy", tuple(y.shape))
    y.backward(dy)
    xd = _nhwc(x.detach())
    w_ohwi = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.empty(N, H, Wd, Cout, device="cuda")
    ops.head_conv_fwd(xd, w_ohwi, out, N, H, Wd, Cin, Cout)
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dx = torch.empty_like(xd)
    dw = torch.zeros(Cout, 3, 3, Cin, device="cuda")
    ops.head_conv_bwd(xd, w_ohwi, dyd, dx, dw, N, H, Wd, Cin, Cout)
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu().permute(0, 3, 1, 2), y.detach(), rtol=1e-4, atol=1e-5)
    _close_bf16(_nchw(dx), x.grad, "head dgrad")
    ref = w.grad.permute(0, 2, 3, 1)
    assert torch.allclose(dw.cpu(), ref, rtol=1e-3, atol=1e-4 * float(ref.abs().max()))


def test_pixel_shuffle2_matches_torch():
    """nn.PixelShuffle(2) on NHWC bf16 channel slices, and its inverse (the gradient permutation)."""
    from mono_depth_estimation_amd import ops
    N, h, w, C = 2, 5, 7, 24
    src = W.normal(44, "ps", (N, 4 * C + 16, h, w))                       # NCHW reference; 16 extra channels = a wider parent
    ref = F.pixel_shuffle(_bf(src[:, 8:8 + 4 * C]), 2)                    # [N, C, 2h, 2w]
    s_nhwc = _nhwc(src)                                                   # [N, h, w, 4C+16]
    dst = torch.zeros(N, 2 * h, 2 * w, C + 8, dtype=ACT, device="cuda")
    ops.pixel_shuffle2(s_nhwc[..., 8:8 + 4 * C], s_nhwc.shape[-1], dst[..., :C], dst.shape[-1], N, h, w, C)
    torch.cuda.synchronize()
    assert torch.equal(_nchw(dst[..., :C]), ref) and float(dst[..., C:].float().abs().max()) == 0.0
    back = torch.zeros_like(s_nhwc)
    ops.pixel_shuffle2(back[..., 8:8 + 4 * C], back.shape[-1], dst[..., :C], dst.shape[-1], N, h, w, C, inverse=True)
    torch.cuda.synchronize()
    assert torch.equal(back[..., 8:8 + 4 * C], s_nhwc[..., 8:8 + 4 * C])
    assert float(back[..., :8].float().abs().max()) == 0.0 and float(back[..., 8 + 4 * C:].float().abs().max()) == 0.0




@pytest.mark.parametrize("N,H,Wd,C,k,s,y0,x0,Hv,Wv", [
    (2, 12, 20, 64, 2, 2, 0, 0, 12, 20),       # VGG's MaxPool2d(2, 2)
    (1, 15, 21, 16, 2, 2, 0, 0, 15, 21),       # odd sizes: the last row / column is outside every window
    (2, 17, 23, 32, 3, 2, 2, 2, 11, 17),       # MaxPool2d(3, 2) cropped [1:-1]: a view at (2, 2) (Eigen scale 2)
    (2, 16, 21, 24, 3, 1, 2, 2, 11, 16),       # [2:-3] then MaxPool2d(3, 1): overlapping windows (Eigen scale 3)
])
def test_maxpool_view(N, H, Wd, C, k, s, y0, x0, Hv, Wv):
    """mde_maxpool_view_fwd / _bwd against torch's MaxPool2d on the cropped tensor, values exact (bf16 in, bf16 out) and the
    gradient routed to the same argmax; pixels outside the view get exactly zero."""
    from mono_depth_estimation_amd import ops
    x = W.normal(11, "x", (N, C, H, Wd)).to(ACT).float()
    xi = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xi[:, :, y0:y0 + Hv, x0:x0 + Wv], k, s)
    OH, OW = ref.shape[2:]
    dy = W.normal(11, "dy", tuple(ref.shape)).to(ACT).float()
    ref.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(ACT).cuda()
    out = torch.empty(N, OH, OW, C, dtype=ACT, device="cuda")
    idx = torch.empty(N, OH, OW, C, dtype=torch.uint8, device="cuda")
    ops.maxpool_view_fwd(xd[:, y0:, x0:], C, Wd, H * Wd, Hv, Wv, out, C, idx, N, C, k, s)
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), ref.detach())
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(ACT).cuda()
    dx = torch.zeros(N, H, Wd, C, dtype=ACT, device="cuda")
    ops.maxpool_view_bwd(dyd, C, idx, dx[:, y0:, x0:], C, Wd, H * Wd, Hv, Wv, N, C, k, s, accumulate=True)
    got = dx.float().cpu().permute(0, 3, 1, 2)
    assert torch.allclose(got, xi.grad.to(ACT).float(), rtol=2.0 ** -7, atol=1e-6), float((got - xi.grad).abs().max())


# ------------------------------------------------------------------------------------ finalize work inside the streaming launches
def _fin(part, ld, M, gamma, beta, rmean, rvar, mom, eps, scale, shift, smean, srstd, zero, mean=None, var=None):
    from mono_depth_estimation_amd import _lib
    dp = lambda t: t.data_ptr() if t is not None else None
    f = _lib.BnFin(dp(part), ld, dp(mean), dp(var), M, dp(gamma), dp(beta), dp(rmean), dp(rvar), mom, eps, dp(scale), dp(shift), dp(smean),
                   dp(srstd), dp(zero), zero.numel() if zero is not None else 0)
    f._keep = (part, gamma, beta, rmean, rvar, scale, shift, smean, srstd, zero, mean, var)
    return f


@pytest.mark.parametrize("N,H,Wd,C,relu,res", [(2, 9, 11, 64, True, 0), (3, 7, 5, 256, False, 0), (2, 6, 6, 128, True, 1), (2, 5, 9, 64, True, 2),
                                               (1, 3, 4, 2048, True, 1), (4, 40, 48, 48, True, 0)])
def test_bn_fused_finalize_equals_the_separate_kernels(N, H, Wd, C, relu, res):
    """mde_bn_apply_fin / mde_bn_bwd_apply_fin / mde_bn_bwd_apply2_fin against finalize + apply, the same arithmetic in another
    launch: every output BIT for bit (activation, mask bits, scale / shift / saved statistics, running statistics, dx, dgamma /
    dbeta), the sums the launch read left untouched, the OTHER direction's buffer zeroed; a site whose forward sums are a column
    range of a wider buffer (the up-projection's halves) and given batch moments (DenseNet) included."""
    from mono_depth_estimation_amd import _lib, ops
    dev, M = "cuda", N * H * Wd
    x = _bf(W.normal(31, "x", (N, C, H, Wd), 1.5, 0.3))
    r = _bf(W.normal(31, "r", (N, C, H, Wd)))
    dy = _bf(W.normal(31, "dy", (N, C, H, Wd)))
    xd, rd, dyd = _nhwc(x), _nhwc(r), _nhwc(dy)
    mk = lambda tag: [W.normal(31, tag + "g", (C,), 0.2, 1.0).to(dev), W.normal(31, tag + "b", (C,), 0.2).to(dev),
                      W.normal(31, tag + "m", (C,), 0.2).to(dev), W.uniform(31, tag + "v", (C,), 0.5, 1.5).to(dev)]
    wide = 2 * C if C <= 1024 else C                       # forward sums in columns [c0, c0 + C) of a [slots][2][wide] buffer
    c0 = wide - C

    sums = {}
    for tag, t in (("a", xd), ("r", rd)):                 # ONE set of sums for both forms (float atomics: a second pass would differ in order)
        sums[tag] = ops.new_stat_buffer(C)
        ops.bn_stats(t, M, C, C, sums[tag])                # (the conv epilogue's role)

    def forward(fused):
        outs = {}
        sites = []
        for tag, t in (("a", xd), ("r", rd)):
            g, b, rm, rv = mk(tag)
            part = torch.zeros(ops.stat_slots(), 2, wide, device=dev)
            part[:, :, c0:] = sums[tag]
            sites.append(dict(g=g, b=b, rm=rm, rv=rv, part=part, sc=torch.empty(C, device=dev), sh=torch.empty(C, device=dev),
                              sm=torch.empty(C, device=dev), sr=torch.empty(C, device=dev), pb=torch.full((ops.stat_slots(), 2, C), 3.0, device=dev)))
        a, b2 = sites
        out = torch.empty_like(xd)
        bits = torch.zeros(M * C // 8, dtype=torch.uint8, device=dev) if (relu and res) else None
        if fused:
            fa = _fin(a["part"].view(-1)[c0:], wide, M, a["g"], a["b"], a["rm"], a["rv"], 0.1, 1e-5, a["sc"], a["sh"], a["sm"], a["sr"], a["pb"])
            fb = _fin(b2["part"].view(-1)[c0:], wide, M, b2["g"], b2["b"], b2["rm"], b2["rv"], 0.1, 1e-5, b2["sc"], b2["sh"], b2["sm"], b2["sr"], b2["pb"])
            ops.bn_apply_fin(xd, C, fa, out, C, M, C, relu, r=rd if res else None, ldr=C if res else 0, fin_r=fb if res == 2 else None, relu_bits=bits)
        else:
            for s in (a, b2) if res == 2 else (a,):
                dense = s["part"][:, :, c0:].contiguous()
                ops.bn_finalize(dense, M, C, s["g"], s["b"], s["rm"], s["rv"], 0.1, 1e-5, s["sc"], s["sh"], s["sm"], s["sr"])
            ops.bn_apply(xd, C, a["sc"], a["sh"], out, C, M, C, relu, r=rd if res else None, ldr=C if res else 0,
                         rscale=b2["sc"] if res == 2 else None, rshift=b2["sh"] if res == 2 else None, relu_bits=bits)
        torch.cuda.synchronize()
        return out, bits, a, b2

    o1, bits1, a1, r1 = forward(False)
    part_before = None
    o2, bits2, a2, r2 = forward(True)
    assert torch.equal(o1, o2) and (bits1 is None or torch.equal(bits1, bits2))
    for s1, s2 in ((a1, a2), (r1, r2)) if res == 2 else ((a1, a2),):
        for k in ("sc", "sh", "sm", "sr", "rm", "rv"):
            assert torch.equal(s1[k], s2[k]), k
        assert float(s2["pb"].abs().max()) == 0.0, "the backward sums were to be zeroed"
        assert float(s2["part"][:, :, c0:].abs().max()) > 0.0 and float(s2["part"][:, :, :c0].abs().max() if c0 else 0.0) == 0.0
    # given batch moments (DenseNet's shared moments): the same launch against mde_bn_finalize_moments + apply
    mean, var = a1["sm"].clone(), (1.0 / a1["sr"] ** 2 - 1e-5).clamp(min=0)
    g, b, rm, rv = mk("a")
    sc, sh, sm, sr = (torch.empty(C, device=dev) for _ in range(4))
    ops.bn_finalize_moments(mean, var, M, C, g, b, rm, rv, 0.1, 1e-5, sc, sh, sm, sr)
    o3 = torch.empty_like(xd)
    ops.bn_apply(xd, C, sc, sh, o3, C, M, C, relu)
    g2, b2_, rm2, rv2 = mk("a")
    sc2, sh2, sm2, sr2 = (torch.empty(C, device=dev) for _ in range(4))
    o4 = torch.empty_like(xd)
    ops.bn_apply_fin(xd, C, _fin(None, 0, M, g2, b2_, rm2, rv2, 0.1, 1e-5, sc2, sh2, sm2, sr2, None, mean=mean, var=var), o4, C, M, C, relu)
    torch.cuda.synchronize()
    assert torch.equal(o3, o4) and torch.equal(sc, sc2) and torch.equal(sh, sh2) and torch.equal(rm, rm2) and torch.equal(rv, rv2) and torch.equal(sr, sr2)

    # ---- backward
    def bfin(part_b, gamma, srstd, dg, db, zero):
        dp = lambda t: t.data_ptr() if t is not None else None
        f = _lib.BnBfin(dp(part_b), C, M, dp(gamma), dp(srstd), dp(dg), dp(db), dp(zero), zero.numel())
        f._keep = (part_b, gamma, srstd, dg, db, zero)
        return f
    mkw = dict(mask_scale=a1["sc"], mask_shift=a1["sh"]) if (relu and res == 0) else (dict(relu_bits=bits1) if bits1 is not None else {})
    outs = []
    pb0 = ops.new_stat_buffer(C)                              # ONE set of sums for both forms (float atomics: a second reduction would differ in order)
    ops.bn_bwd_reduce(dyd, C, None if (mkw or not relu) else o1, C, xd, C, a1["sm"], a1["sr"], M, C, relu, pb0, **mkw)
    for fused in (False, True):
        pb = pb0.clone()
        dg, db = torch.full((C,), 0.5, device=dev), torch.full((C,), -0.25, device=dev)       # (+=: onto what is there)
        dx = torch.empty_like(xd)
        dres = torch.empty_like(xd) if res == 1 else None
        zero = torch.full((ops.stat_slots(), 2, C), 2.0, device=dev)
        if fused:
            ops.bn_bwd_apply_fin(dyd, C, None if (mkw or not relu) else o1, C, xd, C, a1["sm"], a1["sr"], bfin(pb, a1["g"], a1["sr"], dg, db, zero),
                                 M, C, relu, dx, C, dres=dres, ldres=C, **mkw)
            torch.cuda.synchronize()
            assert float(zero.abs().max()) == 0.0 and float(pb.abs().max()) > 0.0
        else:
            coef = torch.empty(3, C, device=dev)
            ops.bn_bwd_finalize(pb, M, C, a1["g"], a1["sr"], dg, db, coef)
            ops.bn_bwd_apply(dyd, C, None if (mkw or not relu) else o1, C, xd, C, a1["sm"], a1["sr"], coef, M, C, relu, dx, C, dres=dres, ldres=C, **mkw)
        torch.cuda.synchronize()
        outs.append((dx, dg, db, dres))
    for u, v in zip(outs[0], outs[1]):
        assert (u is None and v is None) or torch.equal(u, v)
    if res == 2:                                            # the join of two sites
        joins = []
        pa0, pb0 = ops.new_stat_buffer(C), ops.new_stat_buffer(C)
        ops.bn_bwd_reduce2(dyd, C, xd, C, rd, C, a1["sm"], a1["sr"], r1["sm"], r1["sr"], bits1, M, C, pa0, pb0)
        for fused in (False, True):
            pa, pb = pa0.clone(), pb0.clone()
            dga, dba, dgb, dbb = (torch.zeros(C, device=dev) for _ in range(4))
            dxa, dxb = torch.empty_like(xd), torch.empty_like(xd)
            za, zb = torch.ones(8, device=dev), torch.ones(8, device=dev)
            if fused:
                ops.bn_bwd_apply2_fin(dyd, C, xd, C, rd, C, a1["sm"], a1["sr"], r1["sm"], r1["sr"], bits1, bfin(pa, a1["g"], a1["sr"], dga, dba, za),
                                      bfin(pb, r1["g"], r1["sr"], dgb, dbb, zb), M, C, dxa, C, dxb, C)
            else:
                ca, cb = torch.empty(3, C, device=dev), torch.empty(3, C, device=dev)
                ops.bn_bwd_finalize(pa, M, C, a1["g"], a1["sr"], dga, dba, ca)
                ops.bn_bwd_finalize(pb, M, C, r1["g"], r1["sr"], dgb, dbb, cb)
                ops.bn_bwd_apply2(dyd, C, xd, C, rd, C, a1["sm"], a1["sr"], r1["sm"], r1["sr"], bits1, ca, cb, M, C, dxa, C, dxb, C)
            torch.cuda.synchronize()
            joins.append((dxa, dxb, dga, dba, dgb, dbb))
        for u, v in zip(joins[0], joins[1]):
            assert torch.equal(u, v)


def test_bn_apply_with_the_elu_in_front():
    """mde_bn_apply relu = 2: out = ELU(x) * scale + shift -- the eval form of conv -> ELU -> BatchNorm (Bts.py:216-229): the conv
    stores its pre-activation and the ELU value never exists in 16 bits.  Against torch on the same bf16 input; and the point of
    it: values saturated towards -1 keep their distance from -1 (a stored ELU output would round every one of them to -1)."""
    from mono_depth_estimation_amd import ops
    M, C = 4096, 72
    x = _bf(W.normal(61, "x", (M, C)) * 4.0)
    x[:64] = _bf(torch.linspace(-12.0, -7.0, 64 * C).view(64, C))            # ELU within 1e-3 .. 6e-6 of -1
    sc, sh = 0.5 + W.uniform(61, "sc", (C,)), W.normal(61, "sh", (C,))
    out = torch.empty(M, C, dtype=ACT, device="cuda")
    ops.bn_apply(x.to(ACT).cuda(), C, sc.cuda(), sh.cuda(), out, C, M, C, 2)
    ref = F.elu(x) * sc + sh
    _close_bf16(out.float().cpu(), ref, "bn_apply relu=2")
    with pytest.raises(RuntimeError):
        ops.bn_apply(x.to(ACT).cuda(), C, sc.cuda(), sh.cuda(), out, C, M, C, 3)
    # what it is for: a BatchNorm that amplifies (scale 1e3, shift chosen so that ELU = -1 maps to 0) resolves the saturated values
    big, off = torch.full((C,), 1e3), torch.full((C,), 1e3)
    ops.bn_apply(x.to(ACT).cuda(), C, big.cuda(), off.cuda(), out, C, M, C, 2)
    fused = out[:64].float().cpu()
    stored = (_bf(F.elu(x[:64])) * 1e3 + 1e3)                                # the same through a stored 16-bit ELU output
    exact = torch.expm1(x[:64].double()).add(1.0).mul(1e3).float()
    assert float((fused - exact).abs().max()) < 2.0 ** -7 * float(exact.abs().max()) + 1e-3
    assert float((stored - exact).abs().max()) > 20 * float((fused - exact).abs().max())


@pytest.mark.parametrize("act,scale", [("sigmoid", 10.0), ("relu", 1.0), ("elu", 0.5), (None, 2.0)])
def test_map_act_on_fp32_maps(act, scale):
    """mde_map_act_fwd / _bwd: y = scale act(p) on an fp32 map and dp = dy scale act'(.) through the kept output (BTS' get_depth ->
    Sigmoid -> x max_depth behind the one-channel head kernels)."""
    from mono_depth_estimation_amd import ops
    p = W.normal(62, "p", (2, 1, 30, 42)) * 2.0
    dy = W.normal(62, "dy", (2, 1, 30, 42))
    pr = p.clone().requires_grad_(True)
    f = {"sigmoid": torch.sigmoid, "relu": F.relu, "elu": F.elu, None: (lambda t: t)}[act]
    (scale * f(pr)).backward(dy)
    y = torch.empty_like(p).cuda()
    ops.map_act_fwd(p.cuda(), y, act, scale)
    assert torch.allclose(y.cpu(), (scale * f(p)), rtol=1e-5, atol=1e-6)
    dp = torch.empty_like(p).cuda()
    ops.map_act_bwd(dy.cuda(), y, dp, act, scale)
    assert torch.allclose(dp.cpu(), pr.grad, rtol=2e-4, atol=1e-5)
    with pytest.raises(RuntimeError):
        ops.map_act_fwd(p.cuda().flatten()[:7], y.flatten()[:7], act, scale)          # n % 4 != 0
