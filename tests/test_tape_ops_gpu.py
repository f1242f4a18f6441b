"""GPU parity of the kernels behind the tape networks (VNL / MiDaS / BTS) against plain torch fp32 ops on the same
bf16-rounded operands, all through the C ABI:
  * conv K tail (C % 64 != 0), grouped conv forward / input gradient / weight gradient (vs F.conv2d(groups=32, dilation=...)),
  * pointwise bias + activation + residual, spatial mean / broadcast, the AFA gate, bilinear resize (both align_corners
    settings, fwd and bwd), nearest x2 / 2x2 average pool, the softmax head, the NHWC -> NCHW activation head.
Tolerance for bf16 outputs: |hip - ref| <= 2^-8 |ref| + 2^-8 rms(ref) (one rounding of an fp32 result); fp32 outputs 1e-5."""
import pytest
import torch

from mono_depth_estimation_amd.ops import ACT_DTYPE as ACT        # the library's 16-bit storage type (bf16; fp16 under MDE_ACT_DTYPE=fp16)
import torch.nn.functional as F

from oracle import weights as W

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(ACT).to(torch.float32)


def _nhwc(t, pad=0):
    t = t.permute(0, 2, 3, 1).contiguous().to(ACT)
    if pad:
        t = F.pad(t, (0, pad))
    return t.cuda()


def _nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def _close(got, ref, what, tol=2.0 ** -8):
    err = (got - ref).abs()
    bound = tol * ref.abs() + tol * ref.pow(2).mean().sqrt()
    bad = (~(err <= bound)).sum().item()
    assert bad == 0, "%s: %d/%d outside tolerance, max err %.4g (ref rms %.4g)" % (
        what, bad, ref.numel(), err.max().item(), ref.pow(2).mean().sqrt().item())


def _pack_fwd(w):
    o, i, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(o, kh * kw, i).contiguous().to(ACT).cuda()


def _pack_dgrad(w):
    o, i, kh, kw = w.shape
    return w.permute(1, 2, 3, 0).reshape(i, kh * kw, o).contiguous().to(ACT).cuda()


# ---------------------------------------------------------------------------------------------- conv K tail
@pytest.mark.parametrize("Cin,Cout,k,dil", [(48, 96, 3, 1), (144, 48, 1, 1), (32, 8, 1, 1), (104, 152, 3, 2), (200, 64, 3, 1)])
def test_conv_channel_counts_that_are_not_multiples_of_64(Cin, Cout, k, dil):
    """DenseNet's 48-channel growth (Bts.py densenet161), AFA's 32-channel bottleneck (VNL.py:358), the 152-row padded
    prediction conv: forward, input gradient and weight gradient with C % 8 == 0 only."""
    from mono_depth_estimation_amd import ops
    N, H, Wd = 2, 14, 18
    p = dil * (k // 2)
    x = _bf(W.normal(3, "x", (N, Cin, H, Wd)))
    w = _bf(W.normal(3, "w", (Cout, Cin, k, k), std=(2.0 / (k * k * Cin)) ** 0.5))
    dy = _bf(W.normal(3, "dy", (N, Cout, H, Wd)))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, padding=p, dilation=dil)
    ref.backward(dy)
    ld_x = Cin + 16                                             # the neighbouring channels hold junk the kernel must not use
    xd = torch.full((N, H, Wd, ld_x), 100.0, dtype=ACT, device="cuda")
    xd[..., :Cin] = _nhwc(x)
    out = torch.zeros(N, H, Wd, Cout, dtype=ACT, device="cuda")
    d = ops.fwd_desc(N, H, Wd, ld_x, Cin, xd.numel() * 2, k, 1, p, Cout, Cout, dil=dil)
    stats = ops.new_stat_buffer(Cout)
    ops.conv_gemm(d, xd, _pack_fwd(w), out, stats)
    torch.cuda.synchronize()
    _close(_nchw(out), ref.detach(), "conv fwd C=%d" % Cin)
    assert torch.allclose(stats.sum(0)[0].cpu(), ref.detach().sum((0, 2, 3)), rtol=1e-3, atol=1e-2 * ref.detach().abs().sum((0, 2, 3)).max().item() * 1e-1 + 1e-2)
    # input gradient: contraction over Cout (also not a multiple of 64)
    ld_dy = Cout + 8
    dyd = torch.full((N, H, Wd, ld_dy), -50.0, dtype=ACT, device="cuda")
    dyd[..., :Cout] = _nhwc(dy)
    dx = torch.zeros(N, H, Wd, Cin, dtype=ACT, device="cuda")
    descs, zero = ops.dgrad_descs(N, H, Wd, Cin, Cin, H, Wd, ld_dy, Cout, dyd.numel() * 2, k, 1, p, dil=dil)
    assert not zero
    for dd in descs:
        ops.conv_gemm(dd, dyd, _pack_dgrad(w), dx)
    torch.cuda.synchronize()
    _close(_nchw(dx), xr.grad, "conv dgrad Cout=%d" % Cout)
    # weight gradient
    dw = torch.zeros(Cout, k * k, Cin, device="cuda")
    wd = ops.conv_wgrad_desc(N, H, Wd, ld_x, Cin, xd.numel() * 2, H, Wd, ld_dy, Cout, dyd.numel() * 2, k, 1, p, 2, dil=dil)
    ops.conv_wgrad(wd, dyd, xd, dw)
    torch.cuda.synchronize()
    refw = wr.grad.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin)
    assert torch.allclose(dw.cpu(), refw, rtol=2e-3, atol=2e-3 * refw.abs().max().item())


# ---------------------------------------------------------------------------------------------- grouped conv
@pytest.mark.parametrize("D,G,stride,dil,H,Wd", [(128, 4, 1, 1, 12, 20), (256, 8, 2, 1, 14, 18), (512, 16, 2, 1, 13, 17),
                                                 (1024, 32, 1, 2, 6, 10), (256, 8, 1, 1, 10, 12), (512, 64, 1, 1, 6, 6)])
def test_grouped_conv_fwd_dgrad_wgrad(D, G, stride, dil, H, Wd):
    """ResNeXtBottleneck.conv2 (VNL.py:638: 3x3, groups = 32, stride / dilation on it; 32x8d widths for MiDaS) as
    block-diagonal 64-channel GEMM tiles: forward (+ BN statistics), input gradient (strided: four output phases),
    weight gradient."""
    from mono_depth_estimation_amd import ops
    N, groups = 2, D // G
    x = _bf(W.normal(5, "x", (N, D, H, Wd)))
    w = _bf(W.normal(5, "w", (D, G, 3, 3), std=(2.0 / (9 * G)) ** 0.5))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=stride, padding=dil, dilation=dil, groups=groups)
    OH, OW = ref.shape[2:]
    dy = _bf(W.normal(5, "dy", (N, D, OH, OW)))
    ref.backward(dy)
    src = w.permute(0, 2, 3, 1).reshape(D, 9, G).contiguous().cuda()                  # master layout [O][T][G] fp32
    wf = torch.empty(D * 9 * 64, dtype=ACT, device="cuda")
    wdg = torch.empty(D * 9 * 64, dtype=ACT, device="cuda")
    ops.pack_grouped(src, wf, wdg, D, 9, G)
    xd = _nhwc(x)
    out = torch.zeros(N, OH, OW, D, dtype=ACT, device="cuda")
    d = ops.fwd_desc(N, H, Wd, D, 64, xd.numel() * 2, 3, stride, dil, D, D, dil=dil)
    d.grouped = 1
    stats = ops.new_stat_buffer(D)
    ops.conv_gemm(d, xd, wf, out, stats)
    torch.cuda.synchronize()
    _close(_nchw(out), ref.detach(), "grouped fwd")
    s2 = (ref.detach() ** 2).sum((0, 2, 3))
    assert torch.allclose(stats.sum(0)[1].cpu(), s2, rtol=2e-3, atol=1e-3 * s2.max().item())
    dyd = _nhwc(dy)
    dx = torch.full((N, H, Wd, D), 9.0, dtype=ACT, device="cuda")
    descs, zero = ops.dgrad_descs(N, H, Wd, D, D, OH, OW, D, 64, dyd.numel() * 2, 3, stride, dil, dil=dil)
    if zero:
        dx.zero_()
    for dd in descs:
        dd.grouped = 1
        ops.conv_gemm(dd, dyd, wdg, dx)
    torch.cuda.synchronize()
    _close(_nchw(dx), xr.grad, "grouped dgrad")
    dw = torch.zeros(D, 9, G, device="cuda")
    wd = ops.conv_wgrad_desc(N, H, Wd, D, D, xd.numel() * 2, OH, OW, D, D, dyd.numel() * 2, 3, stride, dil, 2, dil=dil)
    wd.group_size = G
    ops.conv_wgrad(wd, dyd, xd, dw)
    torch.cuda.synchronize()
    refw = wr.grad.permute(0, 2, 3, 1).reshape(D, 9, G)
    assert torch.allclose(dw.cpu(), refw, rtol=2e-3, atol=2e-3 * refw.abs().max().item())


# ---------------------------------------------------------------------------------------------- pointwise
@pytest.mark.parametrize("act", ["none", "relu", "elu", "sigmoid"])
@pytest.mark.parametrize("C,with_r,with_b", [(256, True, True), (48, False, True), (8, True, False), (2560, False, False)])
def test_pw_fwd_bwd(act, C, with_r, with_b):
    from mono_depth_estimation_amd import ops
    N, H, Wd = 2, 7, 9
    x = _bf(W.normal(7, "x", (N, C, H, Wd)))
    r = _bf(W.normal(7, "r", (N, C, H, Wd))) if with_r else None
    b = W.normal(7, "b", (C,), 0.5) if with_b else None
    fn = {"none": lambda t: t, "relu": F.relu, "elu": F.elu, "sigmoid": torch.sigmoid}[act]
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if with_r else None
    br = b.clone().requires_grad_(True) if with_b else None
    pre = xr + (br.view(1, C, 1, 1) if with_b else 0) + (rr if with_r else 0)
    ref = fn(pre)
    xd, rd = _nhwc(x, 8), (_nhwc(r) if with_r else None)
    out = torch.zeros(N, H, Wd, C + 16, dtype=ACT, device="cuda")
    M = N * H * Wd
    ops.pw_fwd(xd, C + 8, b.cuda() if with_b else None, rd, C, out, C + 16, M, C, act)
    torch.cuda.synchronize()
    _close(_nchw(out[..., :C]), ref.detach(), "pw fwd " + act)
    assert float(out[..., C:].float().abs().max()) == 0.0
    dy = _bf(W.normal(7, "dy", (N, C, H, Wd)))
    # the kernel differentiates through the bf16 OUTPUT it stored: build the reference the same way
    y = _nchw(out[..., :C])
    gy = {"none": torch.ones_like(y), "relu": (y > 0).float(), "elu": torch.where(y > 0, torch.ones_like(y), y + 1), "sigmoid": y * (1 - y)}[act]
    g = dy * gy
    dx = torch.full((N, H, Wd, C), 1.0, dtype=ACT, device="cuda")
    dr = torch.zeros(N, H, Wd, C, dtype=ACT, device="cuda") if with_r else None
    db = torch.zeros(C, device="cuda") if with_b else None
    ops.pw_bwd(_nhwc(dy), C, out, C + 16, dx, C, True, dr, C, False, db, M, C, act)
    torch.cuda.synchronize()
    _close(_nchw(dx), g + 1.0, "pw dx (accumulating) " + act)
    if with_r:
        _close(_nchw(dr), g, "pw dr " + act)
    if with_b:
        assert torch.allclose(db.cpu(), _bf(g).sum((0, 2, 3)) * 0 + g.sum((0, 2, 3)), rtol=1e-3, atol=1e-3 * g.abs().sum((0, 2, 3)).max().item())


def test_spatial_mean_broadcast_and_gate():
    from mono_depth_estimation_amd import ops
    N, C, H, Wd = 3, 264, 9, 11
    x = _bf(W.normal(9, "x", (N, C, H, Wd)))
    xd = _nhwc(x, 8)
    pooled = torch.zeros(N, 2 * C, dtype=ACT, device="cuda")
    ops.spatial_sum(xd, C + 8, N, H * Wd, C, 1.0 / (H * Wd), pooled[:, C:], 2 * C)
    torch.cuda.synchronize()
    _close(pooled[:, C:].float().cpu(), x.mean((2, 3)), "spatial mean")
    assert float(pooled[:, :C].float().abs().max()) == 0.0
    out = torch.full((N, H, Wd, C), 2.0, dtype=ACT, device="cuda")
    ops.spatial_bcast(pooled[:, C:], 2 * C, 0.5, out, C, N, H * Wd, C, accumulate=True)
    torch.cuda.synchronize()
    _close(_nchw(out), 2.0 + 0.5 * _bf(x.mean((2, 3))).view(N, C, 1, 1).expand(-1, -1, H, Wd), "broadcast")
    # gate
    w = _bf(torch.sigmoid(W.normal(9, "w", (N, C))))
    lat, top, dy = (_bf(W.normal(9, k, (N, C, H, Wd))) for k in ("lat", "top", "dy"))
    wd_ = w.to(ACT).cuda()
    o = torch.zeros(N, H, Wd, C, dtype=ACT, device="cuda")
    ops.gate_fwd(wd_, C, _nhwc(lat), C, _nhwc(top), C, o, C, N, H * Wd, C)
    torch.cuda.synchronize()
    _close(_nchw(o), w.view(N, C, 1, 1) * lat + top, "gate fwd")
    dlat = torch.zeros(N, H, Wd, C, dtype=ACT, device="cuda")
    dtop = torch.full((N, H, Wd, C), 1.0, dtype=ACT, device="cuda")
    dw = torch.zeros(N, C, dtype=ACT, device="cuda")
    ops.gate_bwd(_nhwc(dy), C, wd_, C, _nhwc(lat), C, dlat, C, False, dtop, C, True, dw, C, N, H * Wd, C)
    torch.cuda.synchronize()
    _close(_nchw(dlat), w.view(N, C, 1, 1) * dy, "gate dlat")
    _close(_nchw(dtop), dy + 1.0, "gate dtop")
    _close(dw.float().cpu(), (dy * lat).sum((2, 3)), "gate dw", tol=2.0 ** -7)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("H,Wd,OH,OW", [(6, 9, 12, 18), (4, 6, 8, 12), (7, 5, 13, 10), (8, 8, 8, 8), (1, 1, 4, 6), (12, 16, 6, 8)])
def test_bilinear_resize_fwd_bwd(align, H, Wd, OH, OW):
    from mono_depth_estimation_amd import ops
    N, C = 2, 24
    x = _bf(W.normal(11, "x", (N, C, H, Wd)))
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr, size=(OH, OW), mode="bilinear", align_corners=align)
    dy = _bf(W.normal(11, "dy", (N, C, OH, OW)))
    ref.backward(dy)
    out = torch.zeros(N, OH, OW, C + 8, dtype=ACT, device="cuda")
    ops.resize_bilinear_fwd(_nhwc(x), C, out, C + 8, N, H, Wd, C, OH, OW, align)
    torch.cuda.synchronize()
    _close(_nchw(out[..., :C]), ref.detach(), "resize fwd")
    dx = torch.full((N, H, Wd, C), 0.5, dtype=ACT, device="cuda")
    ops.resize_bilinear_bwd(_nhwc(dy), C, dx, C, N, H, Wd, C, OH, OW, align, accumulate=True)
    torch.cuda.synchronize()
    _close(_nchw(dx), xr.grad + 0.5, "resize bwd")


def test_nearest2_and_avgpool2():
    from mono_depth_estimation_amd import ops
    N, C, H, Wd = 2, 40, 5, 7
    x = _bf(W.normal(13, "x", (N, C, H, Wd)))
    up = torch.zeros(N, 2 * H, 2 * Wd, C, dtype=ACT, device="cuda")
    ops.nearest2_fwd(_nhwc(x), C, up, C, N, H, Wd, C)
    torch.cuda.synchronize()
    assert torch.equal(_nchw(up), F.interpolate(x, scale_factor=2, mode="nearest"))
    big = _bf(W.normal(13, "big", (N, C, 2 * H, 2 * Wd)))
    dst = torch.zeros(N, H, Wd, C, dtype=ACT, device="cuda")
    ops.sum2x2(_nhwc(big), C, dst, C, N, H, Wd, C, 0.25)
    torch.cuda.synchronize()
    _close(_nchw(dst), F.avg_pool2d(big, 2, 2), "avgpool2 fwd")
    ops.sum2x2(_nhwc(big), C, dst, C, N, H, Wd, C, 1.0)             # nearest-x2 backward
    torch.cuda.synchronize()
    _close(_nchw(dst), 4 * F.avg_pool2d(big, 2, 2), "nearest2 bwd")
    spread = torch.full((N, 2 * H, 2 * Wd, C), 1.0, dtype=ACT, device="cuda")
    ops.spread2x2(_nhwc(x), C, spread, C, N, H, Wd, C, 0.25, accumulate=True)
    torch.cuda.synchronize()
    _close(_nchw(spread), 1.0 + 0.25 * F.interpolate(x, scale_factor=2, mode="nearest"), "avgpool2 bwd")


@pytest.mark.parametrize("C,HW", [(150, 64 * 3 + 17), (7, 100), (256, 64)])
def test_softmax_head(C, HW):
    from mono_depth_estimation_amd import ops
    N = 2
    ld = (C + 7) // 8 * 8
    x = _bf(W.normal(15, "x", (N, C, HW), 2.0))
    b = W.normal(15, "b", (C,), 0.3)
    xd = torch.full((N, HW, ld), 40.0, dtype=ACT, device="cuda")
    xd[..., :C] = x.permute(0, 2, 1).to(ACT).cuda()
    logit, prob = torch.empty(N, C, HW, device="cuda"), torch.empty(N, C, HW, device="cuda")
    ops.softmax_head_fwd(xd, ld, b.cuda(), logit, prob, N, HW, C)
    torch.cuda.synchronize()
    lr = (x + b.view(1, C, 1)).requires_grad_(True)
    pr = torch.softmax(lr, 1)
    assert torch.allclose(logit.cpu(), lr.detach(), rtol=0, atol=1e-6)
    assert torch.allclose(prob.cpu(), pr.detach(), rtol=1e-5, atol=1e-7)
    dl, dp = W.normal(15, "dl", (N, C, HW)), W.normal(15, "dp", (N, C, HW))
    (lr * dl).sum().backward(retain_graph=True)
    g1 = lr.grad.clone()
    lr.grad = None
    ((pr * dp).sum() + (lr * dl).sum()).backward()
    g2 = lr.grad.clone()
    for dlog, dprob, ref in ((dl, None, g1), (dl, dp, g2)):
        dx = torch.full((N, HW, ld), 3.0, dtype=ACT, device="cuda")
        db = torch.zeros(ld, device="cuda")
        ops.softmax_head_bwd(dlog.cuda(), dprob.cuda() if dprob is not None else None, prob, dx, ld, db, N, HW, C)
        torch.cuda.synchronize()
        _close(dx[..., :C].float().cpu().permute(0, 2, 1), ref, "softmax head dx")
        assert float(dx[..., C:].float().abs().max()) == 0.0 if ld > C else True
        assert torch.allclose(db[:C].cpu(), ref.sum((0, 2)), rtol=1e-3, atol=1e-3 * ref.abs().sum((0, 2)).max().item())


@pytest.mark.parametrize("C,act,scale", [(7, "sigmoid", 1.0), (1, "sigmoid", 10.0), (20, "none", 1.0)])
def test_to_nchw_act(C, act, scale):
    from mono_depth_estimation_amd import ops
    N, HW = 2, 333
    ld = (C + 7) // 8 * 8
    x = _bf(W.normal(17, "x", (N, C, HW)))
    b = W.normal(17, "b", (C,), 0.3)
    xd = torch.full((N, HW, ld), 9.0, dtype=ACT, device="cuda")
    xd[..., :C] = x.permute(0, 2, 1).to(ACT).cuda()
    y = torch.empty(N, C, HW, device="cuda")
    ops.to_nchw_act_fwd(xd, ld, b.cuda(), y, N, HW, C, act, scale)
    torch.cuda.synchronize()
    pre = (x + b.view(1, C, 1)).requires_grad_(True)
    ref = scale * (torch.sigmoid(pre) if act == "sigmoid" else pre)
    assert torch.allclose(y.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    dy = W.normal(17, "dy", (N, C, HW))
    ref.backward(dy)
    dx = torch.full((N, HW, ld), 3.0, dtype=ACT, device="cuda")
    db = torch.zeros(ld, device="cuda")
    ops.to_nchw_act_bwd(dy.cuda(), y, dx, ld, db, N, HW, C, act, scale)
    torch.cuda.synchronize()
    _close(dx[..., :C].float().cpu().permute(0, 2, 1), pre.grad, "to_nchw dx")
    assert torch.allclose(db[:C].cpu(), pre.grad.sum((0, 2)), rtol=1e-3, atol=1e-3 * pre.grad.abs().sum((0, 2)).max().item())


@pytest.mark.parametrize("O,k,s,p,H,Wd", [(96, 7, 2, 3, 48, 64), (64, 3, 2, 1, 33, 47), (64, 9, 2, 0, 40, 56)])
def test_image_conv_over_the_16_slot_hi_lo_operands(O, k, s, p, H, Wd):
    """The eval-mode image convolution of graph.ImageStem (densenet161's conv0, DORN's conv1, Eigen's 9x9): image slots
    [xh | xl | xh], weight slots [wh | wh | wl] (mde_nchw_to_nhwc_split16 / mde_stem_weight_split16), the k x k taps in
    launches of at most 32 (the later ones accumulating).  Against torch's conv of the FP32 image with the FP32 weights: only
    the output's own storage rounding is left (and, over several launches, that of the partial sums) -- the one-term path on a
    16-bit image differs from that reference by the operands' rounding on top."""
    from mono_depth_estimation_amd import ops
    N, C, Cp = 2, 3, 8
    x = W.uniform(9, "img", (N, C, H, Wd))                                        # generic fp32 pixels, as a loader hands them over
    w = W.normal(9, "w", (O, C, k, k), std=(2.0 / (k * k * C)) ** 0.5)
    ref = F.conv2d(x, w, stride=s, padding=p)
    OH, OW = ref.shape[2:]
    master = torch.zeros(O, k * k, Cp)
    master[:, :, :C] = w.permute(0, 2, 3, 1).reshape(O, k * k, C)
    master = master.cuda()
    x16 = torch.full((N, H, Wd, 16), 7.0, dtype=ACT, device="cuda")
    w16 = torch.full((O * k * k * 16,), 7.0, dtype=ACT, device="cuda")
    ops.nchw_to_nhwc_split16(x.cuda(), x16)
    ops.stem_weight_split16(master, w16, O * k * k, Cp, C)
    xs = x16.float().cpu()
    assert torch.equal(xs[..., 0:3], xs[..., 6:9]) and float(xs[..., 9:].abs().max()) == 0.0
    back = (xs[..., 0:3] + xs[..., 3:6]).permute(0, 3, 1, 2)
    assert float((back - x).abs().max()) < (2.0 ** -15 if ACT == torch.bfloat16 else 2.0 ** -12)
    taps = [(i - p, j - p, i * k + j) for i in range(k) for j in range(k)]
    out = torch.full((N, OH, OW, O), 7.0, dtype=ACT, device="cuda")
    nl = 0
    for t0 in range(0, k * k, 32):
        d = ops.conv_desc(N, H, Wd, 16, 16, x16.numel() * 2, OH, OW, s, s, taps[t0:t0 + 32], k * k, OH, OW, O, ncols=O, accumulate=t0 > 0)
        ops.conv_gemm(d, x16, w16, out)
        nl += 1
    torch.cuda.synchronize()
    _close(_nchw(out), ref, "image conv over hi / lo slots", tol=2.0 ** -8 if nl == 1 else 2.0 ** -6)
    e = float((_nchw(out) - ref).mean((0, 2, 3)).abs().mean() / ref.abs().mean())
    print("mean |per-channel mean error| / mean |value| against the fp32 conv: %.2e" % e)
    assert e < 2e-4
