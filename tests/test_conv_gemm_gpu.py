"""GPU parity of the implicit-GEMM conv kernel (mde_conv_gemm) against plain torch fp32
ops on the same bf16-rounded operands.  Tolerance: the kernel accumulates in fp32 and
rounds once to bf16, so |hip - ref| <= 2^-8 * |ref| + 2^-8 * rms(ref) (stated here)."""
import pytest
import torch

from mono_depth_estimation_amd.ops import ACT_DTYPE as ACT        # the library's 16-bit storage type (bf16; fp16 under MDE_ACT_DTYPE=fp16)
import torch.nn.functional as F

from oracle import weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "plain", "halo128", "halo256", "shallow"])
def conv_form(request):
    """Every test of this file runs five times: with the library's own choice between the plain tiles and the halo-tiled
    form (a 2-D pixel block whose input window is staged once per 64-channel chunk and shared by the taps), with the halo
    form switched off, with each of its two kernels forced wherever the geometry is eligible, and with neither the halo form
    nor the deep staging ring that grids of at most one tile per CU -- most shapes here -- otherwise get (MDE_CONV_HALO and
    MDE_CONV_DEEP are read on every call)."""
    import os
    old = os.environ.get("MDE_CONV_HALO")
    old_d = os.environ.get("MDE_CONV_DEEP")
    if request.param == "shallow":
        os.environ["MDE_CONV_DEEP"] = "0"
    val = {"auto": None, "plain": "0", "halo128": "1", "halo256": "2", "shallow": "0"}[request.param]
    if val is None:
        os.environ.pop("MDE_CONV_HALO", None)
    else:
        os.environ["MDE_CONV_HALO"] = val
    old_w = os.environ.get("MDE_WGRAD_WIN")
    if request.param == "halo256":            # ... and the weight gradients of these tests through the windowed kernel
        os.environ["MDE_WGRAD_WIN"] = "1"
    yield request.param
    if old_w is None:
        os.environ.pop("MDE_WGRAD_WIN", None)
    else:
        os.environ["MDE_WGRAD_WIN"] = old_w
    if old is None:
        os.environ.pop("MDE_CONV_HALO", None)
    else:
        os.environ["MDE_CONV_HALO"] = old
    if old_d is None:
        os.environ.pop("MDE_CONV_DEEP", None)
    else:
        os.environ["MDE_CONV_DEEP"] = old_d


def _bf(t):
    return t.to(ACT).to(torch.float32)


def _nhwc(t):  # NCHW fp32 cpu -> NHWC bf16 cuda
    return t.permute(0, 2, 3, 1).contiguous().to(ACT).cuda()


def _nchw(t):  # NHWC bf16 cuda -> NCHW fp32 cpu
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def _pack_fwd(w):  # OIHW fp32 -> [O][kh*kw][I] bf16 cuda
    o, i, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(o, kh * kw, i).contiguous().to(ACT).cuda()


def _pack_dgrad(w):  # OIHW -> [I][kh*kw][O]
    o, i, kh, kw = w.shape
    return w.permute(1, 2, 3, 0).reshape(i, kh * kw, o).contiguous().to(ACT).cuda()


def _assert_close(got, ref, what, tol=2.0 ** -8):
    err = (got - ref).abs()
    bound = tol * ref.abs() + tol * ref.pow(2).mean().sqrt()
    bad = (~(err <= bound)).sum().item()          # NaN/Inf compare False: they count as bad
    assert bad == 0, "%s: %d/%d outside tolerance, max err %.4g (ref rms %.4g)" % (
        what, bad, ref.numel(), err.max().item(), ref.pow(2).mean().sqrt().item())


@pytest.mark.parametrize("N,H,Wd,Cin,Cout,k,s,p", [
    (2, 12, 20, 64, 128, 3, 1, 1),     # one full column tile
    (2, 12, 20, 128, 64, 1, 1, 0),     # 64-column variant, ragged pixel tile (480 rows)
    (1, 13, 17, 64, 192, 3, 2, 1),     # stride 2, odd sizes, ragged column tile
    (2, 9, 11, 192, 320, 1, 2, 0),     # 1x1 stride 2 (downsample)
    (1, 10, 10, 64, 72, 5, 1, 2),      # 25 taps, columns not a multiple of the tile
    (2, 160, 241, 64, 256, 1, 1, 0),   # short K (1 step), 1206 tiles of 128x128: the single-buffer tile, four workgroups per CU
    (2, 160, 241, 200, 136, 1, 1, 0),  # the same with a K tail (4 steps, the last one 8 channels) and a ragged column tile
    (2, 160, 241, 64, 256, 3, 1, 1),   # 302 tiles of 256x256 > 256 CUs: full rounds + 128x128 tail launch
    (2, 160, 121, 64, 256, 3, 1, 1),   # 152 tiles of 256x256 (59% fill): the cost model takes 192x256 (202 tiles, ragged)
])
def test_conv_forward(N, H, Wd, Cin, Cout, k, s, p):
    from mono_depth_estimation_amd import ops
    x = _bf(W.normal(1, "x", (N, Cin, H, Wd)))
    w = _bf(W.normal(1, "w", (Cout, Cin, k, k), std=(2.0 / (k * k * Cin)) ** 0.5))
    ref = F.conv2d(x, w, stride=s, padding=p)
    xd, wd = _nhwc(x), _pack_fwd(w)
    OH, OW = ref.shape[2:]
    ld_out = Cout + 8                                    # channel-sliced output view
    out = torch.full((N, OH, OW, ld_out), 7.0, dtype=ACT, device="cuda")
    d = ops.fwd_desc(N, H, Wd, Cin, Cin, xd.numel() * 2, k, s, p, Cout, ld_out)
    stats = ops.new_stat_buffer(Cout)
    ops.conv_gemm(d, xd, wd, out, stats)
    torch.cuda.synchronize()
    _assert_close(_nchw(out[..., :Cout]), ref, "conv fwd")
    assert (out[..., Cout:].float() == 7.0).all(), "wrote outside its channel slice"
    st = stats.sum(0).cpu()
    ref_s1, ref_s2 = ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))
    assert torch.allclose(st[0], ref_s1, rtol=1e-3, atol=1e-2 * ref_s2.max().sqrt().item())
    assert torch.allclose(st[1], ref_s2, rtol=1e-3, atol=1e-3 * ref_s2.max().item())


@pytest.mark.parametrize("N,H,Wd,Cin,Cout,k,s,p", [
    (2, 12, 20, 64, 128, 3, 1, 1),
    (1, 14, 18, 128, 64, 3, 2, 1),     # strided: four output phases
    (1, 13, 17, 128, 64, 3, 2, 1),     # odd input size
    (2, 10, 12, 64, 128, 1, 2, 0),     # 1x1 stride 2: three empty phases -> zero fill
    (2, 8, 8, 192, 64, 1, 1, 0),
    (2, 160, 241, 256, 64, 1, 1, 0),   # short K: the single-buffer tile, also with accumulate
    (2, 160, 241, 256, 64, 3, 1, 1),   # split launch (256x256 rounds + 128x128 tail), also with accumulate
])
def test_conv_dgrad(N, H, Wd, Cin, Cout, k, s, p):
    from mono_depth_estimation_amd import ops
    w = _bf(W.normal(2, "w", (Cout, Cin, k, k), std=(2.0 / (k * k * Cout)) ** 0.5))
    OH, OW = ops.out_size(H, k, s, p), ops.out_size(Wd, k, s, p)
    dy = _bf(W.normal(2, "dy", (N, Cout, OH, OW)))
    x = torch.zeros(N, Cin, H, Wd, requires_grad=True)
    F.conv2d(x, w, stride=s, padding=p).backward(dy)
    ref = x.grad
    dyd, wd = _nhwc(dy), _pack_dgrad(w)
    dx = torch.full((N, H, Wd, Cin), 3.0, dtype=ACT, device="cuda")
    descs, zero = ops.dgrad_descs(N, H, Wd, Cin, Cin, OH, OW, Cout, Cout, dyd.numel() * 2, k, s, p)
    if zero:
        dx.zero_()
    for d in descs:
        ops.conv_gemm(d, dyd, wd, dx)
    torch.cuda.synchronize()
    _assert_close(_nchw(dx), ref, "conv dgrad")
    # accumulate: a second pass adds onto the first
    for d in descs:
        d.accumulate = 1
        ops.conv_gemm(d, dyd, wd, dx)
    torch.cuda.synchronize()
    _assert_close(_nchw(dx), 2 * ref, "conv dgrad accumulate", tol=2.0 ** -7)


def _unpool(x):
    n, c, h, w = x.shape
    u = x.new_zeros(n, c, 2 * h, 2 * w)
    u[:, :, ::2, ::2] = x
    return u


@pytest.mark.parametrize("N,h,w,Cin", [(2, 6, 8, 64), (1, 5, 7, 128)])
def test_upproj_phases_fwd_and_dgrad(N, h, w, Cin):
    """Four-phase 5x5 over x == 5x5/pad2 over the zero-stuffed map (FCRN.py:31-44,180,187),
    both 5x5 branches fused along the output channels."""
    from mono_depth_estimation_amd import ops
    Cout = Cin // 2
    x = _bf(W.normal(3, "x", (N, Cin, h, w)))
    wu = _bf(W.normal(3, "wu", (Cout, Cin, 5, 5), std=(2.0 / (25 * Cout)) ** 0.5))
    wb = _bf(W.normal(3, "wb", (Cout, Cin, 5, 5), std=(2.0 / (25 * Cout)) ** 0.5))
    wcat = torch.cat([wu, wb], 0)
    xi = x.clone().requires_grad_(True)
    ref = F.conv2d(_unpool(xi), wcat, padding=2)
    xd = _nhwc(x)
    out = torch.empty(N, 2 * h, 2 * w, 2 * Cout, dtype=ACT, device="cuda")
    for d in ops.upproj_fwd_descs(N, h, w, Cin, Cin, xd.numel() * 2, 2 * Cout, 2 * Cout):
        ops.conv_gemm(d, xd, _pack_fwd(wcat), out)
    torch.cuda.synchronize()
    _assert_close(_nchw(out), ref.detach(), "upproj fwd")
    dy = _bf(W.normal(3, "dy", tuple(ref.shape)))
    ref.backward(dy)
    dyd = _nhwc(dy)
    dx = torch.empty(N, h, w, Cin, dtype=ACT, device="cuda")
    d = ops.upproj_dgrad_desc(N, h, w, Cin, Cin, 2 * Cout, 2 * Cout, dyd.numel() * 2)
    ops.conv_gemm(d, dyd, _pack_dgrad(wcat), dx)
    torch.cuda.synchronize()
    _assert_close(_nchw(dx), xi.grad, "upproj dgrad")


def test_conv_rejects_bad_args():
    from mono_depth_estimation_amd import _lib, ops
    x = torch.zeros(1, 4, 4, 48, dtype=ACT, device="cuda")
    w = torch.zeros(64, 1, 48, dtype=ACT, device="cuda")
    out = torch.zeros(1, 4, 4, 64, dtype=ACT, device="cuda")
    d = ops.fwd_desc(1, 4, 4, 48, 44, x.numel() * 2, 1, 1, 0, 64, 64)
    with pytest.raises(_lib.MdeError, match="multiple of 8"):
        ops.conv_gemm(d, x, w, out)


@pytest.mark.parametrize("N,H,Wd,Cin,Cout,dil", [
    (2, 30, 40, 256, 128, 3), (1, 30, 40, 256, 128, 6), (1, 30, 40, 256, 128, 12), (1, 30, 40, 256, 128, 18),
    (1, 30, 40, 256, 128, 24),                # the reference's atrous stack (Bts.py:62-63,196-205: 256 -> 128 at 1/8 scale)
    (2, 17, 23, 128, 128, 2),                 # VNL's FTB blocks (VNL.py:336-339)
])
def test_atrous_conv_fwd_dgrad_wgrad(N, H, Wd, Cin, Cout, dil):
    """Dilated 3x3 convolution (padding = dilation) is only a different tap table for the same kernels: forward,
    input gradient and weight gradient against torch's dilated conv2d."""
    from mono_depth_estimation_amd import ops
    x = _bf(W.normal(6, "x", (N, Cin, H, Wd)))
    w = _bf(W.normal(6, "w", (Cout, Cin, 3, 3), std=(2.0 / (9 * Cin)) ** 0.5))
    xi, wi = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xi, wi, padding=dil, dilation=dil)
    dy = _bf(W.normal(6, "dy", tuple(ref.shape)))
    ref.backward(dy)
    xd, dyd = _nhwc(x), _nhwc(dy)
    out = torch.empty(N, H, Wd, Cout, dtype=ACT, device="cuda")
    ops.conv_gemm(ops.fwd_desc(N, H, Wd, Cin, Cin, xd.numel() * 2, 3, 1, dil, Cout, Cout, dil=dil), xd, _pack_fwd(w), out)
    dx = torch.empty(N, H, Wd, Cin, dtype=ACT, device="cuda")
    descs, zero = ops.dgrad_descs(N, H, Wd, Cin, Cin, H, Wd, Cout, Cout, dyd.numel() * 2, 3, 1, dil, dil=dil)
    assert len(descs) == 1 and not zero
    ops.conv_gemm(descs[0], dyd, _pack_dgrad(w), dx)
    dw = torch.zeros(Cout, 9, Cin, device="cuda")
    ops.conv_wgrad(ops.conv_wgrad_desc(N, H, Wd, Cin, Cin, xd.numel() * 2, H, Wd, Cout, Cout, dyd.numel() * 2, 3, 1, dil, 2,
                                       dil=dil), dyd, xd, dw)
    torch.cuda.synchronize()
    _assert_close(_nchw(out), ref.detach(), "atrous fwd d=%d" % dil)
    _assert_close(_nchw(dx), xi.grad, "atrous dgrad d=%d" % dil)
    _assert_close(dw.cpu(), wi.grad.permute(0, 2, 3, 1).reshape(Cout, 9, Cin), "atrous wgrad d=%d" % dil)


def test_random_shapes_fwd_dgrad_wgrad():
    """Seeded sweep over convolution geometries (kernel 1 / 3 / 5, stride 1 / 2, dilation 1 / 2, channel counts that are
    multiples of 8 but not of 64, pixel counts on both sides of the tile-selection thresholds: single-buffer 128x128 and
    128x64 tiles, the 2-deep ring, the 256-column tiles, mixed row tiles of the weight gradient): forward + BatchNorm
    statistics, input gradient and weight gradient against torch on the same bf16-rounded operands."""
    from mono_depth_estimation_amd import ops
    rng = __import__("numpy").random.RandomState(2024)
    chans = [8, 24, 48, 64, 72, 96, 128, 152, 192, 200, 256, 320]
    for trial in range(24):
        k = int(rng.choice([1, 1, 3, 3, 5]))
        s = int(rng.choice([1, 1, 2]))
        dil = int(rng.choice([1, 1, 2])) if k == 3 and s == 1 else 1
        p = dil * (k // 2)
        Cin, Cout = int(rng.choice(chans)), int(rng.choice(chans))
        N = int(rng.choice([1, 2, 3]))
        big = trial % 3 == 0                                     # every third case has > 512 tiles of 128 pixels
        H, Wd = (int(rng.randint(150, 260)), int(rng.randint(150, 300))) if big else (int(rng.randint(7, 60)), int(rng.randint(7, 60)))
        if big and k == 5:
            k, p = 3, dil
        x = _bf(torch.randn(N, Cin, H, Wd, generator=torch.Generator().manual_seed(trial)))
        w = _bf(torch.randn(Cout, Cin, k, k, generator=torch.Generator().manual_seed(100 + trial)) * (2.0 / (k * k * Cin)) ** 0.5)
        xi, wi = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        ref = F.conv2d(xi, wi, stride=s, padding=p, dilation=dil)
        OH, OW = ref.shape[2:]
        dy = _bf(torch.randn(tuple(ref.shape), generator=torch.Generator().manual_seed(200 + trial)))
        ref.backward(dy)
        xd, dyd = _nhwc(x), _nhwc(dy)
        tag = "trial %d: N%d %dx%d C%d->%d k%d s%d d%d" % (trial, N, H, Wd, Cin, Cout, k, s, dil)
        out = torch.empty(N, OH, OW, Cout, dtype=ACT, device="cuda")
        stats = ops.new_stat_buffer(Cout)
        ops.conv_gemm(ops.fwd_desc(N, H, Wd, Cin, Cin, xd.numel() * 2, k, s, p, Cout, Cout, dil=dil), xd, _pack_fwd(w), out, stats)
        _assert_close(_nchw(out), ref.detach(), tag + " fwd")
        st = stats.sum(0).cpu()
        r1, r2 = ref.detach().sum((0, 2, 3)), (ref.detach() ** 2).sum((0, 2, 3))
        assert torch.allclose(st[0], r1, rtol=2e-3, atol=2e-2 * r2.max().sqrt().item()), tag + " stats"
        assert torch.allclose(st[1], r2, rtol=2e-3, atol=2e-3 * r2.max().item()), tag + " stats"
        dx = torch.full((N, H, Wd, Cin), 3.0, dtype=ACT, device="cuda")
        descs, zero = ops.dgrad_descs(N, H, Wd, Cin, Cin, OH, OW, Cout, Cout, dyd.numel() * 2, k, s, p, dil=dil)
        if zero:
            dx.zero_()
        for d in descs:
            ops.conv_gemm(d, dyd, _pack_dgrad(w), dx)
        _assert_close(_nchw(dx), xi.grad, tag + " dgrad")
        dw = torch.zeros(Cout, k * k, Cin, device="cuda")
        ks = int(rng.choice([1, 2, 5]))
        ops.conv_wgrad(ops.conv_wgrad_desc(N, H, Wd, Cin, Cin, xd.numel() * 2, OH, OW, Cout, Cout, dyd.numel() * 2, k, s, p, ks, dil=dil),
                       dyd, xd, dw)
        torch.cuda.synchronize()
        _assert_close(dw.cpu(), wi.grad.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin), tag + " wgrad", tol=2.0 ** -7)


@pytest.mark.parametrize("N,H,Wd,Cin,Cout,k,act,with_bias,with_res", [
    (2, 12, 20, 64, 128, 3, "relu", True, False),      # conv bias + ReLU (MiDaS ResidualConvUnit.conv1, DORN, MyNet)
    (2, 12, 20, 64, 128, 3, None, True, True),         # conv bias + residual sum (ResidualConvUnit.conv2)
    (1, 30, 40, 128, 64, 3, "elu", False, False),      # BTS conv + ELU, 64-column tile
    (2, 9, 11, 192, 72, 1, "sigmoid", True, True),     # 1x1, ragged column tile: the scalar tail of the store loop
    (2, 160, 241, 64, 256, 3, "relu", True, True),     # large grid: the 256-pixel halo form takes it when the library chooses
])
def test_fused_epilogue_equals_conv_plus_pointwise_pass(N, H, Wd, Cin, Cout, k, act, with_bias, with_res):
    """mde_conv_gemm_act: the bias joins the fp32 accumulator before the one rounding to the storage type; without a residual
    the activation does too (out = 16-bit(act(conv + bias))), with one: out = 16-bit(act(16-bit(conv + bias) + residual)).
    Checked against torch evaluating exactly that on the same operands in fp32 -- in every form of the conv loop -- and against
    the separate pass (conv launch + mde_pw_fwd), which rounds the bare conv result first: the two may differ by that one
    rounding (an ulp of the conv value), no more."""
    from mono_depth_estimation_amd import ops
    p = k // 2
    x = _bf(W.normal(7, "x", (N, Cin, H, Wd)))
    w = _bf(W.normal(7, "w", (Cout, Cin, k, k), std=(2.0 / (k * k * Cin)) ** 0.5))
    xd, wd = _nhwc(x), _pack_fwd(w)
    bias = W.normal(7, "b", (Cout,), 0.5).cuda() if with_bias else None
    res = _nhwc(_bf(W.normal(7, "r", (N, Cout, H, Wd)))) if with_res else None
    d = ops.fwd_desc(N, H, Wd, Cin, Cin, xd.numel() * 2, k, 1, p, Cout, Cout)
    plain = torch.empty(N, H, Wd, Cout, dtype=ACT, device="cuda")
    ops.conv_gemm(d, xd, wd, plain)
    want = torch.empty_like(plain)
    ops.pw_fwd(plain, Cout, bias, res, Cout if with_res else 0, want, Cout, N * H * Wd, Cout, act)
    got = torch.full_like(plain, 7.0)
    ops.conv_gemm(d, xd, wd, got, bias=bias, res=res, act=act)
    torch.cuda.synchronize()
    f = {"relu": F.relu, "elu": F.elu, "sigmoid": torch.sigmoid, None: (lambda t: t)}[act]
    conv = F.conv2d(x, w, padding=p)
    pre = conv + (bias.cpu().view(1, -1, 1, 1) if with_bias else 0.0)
    ref = f(_bf(pre) + _nchw(res)) if with_res else f(pre)
    g = _nchw(got)
    # one storage rounding of the result (half an ulp: <= 2^-8 |ref|), + with a residual the rounding of (conv + bias) before
    # the sum, at ITS magnitude (the residual may cancel it) -- and torch sums in another order, so a few (conv + bias) values
    # in 20 M land on the other side of a rounding boundary: a whole ulp, <= 2^-7 |conv + bias|
    tol = 2.0 ** -7 * ref.abs() + 2.0 ** -7 * (pre.abs() if with_res else 0.0) + 1e-6 + 2.0 ** -8 * 1e-2
    bad = int((~((g - ref).abs() <= tol)).sum())
    assert bad == 0, "%d/%d outside the rounding bound, max err %.4g" % (bad, ref.numel(), float((g - ref).abs().max()))
    # the separate pass rounds the bare conv result before the bias: at most that rounding apart (through the activation: slope <= 1)
    sep = _nchw(want)
    tol2 = 2.0 ** -7 * (conv.abs() + pre.abs() + ref.abs()) + 1e-6
    assert int((~((g - sep).abs() <= tol2)).sum()) == 0, float((g - sep).abs().max())
    # no systematic shift per channel against fp32 (the double rounding across the constant bias had one)
    if with_bias and not with_res and act != "sigmoid":
        m_f = float(((g - ref).mean((0, 2, 3)).abs() / ref.abs().mean((0, 2, 3))).mean())
        m_s = float(((sep - ref).mean((0, 2, 3)).abs() / ref.abs().mean((0, 2, 3))).mean())
        print("mean |per-channel mean error| / mean |value|: fused %.2e, separate pass %.2e" % (m_f, m_s))


def test_fused_epilogue_rejects_an_accumulating_launch():
    from mono_depth_estimation_amd import _lib, ops
    x = torch.zeros(1, 4, 4, 64, dtype=ACT, device="cuda")
    w = torch.zeros(64, 1, 64, dtype=ACT, device="cuda")
    out = torch.zeros(1, 4, 4, 64, dtype=ACT, device="cuda")
    d = ops.fwd_desc(1, 4, 4, 64, 64, x.numel() * 2, 1, 1, 0, 64, 64)
    d.accumulate = 1
    with pytest.raises(_lib.MdeError, match="accumulating"):
        ops.conv_gemm(d, x, w, out, act="relu")


@pytest.mark.parametrize("N,H,Wd,Cin,Cout,k,s,mode,acc,xmul", [
    (2, 24, 40, 256, 64, 1, 1, "scale", False, 1),     # conv3's input gradient -> bn2's sums (64-column tile)
    (2, 24, 40, 128, 128, 3, 1, "scale", False, 1),    # conv2's -> bn1's
    (1, 26, 30, 128, 128, 3, 2, "scale", False, 1),    # strided: four output phases, each adds its pixels
    (2, 24, 40, 64, 256, 1, 1, "bits", True, 1),       # the next block's conv1, accumulating onto the shortcut's gradient -> bn3's
    (2, 24, 40, 64, 256, 1, 1, "none", False, 1),      # no ReLU (conv2 / bn2 under the decoder)
    (2, 32, 48, 64, 64, 3, 1, "scale", False, 2),      # the up-projection's 3x3: the site's input is the upper half of a 2C tensor
    (2, 160, 241, 256, 64, 3, 1, "scale", True, 1),    # split launch (256x256 rounds + 128x128 tail), accumulating
    (2, 24, 40, 64, 256, 1, 1, "join", True, 1),       # a block with a projection shortcut: bn3's and the shortcut BN's sums
    (2, 12, 20, 128, 64, 3, 1, "join", False, 2),      # the up-projection's join: the second site's input is the lower half of 2C
    (2, 160, 241, 128, 128, 3, 1, "join", True, 1),    # 8-wave tiles
])
def test_dgrad_with_fused_batchnorm_backward_sums(N, H, Wd, Cin, Cout, k, s, mode, acc, xmul):
    """mde_conv_gemm_bnred: the input-gradient launch also adds the BatchNorm-backward sums of the site whose output gradient it
    writes.  (a) the gradient equals the plain launch's bit for bit; (b) the sums equal mde_bn_bwd_reduce's over that
    gradient and the same site input -- the same arithmetic per element, summed in another order -- and a float64 torch
    evaluation of sum(g') and sum(g' * xhat)."""
    from mono_depth_estimation_amd import ops
    p = k // 2
    w = _bf(W.normal(5, "w", (Cout, Cin, k, k), std=(2.0 / (k * k * Cout)) ** 0.5))
    OH, OW = ops.out_size(H, k, s, p), ops.out_size(Wd, k, s, p)
    dyd, wd = _nhwc(_bf(W.normal(5, "dy", (N, Cout, OH, OW)))), _pack_dgrad(w)
    M = N * H * Wd
    xfull = W.normal(5, "x", (N, H, Wd, Cin * xmul), std=1.5).add_(0.3).to(ACT).cuda()     # the site's input (pre-BN)
    xs = xfull[..., :Cin]
    mean, rstd = W.normal(5, "mu", (Cin,), std=0.5).cuda(), W.uniform(5, "rs", (Cin,), 0.5, 2.0).cuda()
    gamma, beta = W.normal(5, "g", (Cin,)).cuda(), W.normal(5, "b", (Cin,), std=0.3).cuda()
    scale = gamma * rstd
    shift = beta - mean * scale
    bits = None
    second = None
    if mode == "join":
        x2full = W.normal(5, "x2", (N, H, Wd, Cin * xmul), std=0.7).add_(-0.2).to(ACT).cuda()
        x2 = x2full[..., Cin * (xmul - 1):]
        mean2, rstd2 = W.normal(5, "mu2", (Cin,), std=0.5).cuda(), W.uniform(5, "rs2", (Cin,), 0.5, 2.0).cuda()
        part_b = ops.new_stat_buffer(Cin)
        second = (x2, mean2, rstd2, part_b, Cin * xmul)
    if mode in ("bits", "join"):
        mask = torch.from_numpy((W.uniform(5, "m", (N, H, Wd, Cin)) > 0.4).numpy()).cuda()
        weights = (2 ** torch.arange(8, device="cuda")).view(1, 1, 1, 1, 8)
        bits = (mask.view(N, H, Wd, Cin // 8, 8).long() * weights).sum(-1).to(torch.uint8).contiguous()
    elif mode == "scale":
        mask = (xs.float() * scale + shift) > 0
    else:
        mask = torch.ones(N, H, Wd, Cin, dtype=torch.bool, device="cuda")
    base = _nhwc(_bf(W.normal(5, "skip", (N, Cin, H, Wd)))) if acc else None
    descs, zero = ops.dgrad_descs(N, H, Wd, Cin, Cin, OH, OW, Cout, Cout, dyd.numel() * 2, k, s, p)

    def run(red):
        dx = base.clone() if acc else torch.full((N, H, Wd, Cin), 3.0, dtype=ACT, device="cuda")
        if zero and not acc:
            dx.zero_()
        for d in descs:
            d.accumulate = int(acc)
            ops.conv_gemm(d, dyd, wd, dx, red=red)
        return dx

    part = ops.new_stat_buffer(Cin)
    red = ops.bn_red(xs, mean, rstd, part, scale if mode == "scale" else None, shift if mode == "scale" else None, bits,
                     x_ld=Cin * xmul, second=second)
    plain, fused = run(None), run(red)
    torch.cuda.synchronize()
    assert torch.equal(plain, fused), "the gradient itself must not change"
    sums = part.double().sum(0)
    part2 = ops.new_stat_buffer(Cin)
    if mode == "join":
        part2b = ops.new_stat_buffer(Cin)
        ops.bn_bwd_reduce2(fused, Cin, xs, Cin * xmul, x2, Cin * xmul, mean, rstd, mean2, rstd2, bits, M, Cin, part2, part2b)
    else:
        ops.bn_bwd_reduce(fused, Cin, None, 0, xs, Cin * xmul, mean, rstd, M, Cin, mode != "none", part2,
                          scale if mode == "scale" else None, shift if mode == "scale" else None, bits)
    torch.cuda.synchronize()
    sums2 = part2.double().sum(0)
    g = torch.where(mask, fused.double(), torch.zeros((), dtype=torch.float64, device="cuda"))

    def check(sums, sums2, x, mean, rstd):
        xhat = ((x.float() - mean) * rstd).double()
        ref = torch.stack([g.sum((0, 1, 2)), (g * xhat).sum((0, 1, 2))])
        mag = torch.stack([g.abs().sum((0, 1, 2)), (g * xhat).abs().sum((0, 1, 2))]) + 1e-30
        e_ref, e_red = float(((sums - ref).abs() / mag).max()), float(((sums - sums2).abs() / mag).max())
        assert e_ref < 2e-6 and e_red < 2e-6, (e_ref, e_red)

    check(sums, sums2, xs, mean, rstd)
    if mode == "join":
        check(part_b.double().sum(0), part2b.double().sum(0), x2, mean2, rstd2)
    if acc and mode in ("bits", "join") and xmul == 1 and len(descs) == 1:
        # mde_bn_red.add: the launch brings the identity shortcut's gradient in itself -- (block-output gradient under ITS mask
        # bits) + result, instead of accumulating onto a copy of it that another pass wrote.  Same bits, same sums.
        src = _nhwc(_bf(W.normal(5, "dout", (N, Cin, H, Wd))))
        m2 = torch.from_numpy((W.uniform(5, "m2", (N, H, Wd, Cin)) > 0.3).numpy()).cuda()
        bits2 = (m2.view(N, H, Wd, Cin // 8, 8).long() * (2 ** torch.arange(8, device="cuda")).view(1, 1, 1, 1, 8)).sum(-1).to(torch.uint8).contiguous()
        masked = torch.where(m2, src, torch.zeros((), dtype=ACT, device="cuda"))
        ref_out = masked.clone()
        descs[0].accumulate = 1
        ops.conv_gemm(descs[0], dyd, wd, ref_out)
        part.zero_()
        if second is not None:
            second[3].zero_()
        out = torch.full((N, H, Wd, Cin), 3.0, dtype=ACT, device="cuda")
        descs[0].accumulate = 0
        ops.conv_gemm(descs[0], dyd, wd, out, red=ops.bn_red_with_add(red, src, bits2))
        torch.cuda.synchronize()
        assert torch.equal(out, ref_out)
        part3 = ops.new_stat_buffer(Cin)
        if mode == "join":
            part3b = ops.new_stat_buffer(Cin)
            ops.bn_bwd_reduce2(out, Cin, xs, Cin, x2, Cin, mean, rstd, mean2, rstd2, bits, M, Cin, part3, part3b)
        else:
            ops.bn_bwd_reduce(out, Cin, None, 0, xs, Cin, mean, rstd, M, Cin, True, part3, None, None, bits)
        torch.cuda.synchronize()
        a3, b3 = part.double().sum(0), part3.double().sum(0)
        assert float(((a3 - b3).abs() / (b3.abs() + 1e-3 * b3.abs().amax(dim=1, keepdim=True))).max()) < 2e-3
    assert float(mask.float().mean()) < 0.95 or mode == "none"


@pytest.mark.parametrize("N,H,Wd,Cin,Cout,k,s,p,dil,transposed", [
    (2, 12, 20, 64, 128, 3, 1, 1, 1, False),
    (2, 12, 20, 128, 64, 1, 1, 0, 1, False),     # 1 x 1: two taps at the same offset
    (1, 13, 17, 64, 192, 3, 2, 1, 1, False),
    (2, 40, 48, 64, 136, 3, 1, 2, 2, False),     # dilated (the halo form's parity classes), ragged column tile
    (1, 10, 10, 64, 72, 5, 1, 2, 1, False),      # 25 taps -> 50: two launches, the second accumulating
    (2, 12, 20, 72, 128, 3, 1, 1, 1, True),      # the transposed packing (a ConvTranspose2d's forward operand), K tail
])
def test_conv_forward_two_term_weight_shadow(N, H, Wd, Cin, Cout, k, s, p, dil, transposed):
    """The eval-mode path: mde_pack_split_batch writes hi = 16-bit(w), lo = 16-bit(w - hi) as the tap-doubled operand and the
    tap-doubled launch (ops.split_descs) contracts with both.  hi + lo holds w to 2^-16 relative (bf16; fp16's second term is
    limited by its subnormals), and the launch equals a torch conv with the weights hi + lo to the kernel's usual bound -- and
    is far closer to the conv with the UNROUNDED fp32 weights than a one-term launch is."""
    from mono_depth_estimation_amd import ops
    x = _bf(W.normal(3, "x", (N, Cin, H, Wd)))
    w = W.normal(3, "w", (Cout, Cin, k, k), std=(2.0 / (k * k * Cin)) ** 0.5)      # generic fp32 values: off the grid
    T = k * k
    # the flat fp32 master [O][T][I] the store would hold; a second job in the table checks the offsets
    flat = torch.zeros(64 + Cout * T * Cin, device="cuda")
    flat[64:] = w.permute(0, 2, 3, 1).reshape(-1).cuda()
    jobs, nblocks = ops.pack_jobs([(0, 8, 1, 8), (64, Cout, T, Cin)], "cuda")
    w2 = torch.full((2 * flat.numel(),), 7.0, dtype=ACT, device="cuda")
    ops.pack_split_batch(flat, w2, jobs, nblocks, transposed=transposed)
    body = w2[128:].float().cpu()
    if transposed:
        hi, lo = body.view(Cin, 2, T, Cout)[:, 0], body.view(Cin, 2, T, Cout)[:, 1]          # [I][T][O]
        back = lambda t: t.permute(2, 0, 1).reshape(Cout, Cin, k, k)
    else:
        hi, lo = body.view(Cout, 2, T, Cin)[:, 0], body.view(Cout, 2, T, Cin)[:, 1]          # [O][T][I]
        back = lambda t: t.permute(0, 2, 1).reshape(Cout, Cin, k, k)
    assert torch.equal(back(hi), _bf(w))
    w_sum = back(hi) + back(lo)
    assert float(((w_sum - w).abs() / w.abs().clamp(min=1e-3)).max()) < (2.0 ** -15 if ACT == torch.bfloat16 else 2.0 ** -10)
    if transposed:
        return                                   # (the launch over [I][2T][O] is DeConvLayer's / ConvT's: tests/test_fcrn_gpu.py, test_mynet_gpu.py)
    ref2 = F.conv2d(x, w_sum, stride=s, padding=p, dilation=dil)
    ref32 = F.conv2d(x, w, stride=s, padding=p, dilation=dil)
    OH, OW = ref2.shape[2:]
    xd = _nhwc(x)
    out2 = torch.full((N, OH, OW, Cout), 7.0, dtype=ACT, device="cuda")
    out1 = torch.full((N, OH, OW, Cout), 7.0, dtype=ACT, device="cuda")
    d = ops.fwd_desc(N, H, Wd, Cin, Cin, xd.numel() * 2, k, s, p, Cout, Cout, dil=dil)
    ops.conv_gemm_eval(d, xd, w2[128:], out2)
    assert len(ops.split_descs(d)) == (2 if 2 * T > 32 else 1)
    ops.conv_gemm(d, xd, _pack_fwd(w), out1)
    torch.cuda.synchronize()
    # (two launches: the partial sum passes through one more 16-bit rounding, at the magnitude of the PARTIAL sum)
    _assert_close(_nchw(out2), ref2, "two-term conv", tol=2.0 ** -8 if 2 * T <= 32 else 2.0 ** -6)
    # against the fp32-weight conv: MEAN signed error per output channel (rounding noise averages out, a weight error does not)
    e2 = (_nchw(out2) - ref32).mean((0, 2, 3)).abs().mean() / ref32.abs().mean()
    e1 = (_nchw(out1) - ref32).mean((0, 2, 3)).abs().mean() / ref32.abs().mean()
    print("two-term / one-term channel-mean error vs the fp32-weight conv: %.2e / %.2e" % (float(e2), float(e1)))


@pytest.mark.parametrize("N,h,w,Cin,Cout,k,s,p", [
    (2, 12, 16, 64, 64, 2, 2, 0),       # FCRN.py:81 deconv2
    (2, 6, 8, 72, 40, 4, 2, 1),         # MyNet.py:61-63: k 4, p 1, channel counts off the 64 grid
    (1, 7, 9, 64, 64, 3, 4, 0),         # Eigen.py:79: stride 4, phases without a tap
])
def test_transposed_conv_forward_two_term_weight_shadow(N, h, w, Cin, Cout, k, s, p):
    """A ConvTranspose2d's forward = the output phases of the strided convolution's input gradient over the TRANSPOSED packing
    [I][T][O] (engine.DeConvLayer, graph.ConvT); in eval mode over the two-term operand [I][2T][O] (mde_pack_split_batch,
    transposed).  Against F.conv_transpose2d with the weights hi + lo the operand holds."""
    from mono_depth_estimation_amd import ops
    x = _bf(W.normal(5, "x", (N, Cin, h, w)))
    wt = W.normal(5, "wt", (Cin, Cout, k, k), std=(2.0 / (k * k * Cout)) ** 0.5)         # nn.ConvTranspose2d layout [Cin][Cout][k][k]
    T = k * k
    H2, W2 = (h - 1) * s - 2 * p + k, (w - 1) * s - 2 * p + k
    # stored as the strided conv's [O = Cin][T][I = Cout] master
    flat = wt.permute(0, 2, 3, 1).reshape(-1).contiguous().cuda()
    jobs, nblocks = ops.pack_jobs([(0, Cin, T, Cout)], "cuda")
    w2d = torch.zeros(2 * flat.numel(), dtype=ACT, device="cuda")
    ops.pack_split_batch(flat, w2d, jobs, nblocks, transposed=True)
    body = w2d.float().cpu().view(Cout, 2, T, Cin)                                       # [I][2][T][O]
    w_sum = (body[:, 0] + body[:, 1]).permute(2, 0, 1).reshape(Cin, Cout, k, k)
    assert float(((w_sum - wt).abs() / wt.abs().clamp(min=1e-3)).max()) < (2.0 ** -15 if ACT == torch.bfloat16 else 2.0 ** -10)
    ref = F.conv_transpose2d(x, w_sum, stride=s, padding=p)
    assert ref.shape[2:] == (H2, W2)
    xd = _nhwc(x)
    out = torch.zeros(N, H2, W2, Cout, dtype=ACT, device="cuda")
    descs, zero_fill = ops.dgrad_descs(N, H2, W2, Cout, Cout, h, w, Cin, Cin, xd.numel() * 2, k, s, p)
    for d in descs:
        ops.conv_gemm_eval(d, xd, w2d, out)
    torch.cuda.synchronize()
    _assert_close(_nchw(out), ref, "two-term transposed conv")


@pytest.mark.parametrize("N,H,Wd,Cin,Cout,k,dil", [(2, 37, 53, 64, 32, 3, 1), (1, 40, 64, 40, 32, 3, 1), (2, 33, 47, 128, 24, 3, 1),
                                                  (1, 29, 31, 64, 8, 3, 2), (2, 21, 40, 32, 32, 5, 1), (1, 16, 16, 64, 16, 3, 1),
                                                  (2, 37, 53, 32, 16, 1, 1), (1, 40, 64, 16, 8, 1, 1), (3, 19, 23, 128, 32, 1, 1)])
def test_narrow_halo_tile_for_32_columns(N, H, Wd, Cin, Cout, k, dil, conv_form):
    """The 128-pixel halo form with a 32-COLUMN tile (conv_gemm_nt<128, 32, 256, ..., HALO>: the full-resolution decoder layers of
    BTS and MiDaS, <= 32 output channels over millions of pixels -- the library takes it there by itself; MDE_CONV_NARROW=2 forces
    it at test sizes; k = 1: the plain single-buffer 128 x 32 tile of the full-resolution 1x1 chains): forward with BatchNorm
    statistics, the fused bias + ELU epilogue, and an accumulating launch."""
    import os
    from mono_depth_estimation_amd import ops
    if conv_form != "auto":
        pytest.skip("one form is under test here")
    old = os.environ.get("MDE_CONV_NARROW")
    os.environ["MDE_CONV_NARROW"] = "2"
    try:
        g = torch.Generator().manual_seed(Cin + Cout + k)
        p = dil * (k // 2)
        x = _bf(torch.randn(N, Cin, H, Wd, generator=g))
        w = _bf(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (k * k * Cin)) ** 0.5)
        b = torch.randn(Cout, generator=g) * 0.5
        ref = F.conv2d(x, w, padding=p, dilation=dil)
        xd, wd = _nhwc(x), _pack_fwd(w)
        d = ops.fwd_desc(N, H, Wd, Cin, Cin, xd.numel() * 2, k, 1, p, Cout, Cout, dil=dil)
        out = torch.full((N, H, Wd, Cout), 7.0, dtype=ACT, device="cuda")
        stats = ops.new_stat_buffer(Cout)
        ops.conv_gemm(d, xd, wd, out, stats)
        _assert_close(_nchw(out), ref, "narrow fwd")
        st = stats.sum(0).cpu()
        r1, r2 = ref.sum((0, 2, 3)), (ref ** 2).sum((0, 2, 3))
        assert torch.allclose(st[0], r1, rtol=2e-3, atol=2e-2 * r2.max().sqrt().item()) and torch.allclose(st[1], r2, rtol=2e-3, atol=2e-3 * r2.max().item())
        out2 = torch.empty_like(out)
        ops.conv_gemm(d, xd, wd, out2, bias=b.cuda(), act="elu")
        _assert_close(_nchw(out2), F.elu(ref + b.view(1, -1, 1, 1)), "narrow fwd + bias + ELU")
        d.accumulate = 1
        base = _bf(torch.randn(N, Cout, H, Wd, generator=g))
        out3 = _nhwc(base)
        ops.conv_gemm(d, xd, wd, out3)
        _assert_close(_nchw(out3), ref + base, "narrow accumulate", tol=2.0 ** -7)
        # into a channel slice of a wider tensor (the decoders' concatenation buffers)
        d.accumulate = 0
        wide = torch.zeros(N, H, Wd, Cout + 16, dtype=ACT, device="cuda")
        d2 = ops.fwd_desc(N, H, Wd, Cin, Cin, xd.numel() * 2, k, 1, p, Cout, Cout + 16, dil=dil)
        ops.conv_gemm(d2, xd, wd, wide[..., 8:])
        _assert_close(_nchw(wide[..., 8:8 + Cout]), ref, "narrow fwd into a slice")
        assert float(wide[..., :8].float().abs().max()) == 0 and float(wide[..., 8 + Cout:].float().abs().max()) == 0
    finally:
        if old is None:
            os.environ.pop("MDE_CONV_NARROW", None)
        else:
            os.environ["MDE_CONV_NARROW"] = old
