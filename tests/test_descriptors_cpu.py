"""Host logic on CPU: the convolution descriptors (tap tables, output phases, weight-gradient
gathers) are interpreted by a tiny numpy model of the kernels' contract
    out[pix][col] = sum_t sum_c in[(gy*sy+dy[t], gx*sx+dx[t])][c] * w[col][wtap[t]][c]
and compared with torch conv2d / autograd — so forward, strided dgrad (as output phases), the
phase-decomposed up-projection (reference FCRN.py:31-44,170-198) and their wgrad gathers are
validated without a GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mono_depth_estimation_amd import ops


def run_desc(d, x, w, out):
    """numpy interpreter of mde_conv_gemm (x: [N,H,W,ld], w: [ncols, wtaps_total, C])."""
    N = d.N
    for t in range(d.ntaps):
        dy, dx, wt = d.dy[t], d.dx[t], d.wtap[t]
        for gy in range(d.GH):
            iy = gy * d.sy + dy
            if not 0 <= iy < d.H:
                continue
            for gx in range(d.GW):
                ix = gx * d.sx + dx
                if not 0 <= ix < d.W:
                    continue
                v = x[:, iy, ix, :d.C] @ w[:d.ncols, wt, :].T          # [N, ncols]
                out[:, gy * d.osy + d.ooy, gx * d.osx + d.oox, :d.ncols] += v
    return out


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy().astype(np.float64)


@pytest.mark.parametrize("H,W,k,s,p", [(7, 9, 3, 1, 1), (8, 10, 3, 2, 1), (7, 9, 1, 2, 0), (6, 6, 5, 1, 2), (9, 7, 5, 2, 2)])
def test_forward_and_dgrad_descriptors(H, W, k, s, p):
    torch.manual_seed(0)
    N, Cin, Cout = 2, 64, 3
    x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, dtype=torch.float64)
    y = F.conv2d(x, w, stride=s, padding=p)
    OH, OW = y.shape[2:]
    d = ops.fwd_desc(N, H, W, Cin, Cin, 0, k, s, p, Cout, Cout)
    assert (d.GH, d.GW) == (OH, OW)
    wf = w.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin).numpy()
    got = run_desc(d, nhwc(x.detach()), wf, np.zeros((N, OH, OW, Cout)))
    assert np.allclose(got, nhwc(y.detach()), atol=1e-9)
    dy = torch.randn_like(y)
    y.backward(dy)
    wd = w.permute(1, 2, 3, 0).reshape(Cin, k * k, Cout).numpy()
    descs, zero = ops.dgrad_descs(N, H, W, Cin, Cin, OH, OW, Cout, 64, 0, k, s, p)
    dyp = np.zeros((N, OH, OW, 64))
    dyp[..., :Cout] = nhwc(dy)
    wdp = np.zeros((Cin, k * k, 64))
    wdp[..., :Cout] = wd
    dx = np.zeros((N, H, W, Cin))
    for dd in descs:
        run_desc(dd, dyp, wdp, dx)
    assert np.allclose(dx, nhwc(x.grad), atol=1e-9)
    assert zero == (s > 1 and k == 1)          # 1x1 stride 2 leaves three empty phases to zero-fill
    assert len(descs) == (1 if k == 1 else s * s)


def _unpool(x):
    n, c, h, w = x.shape
    u = x.new_zeros(n, c, 2 * h, 2 * w)
    u[:, :, ::2, ::2] = x
    return u


def test_upproj_phase_descriptors():
    """Unpool + 5x5/pad 2 == four 3x3/2x3/3x2/2x2 phases over the un-stuffed input; its input
    gradient == one stride-2 5x5 gather over dY (25 useful taps instead of 100 mostly-zero ones)."""
    torch.manual_seed(1)
    N, Cin, Cout, h, w = 1, 64, 4, 5, 6
    x = torch.randn(N, Cin, h, w, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(Cout, Cin, 5, 5, dtype=torch.float64)
    y = F.conv2d(_unpool(x), wt, padding=2)
    descs = ops.upproj_fwd_descs(N, h, w, Cin, Cin, 0, Cout, Cout)
    assert sorted(d.ntaps for d in descs) == [4, 6, 6, 9]
    wf = wt.permute(0, 2, 3, 1).reshape(Cout, 25, Cin).numpy()
    out = np.zeros((N, 2 * h, 2 * w, Cout))
    for d in descs:
        run_desc(d, nhwc(x.detach()), wf, out)
    assert np.allclose(out, nhwc(y.detach()), atol=1e-9)
    dy = torch.randn_like(y)
    y.backward(dy)
    dd = ops.upproj_dgrad_desc(N, h, w, Cin, Cin, 64, 64, 0)
    dyp = np.zeros((N, 2 * h, 2 * w, 64))
    dyp[..., :Cout] = nhwc(dy)
    wdp = np.zeros((Cin, 25, 64))
    wdp[..., :Cout] = wt.permute(1, 2, 3, 0).reshape(Cin, 25, Cout).numpy()
    dx = run_desc(dd, dyp, wdp, np.zeros((N, h, w, Cin)))
    assert np.allclose(dx, nhwc(x.grad), atol=1e-9)


def run_wgrad(d, direct, gathered):
    rows_g = bool(d.rows_from_gathered)
    R, Cc = (d.Cg, d.Cd) if rows_g else (d.Cd, d.Cg)
    dw = np.zeros((R, d.otaps_total, Cc))
    for t in range(d.ntaps):
        for gy in range(d.GH):
            iy = gy * d.sy + d.dy[t]
            if not 0 <= iy < d.H:
                continue
            for gx in range(d.GW):
                ix = gx * d.sx + d.dx[t]
                if not 0 <= ix < d.W:
                    continue
                a, b = direct[:, gy, gx, :d.Cd], gathered[:, iy, ix, :d.Cg]
                dw[:, d.otap[t], :] += (b.T @ a) if rows_g else (a.T @ b)
    return dw


def test_wgrad_descriptors():
    torch.manual_seed(2)
    N, Cin, Cout, H, W, k, s, p = 1, 64, 64, 7, 8, 3, 2, 1
    x = torch.randn(N, Cin, H, W, dtype=torch.float64)
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, stride=s, padding=p)
    dy = torch.randn_like(y)
    y.backward(dy)
    OH, OW = y.shape[2:]
    d = ops.conv_wgrad_desc(N, H, W, Cin, Cin, 0, OH, OW, Cout, Cout, 0, k, s, p, 1)
    got = run_wgrad(d, nhwc(dy), nhwc(x))
    assert np.allclose(got, w.grad.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).numpy(), atol=1e-9)
    # up-projection: direct = x (small grid), gathered = dY (stride 2), rows from the gathered side
    h, ww = 4, 5
    x = torch.randn(N, 64, h, ww, dtype=torch.float64)
    w5 = torch.zeros(64, 64, 5, 5, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(_unpool(x), w5, padding=2)
    dy = torch.randn_like(y)
    y.backward(dy)
    d = ops.upproj_wgrad_desc(N, h, ww, 64, 64, 0, 64, 64, 0, 1)
    got = run_wgrad(d, nhwc(x), nhwc(dy))
    assert np.allclose(got, w5.grad.permute(0, 2, 3, 1).reshape(64, 25, 64).numpy(), atol=1e-9)


def test_ksplit_and_buckets():
    from mono_depth_estimation_amd import dp
    assert ops.choose_ksplit(2457600, 1, 1, 9, wg_per_cu=4, tile_elems=64 * 64) >= 32   # huge pixel range, tiny tile grid: split hard
    assert ops.choose_ksplit(9600, 8, 8, 25) <= 2               # 1600 tiles already cover the chip 3x
    assert ops.choose_ksplit(640, 1, 1, 1) == 1                 # never fewer than 8 K-steps per workgroup
    # measured optima on MI355X (tools/conv_microbench.py wgrad, MB_KS sweep): one nearly full round of
    # workgroups beats several rounds of small ones because every split adds an fp32 atomic per output
    assert 50 <= ops.choose_ksplit(153600, 1, 1, 9) <= 60       # 75 us at 56 vs 127 us at 227
    assert 12 <= ops.choose_ksplit(38400, 2, 2, 9) <= 16        # 75 us at 14 vs 97 us at 28
    assert 100 <= ops.choose_ksplit(614400, 1, 1, 9, wg_per_cu=4, tile_elems=64 * 64) <= 256   # 89 us at 113
    for px, rt, ct, taps in [(38400, 2, 2, 9), (153600, 2, 2, 25), (9600, 4, 4, 9), (614400, 1, 1, 25)]:
        ks = ops.choose_ksplit(px, rt, ct, taps)
        assert px // (64 * ks) >= 8
        t = ops.wgrad_time_model(px, rt, ct, taps, ks)
        assert all(t <= ops.wgrad_time_model(px, rt, ct, taps, k) * (1 + 1e-6) for k in range(1, 65) if px // (64 * k) >= 8)
    b = dp.make_buckets(1000, [0, 100, 300, 600, 900], 800)
    assert b == [(600, 1000), (300, 600), (100, 300), (0, 100)]
    assert dp.make_buckets(10, [], 1 << 20) == [(0, 10)]
    cover = sorted(dp.make_buckets(12345, list(range(0, 12345, 777)), 4000))
    assert cover[0][0] == 0 and cover[-1][1] == 12345 and all(a[1] == b_[0] for a, b_ in zip(cover, cover[1:]))


def test_gradient_writes_into_slices_need_a_concatenation_root():
    """Act.gw is shared between a root and its slices: a partial first write is only sound when the root's gradient was
    zeroed and flagged before the backward pass, i.e. the root came from TapeEngine.buf()."""
    import torch
    from mono_depth_estimation_amd.engine import Act
    from mono_depth_estimation_amd.graph import _take
    plain = Act(torch.device("cpu"), 1, 2, 2, 16)
    assert _take(plain) is False and _take(plain) is True             # whole tensors: first write, then accumulate
    with pytest.raises(AssertionError, match="concatenation target"):
        _take(plain.slice(0, 8))
    cat = Act(torch.device("cpu"), 1, 2, 2, 16)
    cat.concat_root = True                                             # what TapeEngine.buf() sets (and zeroes .g, gw = True)
    cat.gw = True
    assert _take(cat.slice(8, 8)) is True and _take(cat.slice(0, 8).slice(0, 8)) is True
