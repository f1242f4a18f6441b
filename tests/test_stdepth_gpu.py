"""GPU parity of the stdepth composite criterion (csrc/stdepth_loss.hip) through the drop-in closure
mono_depth_estimation_amd.stdepth.setup_criterion — against the vectors minted from the reference's own
setup_criterion body (tests/golden/stdepth.npz) and against the CPU oracle on other shapes.
fp32: losses / terms rtol 2e-5, composite 1e-6 abs, gradients rtol 1e-4 (SSIM terms 5e-4: different summation
order of the 121-tap window)."""
import types

import numpy as np
import pytest
import torch

from oracle import stdepth as OS
from oracle import weights as W

pytestmark = pytest.mark.gpu

CASES = [("mae+composite", True), ("silma", True), ("silms+fbdivergence", True), ("mse", True),
         ("mae+composite+ssim", True), ("allssim+colorssim", True), ("silma+mse+fbdivergence", False),
                 ("mae+composite", True, 20), ("silma+allssim+composite+ssim", True, 20)]   # laina's default: 20 channels, single_layer


def _method(loss):
    return types.SimpleNamespace(loss=loss, variance_focus=0.85, depth_loss_weight=10.0, comp_loss_weight=2.0,
                                 fbdiv_loss_weight=0.2, ssim_loss_weight=2.0)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _close(got, ref, rtol, atol, what):
    got = torch.as_tensor(got).detach().float().cpu()
    ref = torch.as_tensor(ref).detach().float().cpu()
    err = (got - ref).abs()
    bad = (~(err <= atol + rtol * ref.abs())).sum().item()
    assert bad == 0, "%s: %d/%d outside tolerance, max err %.4g" % (what, bad, ref.numel(), err.max().item())


def _batch(seed, C, N, H, Wd, single=None):
    pred = W.uniform(seed, "pred", (N, C, H, Wd), -0.1, 1.1)
    targ = W.uniform(seed, "targ", (N, C, H, Wd), 0.0, 1.0)
    rgba = W.uniform(seed, "rgba", (N, 4, H, Wd), 0.0, 1.0)
    rgba[:, 3] = rgba[:, 3].masked_fill(W.uniform(seed, "hole", (N, H, Wd)) < 0.3, 0.0)
    single = (C == 10) if single is None else single
    d = slice(8, 10) if single else slice(16, 20)
    targ[:, d] = targ[:, d].masked_fill(W.uniform(seed, "dhole", targ[:, d].shape) < 0.2, 0.0)
    pred[:, d] = pred[:, d].abs() + 0.05
    return pred, targ, rgba


@pytest.mark.parametrize("i", range(len(CASES)))
def test_stdepth_golden(golden, i):
    from mono_depth_estimation_amd import stdepth
    g = golden("stdepth")
    loss, single = CASES[i][:2]
    C = CASES[i][2] if len(CASES[i]) > 2 else (10 if single else 20)
    key = "c20s" if (single and C == 20) else "c%d" % C
    pred, targ, rgba = [_t(g["%s_%s" % (key, k)]).cuda() for k in ("pred", "targ", "rgba")]
    crit = stdepth.setup_criterion(_method(loss), single_layer=single)
    p = pred.clone().requires_grad_(True)
    total, full, terms = crit(p, targ, rgba, return_composited=True, return_loss_dict=True)
    total.backward()
    assert list(terms.keys()) == [str(n) for n in g["k%d_names" % i]]
    _close(total, g["k%d_loss" % i], 2e-5, 1e-6, "total")
    _close(torch.stack(list(terms.values())), g["k%d_terms" % i], 2e-5, 1e-6, "terms")
    _close(full, g["k%d_full" % i], 0, 2e-6, "pred_full")
    ssim = "ssim" in loss
    _close(p.grad, g["k%d_grad" % i], 5e-4 if ssim else 1e-4, 2e-7 if ssim else 1e-8, "grad")
    assert len(crit(pred, targ, rgba)) == 1


@pytest.mark.parametrize("loss,single,N,H,Wd", [
    ("mae+composite+ssim+fbdivergence", True, 2, 37, 83),        # partial tiles both ways
    ("silms+mse+allssim+colorssim", True, 1, 16, 64),            # exactly one tile
    ("silma+mae+colorssim+fbdivergence", False, 2, 21, 70),
    ("mae+composite", True, 3, 9, 5),                            # smaller than the SSIM halo
    ("mae+composite+ssim+allssim", True, 1, 5, 7),               # SSIM on a map smaller than its 11-tap window
    ("colorssim+fbdivergence", False, 1, 33, 129),               # one pixel past a tile in both directions
])
def test_stdepth_vs_oracle(loss, single, N, H, Wd):
    from mono_depth_estimation_amd import stdepth
    C = 10 if single else 20
    pred, targ, rgba = _batch(70 + H, C, N, H, Wd)
    if not single:
        pred[:, 17] = pred[:, 16]                                 # depth ties: the stable order decides
    po = pred.clone().requires_grad_(True)
    lo, fo, to = OS.stdepth_loss(po, targ, rgba, loss, single)
    lo.backward()
    ph = pred.cuda().requires_grad_(True)
    lh, fh, th = stdepth.setup_criterion(_method(loss), single)(ph, targ.cuda(), rgba.cuda(), True, True)
    (3.0 * lh).backward()
    assert list(th.keys()) == list(to.keys())
    _close(lh, lo, 2e-5, 1e-6, "total")
    _close(torch.stack(list(th.values())), torch.stack(list(to.values())), 2e-5, 1e-6, "terms")
    _close(fh, fo, 0, 2e-6, "pred_full")
    _close(ph.grad / 3.0, po.grad, 5e-4, 2e-7, "grad")


def test_stdepth_full_size_and_edges():
    """BASELINE-sized maps (480x640): value against the oracle; an all-transparent batch gives the reference's
    NaN (mean of nothing) for the colour terms while the nan_to_num'ed terms stay finite."""
    from mono_depth_estimation_amd import stdepth
    pred, targ, rgba = _batch(90, 10, 2, 480, 640)
    lo, _, to = OS.stdepth_loss(pred, targ, rgba, "silma+composite+ssim", True)
    crit = stdepth.setup_criterion(_method("silma+composite+ssim"), True)
    ph = pred.cuda().requires_grad_(True)
    lh, th = crit(ph, targ.cuda(), rgba.cuda(), return_loss_dict=True)
    lh.backward()
    _close(lh, lo, 2e-5, 1e-6, "total")
    _close(torch.stack(list(th.values())), torch.stack(list(to.values())), 2e-5, 1e-6, "terms")
    assert bool(torch.isfinite(ph.grad).all()) and float(ph.grad.abs().sum()) > 0
    clear = rgba.clone()
    clear[:, 3] = 0
    l2, t2 = stdepth.setup_criterion(_method("silma"), True)(pred.cuda(), targ.cuda(), clear.cuda(), return_loss_dict=True)
    assert bool(torch.isnan(l2)) and bool(torch.isfinite(t2["depth_silog"])) and bool(torch.isnan(t2["color_mae"]))


def test_stdepth_rejects():
    from mono_depth_estimation_amd import stdepth
    pred, targ, rgba = [x.cuda() for x in _batch(91, 20, 1, 8, 8)]
    with pytest.raises(ValueError):
        stdepth.setup_criterion(_method("mae+composite"), False)(pred, targ, rgba)
    l20, = stdepth.setup_criterion(_method("mae"), True)(pred, targ, rgba)      # 20 channels, single-layer layout: fine
    assert bool(torch.isfinite(l20))
    with pytest.raises(ValueError):
        stdepth.setup_criterion(_method("mae"), True)(pred[:, :12], targ[:, :12], rgba)
    with pytest.raises(ValueError):
        stdepth.setup_criterion(_method("nothing"), False)(pred, targ, rgba)
    with pytest.raises(RuntimeError):
        stdepth.setup_criterion(_method("mae"), False)(pred.cpu(), targ.cpu(), rgba.cpu())
