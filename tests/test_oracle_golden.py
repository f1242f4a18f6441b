"""Pin the CPU oracle against vectors minted from the reference's own code
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import fcrn as ofcrn
from oracle import losses as L
from oracle import metrics as M
from oracle import weights as W


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _lg(fn, pred, *rest):
    p = pred.clone().requires_grad_(True)
    out = fn(p, *rest)
    out.backward()
    return out.detach(), p.grad


def _close(a, b, rtol=2e-5, atol=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.allclose(a, b, rtol=rtol, atol=atol), float(np.abs(a - b).max())


@pytest.mark.parametrize("lam", [0.85, 0.5])
def test_silog(golden, lam):
    g = golden("losses")
    l, gr = _lg(lambda e: L.silog(e, _t(g["g1_gt"]), lam), _t(g["g1_est"]))
    _close(l, g["g1_silog_%g" % lam])
    _close(gr, g["g1_silog_%g_grad" % lam], atol=1e-8)


@pytest.mark.parametrize("name,fn", [
    ("masked_depth", L.masked_depth), ("masked_mse", L.masked_mse),
    ("masked_l1", L.masked_l1), ("berhu", L.berhu),
    ("procrustes", L.trimmed_procrustes),
])
def test_pointwise_losses(golden, name, fn):
    g = golden("losses")
    l, gr = _lg(fn, _t(g["g2_pred"]), _t(g["g2_tgt"]))
    _close(l, g["g2_" + name])
    _close(gr, g["g2_%s_grad" % name], atol=1e-8)


@pytest.mark.parametrize("kind", ["ssimse", "ssil1", "mse", "l1", "trim", "ssitrim"])
def test_midas(golden, kind):
    g = golden("losses")
    l, gr = _lg(lambda p, t: L.midas_loss(p, t, 0.5, 4, kind), _t(g["g2_pred"]), _t(g["g2_tgt"]))
    _close(l, g["g2_midas_" + kind])
    _close(gr, g["g2_midas_%s_grad" % kind], rtol=1e-4, atol=1e-8)


def test_trim_defect_is_reproduced(golden):
    """the reference's trimmed MAE never trims -> equals L1/(2M)."""
    g = golden("losses")
    assert float(g["g2_midas_trim"]) == pytest.approx(float(g["g2_midas_l1"]), rel=1e-6)


def test_scale_shift_and_gradient(golden):
    g = golden("losses")
    p, t = _t(g["g2_pred"])[:, 0], _t(g["g2_tgt"])[:, 0]
    m = (t > 0).float()
    s, sh = L.scale_and_shift(p, t, m)
    _close(s, g["g2_scale"], rtol=1e-4)
    _close(sh, g["g2_shift"], rtol=1e-4)
    for tag, bb in (("batch", True), ("image", False)):
        l, gr = _lg(lambda q: L.gradient_multiscale(q, t, m, 4, bb), p)
        _close(l, g["g2_gradient_" + tag])
        _close(gr, g["g2_gradient_%s_grad" % tag], atol=1e-8)


def test_metrics(golden):
    g = golden("metrics")
    got = M.compute(_t(g["pred"]), _t(g["tgt"]))
    for k in M.PINNED:
        _close(got[k], g[k])


def test_unpool_and_upproj(golden):
    g = golden("upproj")
    x = _t(g["x"])
    assert np.array_equal(ofcrn.unpool2x(x).numpy(), g["unpool"])
    m = ofcrn.UpProjModule(16)
    m.load_state_dict({k[3:]: _t(g[k]) for k in g.files if k.startswith("sd.")})
    m.train()
    xi = x.clone().requires_grad_(True)
    y = m(xi)
    y.backward(_t(g["gy"]))
    _close(y.detach(), g["y"], rtol=1e-4, atol=1e-5)
    _close(xi.grad, g["gx"], rtol=1e-4, atol=1e-5)
    for k, p in m.named_parameters():
        _close(p.grad, g["grad." + k], rtol=1e-3, atol=1e-4)
    for k, v in m.state_dict().items():
        if "running" in k:
            _close(v, g["after." + k], rtol=1e-5)


@pytest.fixture(scope="module")
def fcrn_oracle():
    torch.set_num_threads(8)
    net = ofcrn.FCRNOracle(layers=50, output_size=(96, 128), out_channels=1)
    W.fcrn_fixture_state(net, 5)
    rgb, tgt = W.synthetic_batch(5, 2, 96, 128)
    W.calibrate_running_stats(net, rgb)
    return net, rgb, tgt


def test_fcrn50_keys_and_params(golden, fcrn_oracle):
    g = golden("fcrn50")
    net = fcrn_oracle[0]
    assert len(net.state_dict()) == int(g["n_state_keys"]) == 397
    assert sum(p.numel() for p in net.parameters()) == int(g["n_params"]) == 63563008
    assert [k for k, _ in net.named_parameters()] == list(g["grad_names"])


def test_fcrn50_eval(golden, fcrn_oracle):
    g = golden("fcrn50")
    net, rgb, tgt = fcrn_oracle
    net.eval()
    with torch.no_grad():
        y = net(rgb)
    _close(y, g["eval_out"], rtol=1e-4, atol=2e-5)
    _close(L.silog(y, tgt), g["eval_silog"], rtol=1e-4)
    got = M.compute(y, tgt)
    for k in ("absrel", "rmse", "delta1"):
        _close(got[k], g["eval_" + k], rtol=1e-4)


def test_fcrn50_train_fwd_bwd(golden, fcrn_oracle):
    g = golden("fcrn50")
    net, rgb, tgt = fcrn_oracle
    saved = {k: v.clone() for k, v in net.state_dict().items()}  # running stats move in train mode
    net.train()
    net.zero_grad()
    y = net(rgb)
    loss = L.silog(y, tgt)
    loss.backward()
    net.load_state_dict(saved)
    _close(y.detach(), g["train_out"], rtol=1e-4, atol=2e-5)
    _close(loss.detach(), g["train_silog"], rtol=1e-4)
    gn = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])
    assert np.allclose(gn, g["grad_norm"], rtol=2e-3, atol=1e-6), np.abs(gn / g["grad_norm"] - 1).max()
    _close(net.conv3.weight.grad, g["grad_conv3"], rtol=1e-3, atol=1e-5)
    _close(net.conv1.weight.grad[:8], g["grad_conv1_slice"], rtol=2e-3, atol=1e-3)


def test_fcrn50_conditioned_eval(golden):
    """The well-conditioned fixture (the one the 1e-4 AbsRel bound is asserted on)."""
    g = golden("fcrn50_cond")
    net = ofcrn.FCRNOracle(layers=50, output_size=(96, 128), out_channels=1)
    W.fcrn_conditioned_state(net, 7)
    rgb, tgt = W.synthetic_batch(7, 2, 96, 128)
    W.calibrate_running_stats(net, rgb)
    net.eval()
    with torch.no_grad():
        y = net(rgb)
    _close(y, g["eval_out"], rtol=1e-4, atol=2e-5)
    got = M.compute(y, tgt)
    for k in ("absrel", "rmse", "delta1", "delta2", "delta3", "log10"):
        _close(got[k], g["eval_" + k], rtol=1e-4)


# ---------------------------------------------------------------- G3: VNL_Loss, WCEL_Loss, ModelLoss
@pytest.mark.parametrize("select", [True, False])
def test_vnl(golden, select):
    g = golden("vnl")
    fx = float(g["g3_fx"])
    p123 = [g["g3_p123"][i].astype(np.int64) for i in range(3)]
    l, gr = _lg(lambda p: L.vnl(_t(g["g3_gt"]), p, p123, fx, fx, select=select), _t(g["g3_pred"]))
    _close(l, g["g3_vnl_%d" % select])
    _close(gr, g["g3_vnl_%d_grad" % select], rtol=1e-4, atol=1e-7)
    assert np.abs(g["g3_vnl_%d_grad" % select]).sum() > 0


def test_vnl_sampling_stream(golden):
    """The sample indices are part of the result: same seed -> the reference's own draw."""
    g = golden("vnl")
    np.random.seed(1234)
    p = L.vnl_select_index(48, 64)
    assert all(np.array_equal(p[i], g["g3_p123"][i]) for i in range(3))
    np.random.seed(99)
    p = L.vnl_select_index(24, 32)
    assert all(np.array_equal(p[i], g["g3_model_p123"][i]) for i in range(3))


def _g3_logit(g):
    return W.normal(int(g["g3_logit_seed"]), "logit", (2, 150, 24, 32), std=2.0)


def test_wcel(golden):
    g = golden("vnl")
    bins, dgt = _t(g["g3_bins"]), _t(g["g3_dgt"])
    assert int((bins > 149).sum()) > 0            # invalid labels are exercised
    l, gr = _lg(lambda x: L.wcel(x, bins, dgt, L.wcel_weight(150)), _g3_logit(g))
    _close(l, g["g3_wcel"])
    gr = gr.numpy()
    _close(gr[:, ::7, ::3, ::5], g["g3_wcel_grad_sample"], rtol=1e-4, atol=1e-9)
    _close(gr.sum((2, 3)), g["g3_wcel_grad_csum"], rtol=1e-4, atol=1e-7)
    _close(gr.sum(1), g["g3_wcel_grad_psum"], rtol=1e-4, atol=1e-7)
    _close(np.abs(gr).sum((2, 3)), g["g3_wcel_grad_abs"], rtol=1e-4, atol=1e-7)


def test_model_loss(golden):
    g = golden("vnl")
    bins, dgt, border = _t(g["g3_bins"]), _t(g["g3_dgt"]), _t(g["g3_border"])
    p123 = [g["g3_model_p123"][i].astype(np.int64) for i in range(3)]

    def model(x):
        depth = 10 ** (torch.softmax(x, 1).permute(0, 2, 3, 1) * border).sum(3, keepdim=True)
        return L.model_loss(depth.permute(0, 3, 1, 2), x, bins, dgt, L.wcel_weight(150), p123, 30.0, 30.0, 6)
    l, gr = _lg(model, _g3_logit(g))
    _close(l, g["g3_model"])
    _close(gr.numpy()[:, ::7, ::3, ::5], g["g3_model_grad_sample"], rtol=2e-4, atol=1e-8)
    _close(np.abs(gr.numpy()).sum((2, 3)), g["g3_model_grad_abs"], rtol=2e-4, atol=1e-7)


# ---------------------------------------------------------------- G6: the stdepth composite criterion
STDEPTH_CASES = [("mae+composite", True), ("silma", True), ("silms+fbdivergence", True), ("mse", True),
                 ("mae+composite+ssim", True), ("allssim+colorssim", True), ("silma+mse+fbdivergence", False),
                 ("mae+composite", True, 20), ("silma+allssim+composite+ssim", True, 20)]   # laina's default: 20 channels, single_layer


@pytest.mark.parametrize("i", range(len(STDEPTH_CASES)))
def test_stdepth_loss(golden, i):
    from oracle import stdepth as S
    g = golden("stdepth")
    loss, single = STDEPTH_CASES[i][:2]
    C = STDEPTH_CASES[i][2] if len(STDEPTH_CASES[i]) > 2 else (10 if single else 20)
    key = "c20s" if (single and C == 20) else "c%d" % C
    pred, targ, rgba = [_t(g["%s_%s" % (key, k)]) for k in ("pred", "targ", "rgba")]
    p = pred.clone().requires_grad_(True)
    total, full, terms = S.stdepth_loss(p, targ, rgba, loss, single)
    total.backward()
    assert list(terms.keys()) == [str(n) for n in g["k%d_names" % i]]
    _close(total.detach(), g["k%d_loss" % i])
    _close(torch.stack(list(terms.values())).detach(), g["k%d_terms" % i])
    _close(full.detach(), g["k%d_full" % i], atol=1e-6)
    _close(p.grad, g["k%d_grad" % i], rtol=1e-4, atol=1e-7)


# ---------------------------------------------------------------- G5c: the other decoders (FCRN.py:68-110)
@pytest.mark.parametrize("dec", ["upconv", "deconv2", "deconv3", "fasterupproj", "fasterupconv"])
def test_fcrn50_other_decoders(golden, dec):
    g = golden("fcrn_decoders")
    net = ofcrn.FCRNOracle(layers=50, output_size=(64, 96), out_channels=1, decoder=dec)
    assert list(net.state_dict().keys()) == [str(k) for k in g[dec + "_state_keys"]]
    W.fcrn_conditioned_state(net, 8)
    rgb, tgt = W.synthetic_batch(8, 2, 64, 96)
    W.calibrate_running_stats(net, rgb)
    net.eval()
    with torch.no_grad():
        y = net(rgb)
    _close(y, g[dec + "_eval_out"], rtol=1e-4, atol=2e-5)
    got = M.compute(y, tgt)
    for k in ("absrel", "rmse", "delta1"):
        _close(got[k], g[dec + "_eval_" + k], rtol=1e-4)
    net.train()
    loss = L.silog(net(rgb), tgt)
    loss.backward()
    _close(loss.detach(), g[dec + "_train_silog"], rtol=1e-4)
    gn = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])
    assert np.allclose(gn, g[dec + "_grad_norm"], rtol=2e-3, atol=1e-6), np.abs(gn / g[dec + "_grad_norm"] - 1).max()


# ---------------------------------------------------------------- G5d: BasicBlock trunks (layers = 18 / 34)
@pytest.mark.parametrize("layers", [18, 34])
def test_fcrn_basic_trunks(golden, layers):
    g, tag = golden("fcrn_basic_trunks"), "r%d" % layers
    net = ofcrn.FCRNOracle(layers=layers, output_size=(64, 96), out_channels=1)
    assert list(net.state_dict().keys()) == [str(k) for k in g[tag + "_state_keys"]]
    W.fcrn_conditioned_state(net, 10 + layers, basic=True)
    rgb, tgt = W.synthetic_batch(10 + layers, 2, 64, 96)
    W.calibrate_running_stats(net, rgb)
    net.eval()
    with torch.no_grad():
        y = net(rgb)
    _close(y, g[tag + "_eval_out"], rtol=1e-4, atol=2e-5)
    got = M.compute(y, tgt)
    for k in ("absrel", "rmse", "delta1"):
        _close(got[k], g[tag + "_eval_" + k], rtol=1e-4)
    net.train()
    loss = L.silog(net(rgb), tgt)
    loss.backward()
    _close(loss.detach(), g[tag + "_train_silog"], rtol=1e-4)
    gn = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])
    assert np.allclose(gn, g[tag + "_grad_norm"], rtol=2e-3, atol=1e-6), np.abs(gn / g[tag + "_grad_norm"] - 1).max()


# ---------------------------------------------------------------- G5e: in_channels != 3
@pytest.mark.parametrize("cin", [4, 1])
def test_fcrn_in_channels(golden, cin):
    g, tag = golden("fcrn_in_channels"), "c%d" % cin
    net = ofcrn.FCRNOracle(layers=50, output_size=(64, 96), in_channels=cin, out_channels=1)
    W.fcrn_conditioned_state(net, 40 + cin)
    x = W.uniform(40 + cin, "x", (2, cin, 64, 96))
    _, tgt = W.synthetic_batch(40 + cin, 2, 64, 96)
    W.calibrate_running_stats(net, x)
    net.eval()
    with torch.no_grad():
        y = net(x)
    _close(y, g[tag + "_eval_out"], rtol=1e-4, atol=2e-5)
    _close(M.compute(y, tgt)["absrel"], g[tag + "_eval_absrel"], rtol=1e-4)
    net.train()
    loss = L.silog(net(x), tgt)
    loss.backward()
    _close(loss.detach(), g[tag + "_train_silog"], rtol=1e-4)
    _close(net.conv1.weight.grad, g[tag + "_conv1_grad"], rtol=2e-3, atol=2e-3 * float(np.abs(g[tag + "_conv1_grad"]).max()))


def test_ssim_metric_restatement_against_the_reference_window(golden):
    """'ssim' is a torchmetrics call in the reference (metrics.py:123; absent here): the oracle restates torchmetrics 0.7.3's
    published definition, and tests/golden/ssim.npz anchors it to the reference's OWN SSIM code (stdepth_utils.ssim, same
    window and index) on the pixels where the two definitions coincide -- see gen_golden.gen_ssim."""
    g = golden("ssim")
    got = M.ssim(_t(g["pred"]), _t(g["tgt"]))
    assert 0.2 < float(g["ssim_interior"]) < 0.9
    _close(got, g["ssim_interior"], rtol=2e-5)
    x = _t(g["tgt"]) + 0.5
    assert abs(float(M.ssim(x, x)) - 1.0) < 1e-6
