"""End-to-end GPU parity of the HIP BTS network (mono_depth_estimation_amd.network.Bts.BtsModel) against the
CPU oracle (oracle/nets.py: bts_forward, pinned bit-for-bit to the reference's own network/Bts.py by
tests/golden/bts_net.npz), the plane-depth kernel against torch autograd, and configuration 3 at 16 x 3 x 480 x 640.

This net amplifies storage rounding (84 BatchNorms that each re-normalise a concatenation; local planar guidance divides
by a plane-ray product): rounding the fp32 ORACLE's own activations to bf16 moves its five outputs by `noise` = 3-10 %.
Tolerances are relative to that: every output within 1.5 noise + 1e-2 of the oracle and of the reference, train-mode SILog
within 2 |loss(rounding oracle) - loss(reference)| + 1 %, gradient norms within 25 % for 85 % of the tensors."""
import numpy as np
import pytest
import torch

from mono_depth_estimation_amd.ops import ACT_DTYPE as ACT        # the library's 16-bit storage type (bf16; fp16 under MDE_ACT_DTYPE=fp16)

from oracle import losses as L
from oracle import nets
from oracle import weights as W

pytestmark = pytest.mark.gpu
SIZE = (64, 96)
NAMES = ("d8", "d4", "d2", "r1", "final")


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_plane_depth_kernel_against_autograd():
    """reduction_1x1's plane tail + F.normalize + local_planar_guidance (Bts.py:105-146,228-232) in one kernel, fwd and bwd."""
    from mono_depth_estimation_amd import ops
    for up in (8, 4, 2):
        N, h, w, md = 2, 5, 7, 10.0
        x = W.normal(up, "x", (N, 3, h, w), 1.5).to(ACT).float()
        xr = x.clone().requires_grad_(True)
        ref = nets._bts_lpg(torch.cat([torch.nn.functional.normalize(nets._plane_from_params(xr, md)[:, :3], 2, 1),
                                       nets._plane_from_params(xr, md)[:, 3:4]], 1), up).unsqueeze(1) / md
        xd = torch.zeros(N, h, w, 8, dtype=ACT, device="cuda")
        xd[..., :3] = x.permute(0, 2, 3, 1).to(ACT).cuda()
        out = torch.empty(N, 1, h * up, w * up, device="cuda")
        ops.plane_depth_fwd(xd, 8, out, N, h, w, up, md)
        torch.cuda.synchronize()
        assert torch.allclose(out.cpu(), ref.detach(), rtol=2e-5, atol=1e-6)
        dy = W.normal(up, "dy", tuple(ref.shape))
        ref.backward(dy)
        dx = torch.full((N, h, w, 8), 5.0, dtype=ACT, device="cuda")
        ops.plane_depth_bwd(xd, 8, dy.cuda(), dx, 8, N, h, w, up, md)
        torch.cuda.synchronize()
        got = dx[..., :3].float().cpu().permute(0, 3, 1, 2)
        err = (got - xr.grad).abs()
        assert bool((err <= 2.0 ** -7 * xr.grad.abs() + 2.0 ** -8 * xr.grad.pow(2).mean().sqrt()).all()), float(err.max())
        assert float(dx[..., 3:].float().abs().max()) == 0.0
        # the concatenation slot: every step-th pixel of the map as one bf16 channel, and its gradient
        cat = torch.zeros(N, h * up // 2, w * up // 2, 16, dtype=ACT, device="cuda")
        ops.map_to_slot(out, cat[..., 9:], 16, N, h * up, w * up, 2)
        torch.cuda.synchronize()
        assert torch.equal(cat[..., 9].float().cpu(), out.cpu()[:, 0, ::2, ::2].to(ACT).float())
        assert float(cat[..., :9].float().abs().max()) == 0.0 and float(cat[..., 10:].float().abs().max()) == 0.0
        g = torch.zeros_like(out)
        ops.slot_to_map_add(cat[..., 9:], 16, g, N, h * up, w * up, 2)
        torch.cuda.synchronize()
        assert torch.equal(g.cpu()[:, 0, ::2, ::2], cat[..., 9].float().cpu()) and float(g[:, 0, 1::2].abs().max()) == 0.0


@pytest.fixture(scope="module")
def setup():
    from mono_depth_estimation_amd.network import Bts
    torch.manual_seed(0)
    net = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=1, encoder_version="densenet161_bts")
    sd = W.bts_fixture_state(net, 47)
    rgb, tgt = W.synthetic_batch(47, 2, *SIZE)
    P = nets.leaf_state(sd)
    with torch.no_grad():
        nets.bts_forward(P, rgb, True, momentum=1.0)
    net.load_state_dict({k: v.clone() for k, v in P.items()})
    return net.cuda(), P, rgb, tgt


def test_bts_eval_against_oracle_and_reference(setup, golden):
    net, P, rgb, tgt = setup
    g = golden("bts_net")
    net.eval()
    with torch.no_grad():
        ys = net(rgb.cuda())
        yo = nets.bts_forward(P, rgb, False)
        yq = nets.bts_forward(P, rgb, False, q=nets.bf16_round)
    assert len(ys) == 5
    for nme, y, o, q in zip(NAMES, ys, yo, yq):
        assert y.shape == (2, 1, *SIZE) and y.dtype == torch.float32 and torch.isfinite(y).all()
        noise, e_o, e_q, e_ref = _rel(q, o), _rel(y.cpu(), o), _rel(y.cpu(), q), _rel(y.cpu(), torch.from_numpy(g["eval_" + nme]))
        print("BTS eval %-5s: HIP vs fp32 oracle %.3e, vs bf16-rounding oracle %.3e, vs reference %.3e; rounding noise %.3e" % (nme, e_o, e_q, e_ref, noise))
        assert e_o < 1.5 * noise + 1e-2 and e_ref < 1.5 * noise + 1e-2, nme
    t = tgt * 10.0
    m = t > 0
    absrel = lambda d: float(((d - t).abs() / t.clamp(min=1e-9))[m].mean())
    a_ref, a_hip, a_q = absrel(torch.from_numpy(g["eval_final"])), absrel(ys[4].cpu()), absrel(yq[4])
    print("BTS eval AbsRel(final depth): reference %.5f, HIP %.5f, bf16-rounding oracle %.5f" % (a_ref, a_hip, a_q))
    assert abs(a_hip - a_ref) < 2.0 * abs(a_q - a_ref) + 5e-3


def test_bts_train_step_against_oracle_and_reference(setup, golden):
    from mono_depth_estimation_amd import criteria
    net, P0, rgb, tgt = setup
    g = golden("bts_net")
    net.train()
    net.zero_grad(set_to_none=True)
    ys = net(rgb.cuda())
    t = (tgt * 10.0).cuda()
    loss = criteria.silog_loss(0.85)(ys[4], t)
    # every head takes part in the backward pass (BTS trains on the final depth; the others are returned for inspection)
    aux = sum((y * w).sum() for y, w in zip(ys[:4], (1e-4, 2e-4, 3e-4, 4e-4)))
    (loss + aux).backward()
    with torch.no_grad():
        yq = nets.bts_forward(nets.leaf_state(P0), rgb, True, q=nets.bf16_round)
        loss_q = float(L.silog(yq[4], tgt * 10.0, 0.85))
    ref_loss = float(g["train_loss"])
    print("BTS train SILog: reference %.4f, HIP %.4f, bf16-rounding oracle %.4f" % (ref_loss, float(loss), loss_q))
    assert abs(float(loss) - ref_loss) < 2.0 * abs(loss_q - ref_loss) + 1e-2 * ref_loss
    P = nets.leaf_state(P0, requires_grad=True)
    yo = nets.bts_forward(P, rgb, True)
    (L.silog(yo[4], tgt * 10.0, 0.85) + sum((y * w).sum() for y, w in zip(yo[:4], (1e-4, 2e-4, 3e-4, 4e-4)))).backward()
    # the direction floor: what rounding the oracle's own activations to bf16 does to ITS gradients (the trunk's 78 dense
    # layers see 2 x 3 ... 16 x 24 maps of 2 images: batch statistics over 12-768 samples, ReLU masks that flip)
    Pq = nets.leaf_state(P0, requires_grad=True)
    yq2 = nets.bts_forward(Pq, rgb, True, q=nets.bf16_round)
    (L.silog(yq2[4], tgt * 10.0, 0.85) + sum((y * w).sum() for y, w in zip(yq2[:4], (1e-4, 2e-4, 3e-4, 4e-4)))).backward()
    cosf = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    ratios, cosines, floor = [], {}, {}
    for k, p in net.named_parameters():
        go, gh = P[k].grad, p.grad.detach().cpu()
        assert gh.shape == go.shape and torch.isfinite(gh).all(), k
        if float(go.norm()) > 1e-9:
            ratios.append(float(gh.norm() / go.norm()))
            cosines[k], floor[k] = cosf(gh, go), cosf(Pq[k].grad, go)
    ratios = np.array(ratios)
    print("BTS gradient-norm ratios HIP / oracle, percentiles 1 10 50 90 99:", np.percentile(ratios, [1, 10, 50, 90, 99]))
    show = ("decoder.get_depth.0.weight", "decoder.conv1.0.weight", "decoder.reduc1x1.reduc.final.0.weight", "decoder.reduc8x8.reduc.plane_params.weight",
            "decoder.reduc2x2.reduc.inter_64_32.0.weight", "decoder.upconv1.conv.weight", "decoder.daspp_24.atrous_conv.first_bn.weight",
            "decoder.daspp_conv.0.weight", "decoder.conv5.0.weight", "decoder.bn5.weight", "encoder.base_model.denseblock4.denselayer24.conv2.weight",
            "encoder.base_model.denseblock3.denselayer1.norm1.weight", "encoder.base_model.transition1.conv.weight", "encoder.base_model.conv0.weight")
    print("cosines HIP vs fp32 oracle (rounding oracle vs fp32 oracle):", {k: (round(cosines[k], 3), round(floor[k], 3)) for k in show})
    assert np.mean(np.abs(ratios - 1) < 0.25) >= 0.85, np.percentile(ratios, [1, 10, 50, 90, 99])
    for k in show[:6]:
        assert cosines[k] >= 0.9, (k, cosines[k])
    # Trunk directions decorrelate on both sides (rounding oracle: 0.43 at conv0; the HIP path also stores its GRADIENTS in
    # bf16, which the rounding oracle does not: 0.27-0.30).  Asserted here: no tensor far below the rounding oracle's own
    # direction, all positively correlated; the trunk's backward is pinned tightly on a shallow DenseNet below.
    gaps = np.array([floor[k] - cosines[k] for k in cosines])
    print("direction gap (rounding oracle's cosine - HIP's), percentiles 50 90 99 100:", np.round(np.percentile(gaps, [50, 90, 99, 100]), 3))
    # measured (two runs): gap percentiles 50 / 90 / 99 / 100 = 0.09-0.10 / 0.18-0.19 / 0.26-0.27 / 0.33-0.37.  The tail is one or
    # two tensors; the gate below allows 2 % of the tensors past 0.35 and cannot honestly be set tighter than the measured tail.
    worse = [k for k in cosines if cosines[k] < floor[k] - 0.35]
    print("tensors whose direction is > 0.35 below the rounding oracle's: %d of %d" % (len(worse), len(cosines)), worse[:8])
    assert len(worse) <= 0.02 * len(cosines), worse[:20]
    assert np.mean(np.array(list(cosines.values())) > 0.1) >= 0.98
    sd = net.state_dict()
    assert _rel(sd["encoder.base_model.norm5.running_mean"].cpu(), torch.from_numpy(g["rm_norm5"])) < 5e-2
    assert _rel(sd["decoder.bn4_2.running_var"].cpu(), torch.from_numpy(g["rv_bn4_2"])) < 5e-2


def _conditioned(batch=2):
    from mono_depth_estimation_amd.network import Bts
    torch.manual_seed(0)
    net = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=1, encoder_version="densenet161_bts")
    sd = W.bts_conditioned_state(net, 53)
    rgb, tgt = W.synthetic_batch(53, batch, *SIZE)
    P = nets.leaf_state(sd)
    with torch.no_grad():
        nets.bts_forward(P, rgb, True, momentum=1.0)          # running statistics = this batch's (as the golden's generator)
    return net, P, rgb, tgt


def test_bts_conditioned_absrel_within_1e4_of_the_reference(golden):
    """The north-star bound on BTS: on a state where the fp32 reference itself is stable under bf16 storage
    (oracle/weights.bts_conditioned_state: the BatchNorms that read a dense block's concatenation damp the channels the
    block produced, so rounding noise does not compound over the 78 layers, and the coarse levels enter the decoder damped;
    8-image eval batch -- over realisations of the rounding the oracle's own AbsRel moves by +5e-5 +- 1e-5) the HIP path's
    AbsRel of the final depth is within 1e-4 of the REFERENCE's (tests/golden/bts_cond.npz, minted from network/Bts.py +
    metrics.py), log10 likewise, 'rmse' within 1e-4 sqrt(10) (it carries the square root of the depth unit; max_depth 10),
    delta1 within a threshold count's noise, and each of the five outputs within 1.5x the rounding noise."""
    from mono_depth_estimation_amd import metrics
    net, P, rgb, tgt = _conditioned(W.BTS_COND_BATCH)
    g = golden("bts_cond")
    net.load_state_dict({k: v.clone() for k, v in P.items()})
    net = net.cuda().eval()
    with torch.no_grad():
        ys = net(rgb.cuda())
        yo = nets.bts_forward(P, rgb, False)
        yq = nets.bts_forward(P, rgb, False, q=nets.bf16_round)
    for nme, y, o, q in zip(NAMES, ys, yo, yq):
        noise, e_ref = _rel(q, o), _rel(y.cpu()[:2], torch.from_numpy(g["eval_" + nme]))
        print("BTS conditioned eval %-5s: HIP vs reference %.3e; the oracle's rounding noise %.3e" % (nme, e_ref, noise))
        assert _rel(o[:2], torch.from_numpy(g["eval_" + nme])) < 1e-5, nme            # the oracle IS the reference on this state too
        assert e_ref < 1.5 * noise + 1e-3, nme
    names = ["absrel", "rmse", "delta1", "log10"]
    vals = metrics.MetricComputation(names).compute(ys[4], (tgt * 10.0).cuda())
    for n, v in zip(names, vals):
        print("BTS conditioned %-7s reference %.6f HIP %.6f (delta %.2e)" % (n, float(g["eval_" + n]), float(v), abs(float(v) - float(g["eval_" + n]))))
    assert abs(float(vals[0]) - float(g["eval_absrel"])) <= 1e-4
    assert abs(float(vals[1]) - float(g["eval_rmse"])) <= 1e-4 * 10 ** 0.5 and abs(float(vals[3]) - float(g["eval_log10"])) <= 1e-4
    assert abs(float(vals[2]) - float(g["eval_delta1"])) <= 2e-3           # a threshold count over 49 K pixels


def test_bts_loss_curve_agrees_with_the_reference(golden):
    """Convergence parity for the DenseNet network (tests/test_tape_convergence_gpu.py does it for MiDaS,
    tests/test_fcrn_convergence_gpu.py for FCRN): 20 AdamW steps as modules/bts.py:139-152 configures them (eps 1e-3, weight
    decay 1e-2 on the encoder / 0 on the decoder) on one batch from the conditioned state -- the HIP path through its fused
    flat-range step against the curve the REFERENCE's own network/Bts.py traced under torch.optim.AdamW
    (tests/golden/bts_curve.npz, minted by gen_golden.py::gen_bts_curve; round 3 trained the CPU oracle at test time: 129 s).
    The SILog curves stay within 1 % of each other at every step and both fall.  Then the state the HIP path reached -- fp32
    masters that are NOT on the 16-bit grid -- goes to the fp32 CPU oracle: on identical weights the two eval paths agree in
    AbsRel to 1e-4 + what storage rounding of the activations alone does to the oracle (three realisations)."""
    from mono_depth_estimation_amd import criteria, metrics
    net, P0, rgb, tgt = _conditioned()
    g, curve = golden("bts_cond"), golden("bts_curve")
    net.load_state_dict({k: v.detach().clone() for k, v in P0.items()})
    net = net.cuda().train()
    x, t = rgb.cuda(), (tgt * 10.0).cuda()
    crit = criteria.silog_loss(0.85)
    steps, lr = int(curve["steps"]), float(curve["lr"])
    lh = []
    for _ in range(steps):
        net.zero_grad(set_to_none=True)
        loss = crit(net(x)[4], t)
        loss.backward()
        net._store.adam_step(lr, lr, eps=1e-3, weight_decay=(1e-2, 0.0), decoupled=True)
        lh.append(float(loss))
    lh, lo = np.array(lh), curve["silog"]
    assert abs(lo[0] - float(g["train_loss"])) <= 1e-4 * lo[0]                 # both goldens start from the same state
    print("BTS SILog, HIP      :", np.round(lh[[0, 1, 2, 4, 9, 14, 19]], 4))
    print("BTS SILog, reference:", np.round(lo[[0, 1, 2, 4, 9, 14, 19]], 4))
    band = np.abs(lh - lo) / lo
    print("relative gap between the curves: max %.4f mean %.4f; fall HIP %.4f reference %.4f" % (band.max(), band.mean(), lh[-1] / lh[0], lo[-1] / lo[0]))
    assert np.isfinite(lh).all() and lh[-1] < 0.97 * lh[0] and lo[-1] < 0.97 * lo[0]
    # (measured 0.62 % / 0.42 %: the gap opens at the FIRST step and then stays -- AdamW's update is g / (|g| + 1e-3) per element,
    #  nearly a sign; the bf16-stored gradients of the DenseNet trunk flip the sign of elements near zero, so a HIP step
    #  descends a little less than the fp32 reference's.  VNL under SGD, linear in g, tracks its oracle to 0.09 %.)
    assert band.max() < 1e-2 and band.mean() < 6e-3
    # the trained state, on identical weights: eight images (the conditioned fixtures' eval batch), the HIP path's eval forward
    # (two-term weight shadow) against the fp32 oracle holding the SAME fp32 masters
    trained = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    rgb8, tgt8 = W.synthetic_batch(53, W.BTS_COND_BATCH, *SIZE)
    net.eval()
    with torch.no_grad():
        yh = net(rgb8.cuda())[4]
        yo = nets.bts_forward(trained, rgb8, False)[4]
        yqs = [nets.bts_forward(trained, rgb8, False, q=nets.rounding_draw(k))[4] for k in range(3)]
        net._store.split_eval = False
        y1 = net(rgb8.cuda())[4]
        net._store.split_eval = True
    mc = metrics.MetricComputation(["absrel"])
    t8 = (tgt8 * 10.0).cuda()
    a_h, a_1, a_o = (float(mc.compute(y, t8)[0]) for y in (yh, y1, yo.cuda()))
    floor = max(abs(float(mc.compute(y.cuda(), t8)[0]) - a_o) for y in yqs)
    print("trained state, eval AbsRel: HIP %.6f oracle %.6f (delta %.2e; with the one-term weight shadow %.2e); activation rounding alone "
          "moves the oracle by up to %.2e" % (a_h, a_o, abs(a_h - a_o), abs(a_1 - a_o), floor))
    assert abs(a_h - a_o) <= 1e-4 + floor


@pytest.mark.parametrize("version,seed", [("resnet50_bts", 57), ("resnext50_bts", 59), ("resnet101_bts", 63), ("resnext101_bts", 65)])
def test_bts_resnet_encoders_against_oracle_and_reference(version, seed, golden):
    """Bts.py:293-307: the ResNet-50 and the ResNeXt-50 32x4d encoders (the ResNeXt's 3x3 convs as block-diagonal grouped tiles)
    under the same decoder plan.  Eval: the five outputs within 1.5 x the oracle's own bf16-rounding noise (+ 1e-2) of the
    oracle and of the REFERENCE (tests/golden/bts_res*.npz); train: SILog within 2 x the rounding oracle's shift + 1 %;
    gradient norms within 10 % for 98 % of the tensors; `fc` (in the state_dict, never in the forward walk) stays untouched
    by the fused AdamW step."""
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import Bts
    g = golden("bts_" + version[:-4])
    torch.manual_seed(0)
    net = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=1, encoder_version=version)
    sd = W.bts_resnet_fixture_state(net, seed)
    rgb, tgt = W.synthetic_batch(seed, 2, *SIZE)
    P0 = nets.leaf_state(sd)
    with torch.no_grad():
        nets.bts_forward(P0, rgb, True, momentum=1.0)
    net.load_state_dict({k: v.clone() for k, v in P0.items()})
    net = net.cuda().eval()
    with torch.no_grad():
        ys = net(rgb.cuda())
        yo = nets.bts_forward(P0, rgb, False)
        yq = nets.bts_forward(P0, rgb, False, q=nets.bf16_round)
    for nme, y, o, q in zip(NAMES, ys, yo, yq):
        ref = torch.from_numpy(g["eval_" + nme].astype(np.float32))
        noise, e_o, e_ref = _rel(q, o), _rel(y.cpu(), o), _rel(y.cpu(), ref)
        print("BTS %s eval %-5s: HIP vs fp32 oracle %.3e, vs reference %.3e; rounding noise %.3e" % (version, nme, e_o, e_ref, noise))
        assert e_o < 1.5 * noise + 1e-2 and e_ref < 1.5 * noise + 1e-2, nme
    net.train()
    net.zero_grad(set_to_none=True)
    t = (tgt * 10.0).cuda()
    loss = criteria.silog_loss(0.85)(net(rgb.cuda())[4], t)
    loss.backward()
    with torch.no_grad():
        loss_q = float(L.silog(nets.bts_forward(nets.leaf_state(P0), rgb, True, q=nets.bf16_round)[4], tgt * 10.0, 0.85))
    ref_loss = float(g["train_loss"])
    print("BTS %s train SILog: reference %.4f, HIP %.4f, bf16-rounding oracle %.4f" % (version, ref_loss, float(loss), loss_q))
    assert abs(float(loss) - ref_loss) < 2.0 * abs(loss_q - ref_loss) + 1e-2 * ref_loss
    named = dict(net.named_parameters())
    ratios = np.array([float(named[str(k)].grad.norm()) / float(v) for k, v in zip(g["grad_names"], g["grad_norms"]) if v > 1e-8])
    print("gradient-norm ratios HIP / reference, percentiles 1 10 50 90 99:", np.round(np.percentile(ratios, [1, 10, 50, 90, 99]), 3))
    # measured (resnet50 twice, resnext50 once): 1st ... 99th percentile 0.965 ... 1.034 -- a residual trunk with damped branch
    # BatchNorms does not decorrelate the way the DenseNet trunk does
    assert np.mean(np.abs(ratios - 1) < 0.10) >= 0.98, np.round(np.percentile(ratios, [1, 10, 50, 90, 99]), 3)
    fc = named["encoder.base_model.fc.weight"]
    before = fc.detach().clone()
    assert fc.grad is None
    net._store.adam_step(1e-4, 1e-4, eps=1e-3, weight_decay=(1e-2, 0.0), decoupled=True)
    assert torch.equal(named["encoder.base_model.fc.weight"].detach(), before)
    assert not torch.equal(named["encoder.base_model.conv1.weight"].detach().cpu(), P0["encoder.base_model.conv1.weight"])


def test_image_residual_kernel_against_autograd():
    """mde_image_residual_fwd / _bwd against torch autograd on the same fp32 maps: clamp(2 d - 1 + image) on the six colour
    channels, the mean of the image on the two alpha channels, the last two channels untouched; the gradient is 2 where the
    clamp is open, 0 where it is shut, 1 on the depth channels."""
    from mono_depth_estimation_amd import ops
    gen = torch.Generator().manual_seed(5)
    d = torch.rand(3, 10, 20, 28, generator=gen).cuda()
    rgb = torch.rand(3, 3, 20, 28, generator=gen).cuda()
    dout = torch.randn(3, 10, 20, 28, generator=gen).cuda()
    y, dd = torch.empty_like(d), torch.empty_like(d)
    ops.image_residual_fwd(d, rgb, y)
    ops.image_residual_bwd(dout, d, rgb, dd)
    dr = d.clone().requires_grad_(True)
    mean = rgb.mean(dim=1)
    ref = torch.cat([torch.clamp(dr[:, :3] * 2 - 1 + rgb, 0, 1), torch.clamp(dr[:, 3] * 2 - 1 + mean, 0, 1).unsqueeze(1),
                     torch.clamp(dr[:, 4:7] * 2 - 1 + rgb, 0, 1), torch.clamp(dr[:, 7] * 2 - 1 + mean, 0, 1).unsqueeze(1), dr[:, 8:]], 1)
    ref.backward(dout)
    share = float(((ref[:, :8] <= 0) | (ref[:, :8] >= 1)).float().mean())
    assert 0.2 < share < 0.8, share                                  # both sides of the clamp are exercised
    assert (y - ref.detach()).abs().max() < 1e-6
    # values within one ulp of a clamp edge may fall on either side of it (the mean of three floats is summed in another order)
    edge = ((ref.detach() - 0).abs() < 1e-6) | ((ref.detach() - 1).abs() < 1e-6)
    assert torch.equal(torch.where(edge, dr.grad, dd), dr.grad) and float(edge.float().mean()) < 0.9


def test_bts_image_residuals_against_oracle_and_reference(golden):
    """BtsModel(out_channels=10, image_residuals=True) (Bts.py:264-271) through the HIP plan: the ten-channel output against the
    fp32 oracle and the REFERENCE's (tests/golden/bts_imgres.npz) within the rounding oracle's noise, mean |final - target| in
    train mode, and the gradient norms of every tensor against the reference's."""
    from mono_depth_estimation_amd.network import Bts
    g = golden("bts_imgres")
    torch.manual_seed(0)
    net = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=10, image_residuals=True, encoder_version="densenet161_bts")
    P0 = nets.leaf_state(W.bts_conditioned_state(net, 61))
    rgb, _ = W.synthetic_batch(61, 2, *SIZE)
    target = W.uniform(61, "layers", (2, 10, *SIZE), 0.0, 1.0)
    with torch.no_grad():
        nets.bts_forward(P0, rgb, True, momentum=1.0, image_residuals=True)
    net.load_state_dict({k: v.clone() for k, v in P0.items()})
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(rgb.cuda())[4].cpu()
        o = nets.bts_forward(P0, rgb, False, image_residuals=True)[4]
        q = nets.bts_forward(P0, rgb, False, q=nets.bf16_round, image_residuals=True)[4]
    ref = torch.from_numpy(g["eval_final"].astype(np.float32))
    assert y.shape == (2, 10, *SIZE) and float(y[:, :8].min()) >= 0.0 and float(y[:, :8].max()) <= 1.0
    noise, e_o, e_ref = _rel(q, o), _rel(y, o), _rel(y, ref)
    print("BTS image residuals, eval: HIP vs fp32 oracle %.3e, vs reference %.3e; rounding noise %.3e" % (e_o, e_ref, noise))
    assert e_o < 1.5 * noise + 2e-3 and e_ref < 1.5 * noise + 2e-3
    net.train()
    net.zero_grad(set_to_none=True)
    loss = (net(rgb.cuda())[4] - target.cuda()).abs().mean()
    loss.backward()
    loss = loss.detach()
    with torch.no_grad():
        loss_q = float((nets.bts_forward(nets.leaf_state(P0), rgb, True, q=nets.bf16_round, image_residuals=True)[4] - target).abs().mean())
    ref_loss = float(g["train_loss"])
    print("BTS image residuals, train mean |final - target|: reference %.5f, HIP %.5f, bf16-rounding oracle %.5f" % (ref_loss, float(loss), loss_q))
    assert abs(float(loss) - ref_loss) < 2.0 * abs(loss_q - ref_loss) + 2e-3 * ref_loss
    named = dict(net.named_parameters())
    ratios = np.array([float(named[str(k)].grad.norm()) / float(v) for k, v in zip(g["grad_names"], g["grad_norms"]) if v > 1e-8])
    pct = np.round(np.percentile(ratios, [1, 10, 50, 90, 99]), 3)
    print("gradient-norm ratios HIP / reference, percentiles 1 10 50 90 99:", pct)
    assert np.mean(np.abs(ratios - 1) < 0.15) >= 0.95, pct


def test_shallow_densenet_trunk_gradients():
    """The DenseNet machinery (7x7/2 image stem on the GEMM kernel, max-pool, dense layers writing into the block's
    concatenation, batch moments reduced once per channel group and shared by every later BatchNorm, transition with 2x2
    average pool, final norm + ReLU) on a SHALLOW trunk (3 + 2 dense layers, growth 16) with 24 x 32 .. 6 x 8 maps, where
    storage rounding is not amplified: outputs within 3e-2, every parameter gradient within 8 % in norm and cosine >= 0.95 (measured 0.958-0.9998)."""
    import torch.nn as nn
    from mono_depth_estimation_amd import graph as G
    from mono_depth_estimation_amd.network import Bts

    class Engine(Bts.BtsEngine):
        def _plan(self):
            skips = self._dense_trunk(self.m.feats, self.N, self.H, self.W)
            c = self.add(G.Conv(self, skips[-1], self.m.head.weight, 1)).out
            self.heads = [self.add(G.ToNCHW(self, c, None, 8, "none"))]

    class Mini(G.TapeModule):
        _engine_cls = Engine

        def __init__(self):
            super().__init__()
            self.feats = Bts._densenet_features(16, (3, 2), 32)
            self.head = nn.Conv2d(72, 8, 1, bias=False)
            self._init_runtime()

        def _make_store(self, device):
            return G.NetStore(self, device, is_encoder=lambda n: n.startswith("feats."))

        def forward(self, x):
            return self._run(x)[0]

    torch.manual_seed(0)
    net = Mini()
    sd = W.net_conditioned_state(net, 61)
    rgb, _ = W.synthetic_batch(61, 4, 48, 64)
    dy = W.normal(61, "dy", (4, 8, 6, 8))

    def oracle(P, train):
        n = nets.Net(P, train)
        f = nets.densenet_features(n, rgb, "feats.", blocks=(3, 2))
        return torch.nn.functional.conv2d(torch.relu(f[-1]), P["head.weight"])
    P = nets.leaf_state(sd, requires_grad=True)
    yo = oracle(P, True)
    (yo * dy).sum().backward()
    net = net.cuda().train()
    net._store.set_deterministic(True)                    # order-independent sums: reproducible figures
    try:
        y = net(rgb.cuda())
        (y * dy.cuda()).sum().backward()
    finally:
        net._store.set_deterministic(False)
    assert _rel(y.detach().cpu(), yo.detach()) < 3e-2, _rel(y.detach().cpu(), yo.detach())
    # The bound on every gradient norm is MEASURED: the fp32 oracle with activations and activation gradients rounded to the
    # storage type (tests/rounding.py), six realisations of the rounding; the HIP path's ratio to the fp32 oracle must stay
    # within twice the rounding oracle's largest excursion from 1 (+ 2 %).  (Round 3 gave feats.norm0.bias a flat 15 % after its ratio went
    # 0.96 -> 0.917 with the convs' K order: the rounding oracle puts that tensor at 0.989 +- 0.034, range 0.94 ... 1.05 --
    # a shift in front of ReLU -> max-pool -> BatchNorm is a cancellation residue, and both figures are draws from that.)
    import rounding as R

    def run(k):
        Pk = nets.leaf_state(sd, requires_grad=True)
        n = nets.Net(Pk, True, q=R.q_both(k)) if k is not None else nets.Net(Pk, True)
        f = nets.densenet_features(n, rgb, "feats.", blocks=(3, 2))
        (torch.nn.functional.conv2d(torch.relu(f[-1]), Pk["head.weight"]) * dy).sum().backward()
        return {k_: float(v.grad.norm()) for k_, v in Pk.items() if v.grad is not None}
    base, noise = R.grad_norm_noise(run, draws=6)
    stats = {}
    for k, p in net.named_parameters():
        go, gh = P[k].grad, p.grad.detach().cpu()
        stats[k] = (float(gh.norm() / go.norm()), float((gh * go).sum() / (gh.norm() * go.norm() + 1e-30)))
    print("shallow densenet: output rel %.3e; worst norm ratio %s; worst cosine %s" % (
        _rel(y.detach().cpu(), yo.detach()), max(stats.items(), key=lambda kv: abs(kv[1][0] - 1)), min(stats.items(), key=lambda kv: kv[1][1])))
    print("the rounding oracle's largest excursions:", sorted(((round(v, 3), k) for k, v in noise.items()), reverse=True)[:4])
    for k, (ratio, cos) in stats.items():
        if k == "feats.norm0.weight":
            continue      # a per-channel scale right in front of another BatchNorm (ReLU and max-pool commute with it): its
                          # true gradient is a cancellation residue (the rounding oracle itself: ratio 1.6 ... 3.1)
        assert abs(ratio - 1) <= 2.0 * noise[k] + 2e-2 and cos >= 0.95, (k, ratio, cos, noise[k])
    for k in ("feats.norm5.running_mean", "feats.denseblock1.denselayer3.norm1.running_var", "feats.transition1.norm.running_mean"):
        assert _rel(net.state_dict()[k].cpu(), P[k]) < 1e-2, k


def test_bts_adamw_steps_reduce_the_loss(setup):
    """modules/bts.py:139-152: AdamW eps 1e-3, weight decay 1e-2 on the encoder / 0 on the decoder, through the fused step."""
    from mono_depth_estimation_amd import criteria
    net, _, rgb, tgt = setup
    crit = criteria.silog_loss(0.85)
    x, t = rgb.cuda(), (tgt * 10.0).cuda()
    net.train()
    losses = []
    for _ in range(4):
        net.zero_grad(set_to_none=True)
        loss = crit(net(x)[4], t)
        loss.backward()
        net._store.adam_step(1e-4, 1e-4, eps=1e-3, weight_decay=(1e-2, 0.0), decoupled=True)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_config3_bts_16x3x480x640_with_silog():
    """BTS DenseNet-161, bts_size 512, max_depth 1.0, 16 images at 480 x 640, SILog on the final depth."""
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import Bts
    torch.manual_seed(4)
    net = Bts.BtsModel(bts_size=512, max_depth=1.0, out_channels=1, encoder_version="densenet161_bts").cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if k.endswith("plane_params.weight") or k.endswith("final.0.weight") or k == "decoder.get_depth.0.weight":
                p.mul_(0.05)
    g = torch.Generator(device="cuda")
    g.manual_seed(8)
    x = torch.rand(16, 3, 480, 640, generator=g, device="cuda")
    gt = (0.05 + 0.95 * torch.rand(16, 1, 480, 640, generator=g, device="cuda")).masked_fill(torch.rand(16, 1, 480, 640, generator=g, device="cuda") < 0.1, 0.0)
    net.eval()
    with torch.no_grad():
        ys = net(x)
        assert len(ys) == 5 and all(y.shape == (16, 1, 480, 640) and torch.isfinite(y).all() for y in ys)
        assert float(ys[4].min()) >= 0 and float(ys[4].max()) <= 1.0
        y2 = net(x[2:4].contiguous())
        assert float((y2[4] - ys[4][2:4]).abs().max()) <= 5e-2
    del ys, y2
    crit = criteria.silog_loss(0.85)
    net.train()
    losses = []
    for it in range(3):
        net.zero_grad(set_to_none=True)
        loss = crit(net(x)[4], gt)
        loss.backward()
        if it == 0:
            for k, p in net.named_parameters():
                assert p.grad is not None and torch.isfinite(p.grad).all(), k
        net._store.adam_step(1e-4, 1e-4, eps=1e-3, weight_decay=(1e-2, 0.0), decoupled=True)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_bts_loss_on_an_auxiliary_output_only():
    """A loss that supervises lpg8x8 alone (reference Bts.py returns the five maps; a caller may weight any of them): the final
    depth's head then receives no gradient at all -- `get_depth` gets none, everything the 1/8 map depends on gets finite ones."""
    from mono_depth_estimation_amd.network import Bts
    torch.manual_seed(3)
    net = Bts.BtsModel(bts_size=512, max_depth=10, out_channels=1, encoder_version="densenet121_bts").cuda().train()
    x = torch.rand(2, 3, 64, 96, device="cuda")
    ys = net(x)
    ys[0].mean().backward()
    named = dict(net.named_parameters())
    g_final = named["decoder.get_depth.0.weight"].grad
    assert g_final is None or float(g_final.abs().max()) == 0.0
    g8 = named["decoder.reduc8x8.reduc.plane_params.weight"].grad
    assert g8 is not None and torch.isfinite(g8).all() and float(g8.abs().max()) > 0
    g0 = named["encoder.base_model.conv0.weight"].grad
    assert g0 is not None and torch.isfinite(g0).all() and float(g0.abs().max()) > 0
