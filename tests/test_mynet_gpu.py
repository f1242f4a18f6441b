"""GPU parity of the HIP MyNet path: the new kernels (weighter tail, branch combination, transposed conv and
PixelShuffle tape ops) against plain torch fp32 on the same bf16-rounded operands, and the network
(mono_depth_estimation_amd.network.MyNet.MyModel) end to end against the CPU oracle (oracle/nets.py: mynet_forward, pinned to the
reference's own network/MyNet.py by tests/golden/mynet.npz) and the golden vectors, plus the module's default geometry
(384 x 384, modules/my.py:27-28) as a property test.

Tolerances relative to what rounding the ORACLE's own activations to bf16 does (`noise`): eval output within 1.5 noise + 3e-3
of the oracle and the reference; MidasLoss(0.5, 'mse') within 1 %; gradient norms within 25 % for 85 % of the tensors;
direction: decoder tensors >= 0.9 or within 0.1 of the rounding oracle's own cosine, no tensor more than 0.35 below it (the
DenseNet trunk at random weights decorrelates under rounding on both sides, see tests/test_bts_net_gpu.py)."""
import numpy as np
import pytest
import torch

from mono_depth_estimation_amd.ops import ACT_DTYPE as ACT        # the library's 16-bit storage type (bf16; fp16 under MDE_ACT_DTYPE=fp16)
import torch.nn.functional as F

from oracle import losses as L
from oracle import nets
from oracle import weights as W

pytestmark = pytest.mark.gpu
SIZE = (64, 96)


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def _bf(t):
    return t.to(ACT).to(torch.float32)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(ACT).cuda()


# ---------------------------------------------------------------------------------------------- kernels
def test_weighted_pool_against_torch():
    from mono_depth_estimation_amd import ops
    torch.manual_seed(3)
    N, C, H, W_ = 3, 32, 8, 12
    a = _bf(torch.randn(N, C, H, W_)).requires_grad_(True)
    w = (torch.randn(1, H * W_) * 0.05).requires_grad_(True)
    b = torch.tensor([0.1], requires_grad=True)
    s = torch.sigmoid(torch.sum(F.linear(a.flatten(2), w, b), dim=1))          # MyNet.py:104-117 -> [N][1]
    ah = _nhwc(a.detach())
    pre, sc = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ops.weighted_pool_fwd(ah, C, w.detach().cuda(), b.detach().cuda(), pre, sc, N, H * W_, C)
    assert (sc.cpu() - s.detach()[:, 0]).abs().max() < 1e-5
    ds = torch.randn(N, 1)
    s.backward(ds)
    da = torch.empty_like(ah)
    dw, db = torch.zeros(H * W_, device="cuda"), torch.zeros(1, device="cuda")
    ops.weighted_pool_bwd(ds[:, 0].contiguous().cuda(), sc, ah, C, w.detach().cuda(), da, C, False, dw, db, N, H * W_, C)
    assert (da.float().cpu().permute(0, 3, 1, 2) - a.grad).abs().max() < 2 ** -7 * a.grad.abs().max()
    assert (dw.cpu() - w.grad[0]).abs().max() < 1e-4 * w.grad.abs().max() + 1e-6
    assert abs(float(db) - float(b.grad)) < 1e-4 * abs(float(b.grad)) + 1e-6
    da2 = ah.clone()
    ops.weighted_pool_bwd(ds[:, 0].contiguous().cuda(), sc, ah, C, w.detach().cuda(), da2, C, True, dw, db, N, H * W_, C)
    assert (da2.float().cpu().permute(0, 3, 1, 2) - (a.grad + a.detach())).abs().max() < 2 ** -6 * (a.grad + a.detach()).abs().max()
    assert (dw.cpu() - 2 * w.grad[0]).abs().max() < 2e-4 * w.grad.abs().max() + 1e-6       # (the weight is shared: gradients add)


def test_combine3_against_torch():
    from mono_depth_estimation_amd import ops
    torch.manual_seed(5)
    N, HW = 3, 1000
    maps = [torch.rand(N, HW).requires_grad_(True) for _ in range(3)]
    scales = [torch.rand(N).requires_grad_(True) for _ in range(3)]
    out = sum(m * s[:, None] for m, s in zip(maps, scales)) / 3.0 * 10.0
    mh, sh = [m.detach().cuda() for m in maps], [s.detach().cuda() for s in scales]
    oh = torch.empty(N, HW, device="cuda")
    ops.combine3_fwd(mh, sh, 10.0 / 3.0, N, HW, oh)
    assert (oh.cpu() - out.detach()).abs().max() < 1e-5
    g = torch.randn(N, HW)
    out.backward(g)
    dm = [torch.ones(N, HW, device="cuda") for _ in range(3)]               # the kernel ADDS into the maps' gradients
    ds = torch.full((3, N), 7.0, device="cuda")
    ops.combine3_bwd(g.cuda(), mh, sh, 10.0 / 3.0, N, HW, dm, ds)
    for k in range(3):
        assert (dm[k].cpu() - 1.0 - maps[k].grad).abs().max() < 1e-5
        assert (ds[k].cpu() - scales[k].grad).abs().max() < 1e-4 * scales[k].grad.abs().max()


# ---------------------------------------------------------------------------------------------- the network
@pytest.fixture(scope="module")
def setup():
    from mono_depth_estimation_amd.network import MyNet
    torch.manual_seed(0)
    net = MyNet.MyModel(input_size=SIZE, encoder_version="densenet161_bts")
    sd = W.mynet_fixture_state(net, 71)
    rgb, tgt = W.synthetic_batch(71, 2, *SIZE)
    P = nets.leaf_state(sd)
    with torch.no_grad():
        nets.mynet_forward(P, rgb, True, momentum=1.0)
    net.load_state_dict({k: v.clone() for k, v in P.items()})
    return net.cuda(), P, rgb, tgt


def test_mynet_eval_against_oracle_and_reference(setup, golden):
    net, P, rgb, tgt = setup
    g = golden("mynet")
    net.eval()
    with torch.no_grad():
        y = net(rgb.cuda())
        yo = nets.mynet_forward(P, rgb, False)
        yq = nets.mynet_forward(P, rgb, False, q=nets.bf16_round)
    assert y.shape == (2, 1, *SIZE) and y.dtype == torch.float32
    ref = torch.from_numpy(g["eval_out"])
    noise, e_o, e_q, e_ref = _rel(yq, yo), _rel(y.cpu(), yo), _rel(y.cpu(), yq), _rel(y.cpu(), ref)
    print("MyNet eval: HIP vs fp32 oracle %.3e, vs bf16-rounding oracle %.3e, vs reference %.3e; rounding noise %.3e" % (e_o, e_q, e_ref, noise))
    assert noise < 3e-2 and e_o < 1.5 * noise + 3e-3 and e_ref < 1.5 * noise + 3e-3 and e_q < 1.2 * noise + 3e-3
    t = tgt * 10.0
    absrel = lambda d: float(((d - t).abs() / t)[t > 0].mean())
    print("MyNet eval AbsRel: reference %.5f, HIP %.5f, rounding oracle %.5f" % (absrel(ref), absrel(y.cpu()), absrel(yq)))
    assert abs(absrel(y.cpu()) - absrel(ref)) < 2e-3 + 1.5 * abs(absrel(yq) - absrel(ref))


def test_mynet_train_step_against_oracle_and_reference(setup, golden):
    from mono_depth_estimation_amd import criteria
    net, P0, rgb, tgt = setup
    g = golden("mynet")
    net.train()
    net.zero_grad(set_to_none=True)
    y = net(rgb.cuda())
    loss = criteria.MidasLoss(alpha=0.5, loss="mse", reduction="batch-based")(y, (tgt * 10.0).cuda())
    loss.backward()
    ref_loss = float(g["train_loss"])
    print("MyNet train MidasLoss: reference %.5f, HIP %.5f" % (ref_loss, float(loss)))
    assert abs(float(loss) - ref_loss) < 1e-2 * ref_loss
    P = nets.leaf_state(P0, requires_grad=True)
    L.midas_loss(nets.mynet_forward(P, rgb, True), tgt * 10.0, alpha=0.5, loss="mse").backward()
    Pq = nets.leaf_state(P0, requires_grad=True)
    L.midas_loss(nets.mynet_forward(Pq, rgb, True, q=nets.bf16_round), tgt * 10.0, alpha=0.5, loss="mse").backward()
    cosf = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    ratios, cosines, floor = [], {}, {}
    for k, p in net.named_parameters():
        go = P[k].grad
        if go is None:                                  # resConfUnit1 of every refine block: unused by the forward (MyNet.py:137-140)
            assert ".resConfUnit1." in k and (p.grad is None or float(p.grad.abs().max()) == 0.0), k
            continue
        gh = p.grad.detach().cpu()
        assert gh.shape == go.shape and torch.isfinite(gh).all(), k
        if float(go.norm()) > 1e-9:
            ratios.append(float(gh.norm() / go.norm()))
            cosines[k], floor[k] = cosf(gh, go), cosf(Pq[k].grad, go)
    ratios = np.array(ratios)
    cs, fl = np.array(list(cosines.values())), np.array(list(floor.values()))
    print("MyNet gradient-norm ratios HIP / oracle, percentiles 1 10 50 90 99:", np.percentile(ratios, [1, 10, 50, 90, 99]))
    print("MyNet gradient cosines vs the fp32 oracle, percentiles 1 10 50: HIP %s, bf16-rounding oracle %s" % (
        np.percentile(cs, [1, 10, 50]).round(3), np.percentile(fl, [1, 10, 50]).round(3)))
    # The DenseNet trunk at random weights is chaotic under rounding for the ORACLE as well (tests/test_bts_net_gpu.py, where the
    # same trunk plan is pinned tightly on a shallow DenseNet): norms and directions are asserted against the rounding oracle's.
    assert np.mean(np.abs(ratios - 1) < 0.25) >= 0.85, np.percentile(ratios, [1, 10, 50, 90, 99])
    worse = [k for k in cosines if cosines[k] < floor[k] - 0.35]
    assert len(worse) <= 0.02 * len(cosines), worse[:10]
    assert np.mean(cs > 0.1) >= 0.98
    d = "decoder."
    print("decoder cosines (HIP, rounding oracle):", {k[len(d):]: (round(cosines[k], 3), round(floor[k], 3)) for k in cosines
                                                       if k.startswith(d) and (k.endswith("conv.weight") or "tconv" in k or "mlp" in k or "get_depth" in k)})
    for k in (d + "get_depth.1.weight", d + "weighter.mlp.weight", d + "weighter.mlp.bias", d + "weighter.conv.conv.weight", d + "weighter.conv.bn.weight",
              d + "sharpness.tconv0.weight", d + "sharpness.tconv2.weight", d + "sharpness.tconv1.bias", d + "sharpness.up0.1.weight",
              d + "details.down.conv.weight", d + "details.conv_final.conv.weight", d + "global_con.conv.conv.weight", d + "global_con.conv.bn.bias",
              d + "refine0.resConfUnit2.conv1.weight", d + "refine3.resConfUnit2.conv2.bias"):
        assert cosines[k] >= min(0.9, floor[k] - 0.1), (k, cosines[k], floor[k])
    assert _rel(net.state_dict()[d + "weighter.conv.bn.running_var"].cpu(), torch.from_numpy(g["rv_weighter"])) < 2e-2


def test_mynet_adam_steps_reduce_the_loss(setup):
    """modules/my.py:66-70: Adam, encoder at 1x, decoder at 10x; through the fused flat-range step."""
    from mono_depth_estimation_amd import criteria
    net, _, rgb, tgt = setup
    crit = criteria.MidasLoss(alpha=0.5, loss="mse", reduction="batch-based")
    x, t = rgb.cuda(), (tgt * 10.0).cuda()
    net.train()
    losses = []
    for _ in range(4):
        net.zero_grad(set_to_none=True)
        loss = crit(net(x), t)
        loss.backward()
        net._store.adam_step(1e-5, 1e-4)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_mynet_default_geometry_384x384():
    """The module's defaults (modules/my.py:27-36,160: 384 x 384, batch 16 — 4 here): one training step checked through
    properties; a different image size is refused by name (the reference fails inside its Linear layer)."""
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import MyNet
    torch.manual_seed(0)
    net = MyNet.MyModel().cuda().train()
    rgb, tgt = W.synthetic_batch(73, 4, 384, 384)
    y = net(rgb.cuda())
    assert y.shape == (4, 1, 384, 384) and float(y.min()) >= 0.0 and float(y.max()) <= 10.0
    loss = criteria.MidasLoss(alpha=0.5, loss="mse", reduction="batch-based")(y, (tgt * 10.0).cuda())
    loss.backward()
    assert np.isfinite(float(loss))
    for k, p in net.named_parameters():
        if ".resConfUnit1." in k:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert float(net.decoder.weighter.mlp.weight.grad.abs().max()) > 0 and float(net.encoder.base_model.conv0.weight.grad.abs().max()) > 0
    net._store.adam_step(1e-4, 1e-3)
    with pytest.raises(NotImplementedError, match="built for"):
        net(torch.rand(1, 3, 352, 384, device="cuda"))
