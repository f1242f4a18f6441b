"""GPU parity of the HIP DORN path: the new kernels (csrc/ordinal.hip, ceil-mode max-pool) against plain
torch fp32 on the same bf16-rounded operands, criteria.ordLoss against the reference's golden values, and the network
(mono_depth_estimation_amd.network.Dorn.DORN) end to end against the CPU oracle (oracle/nets.py: dorn_forward, pinned to the
reference's own network/Dorn.py by tests/golden/dorn_net.npz) and the golden vectors.

The three nn.Dropout2d of the scene module draw their masks with torch's DEVICE generator in the HIP path (the reference's CPU
draw cannot be reproduced on a GPU, nor can torch's own CUDA draw be on a CPU): the train-mode comparisons read the masks the
HIP run drew and hand them to the oracle.

Tolerances relative to what rounding the ORACLE's own activations to bf16 does (`noise`): ordinal probabilities within
1.5 noise + 2e-3 (relative L2); decoded labels differ on at most the fraction of pixels the rounding oracle itself flips, +1 %;
ordLoss within 1 %; gradient norms within 15 % for 90 % of the tensors; direction cosine >= 0.95 in the scene module, >= 0.85
in the trunk, or within 0.03 / 0.05 of the rounding oracle's own cosine where that is lower (ReLU masks flip under rounding)."""
import types

import numpy as np
import pytest
import torch

from mono_depth_estimation_amd.ops import ACT_DTYPE as ACT        # the library's 16-bit storage type (bf16; fp16 under MDE_ACT_DTYPE=fp16)
import torch.nn.functional as F

from oracle import losses as L
from oracle import nets
from oracle import weights as W

pytestmark = pytest.mark.gpu

DORN_ARGS = dict(input_size=(65, 81), kernel_size=4, ord_num=12, alpha=0.02, beta=10.0, discretization="SID", pretrained=0,
                 pyramid=[2, 3, 4], batch_norm=0, dropout=0.5)
DORN_KW = dict(size=(65, 81), kernel_size=4, pyramid=(2, 3, 4), dropout=0.5)


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def _bf(t):
    return t.to(ACT).to(torch.float32)


# ---------------------------------------------------------------------------------------------- kernels
def test_chan_scale_and_flat_avgpool_against_torch():
    from mono_depth_estimation_amd import ops
    torch.manual_seed(3)
    N, H, W_, C, k = 3, 9, 11, 64, 4
    x = _bf(torch.randn(N, C, H, W_))
    m = (torch.rand(N, C) > 0.5).float() * 2.0
    xh = x.permute(0, 2, 3, 1).contiguous().to(ACT).cuda()
    out = torch.empty_like(xh)
    ops.chan_scale(xh, C, m.cuda(), out, C, N, H * W_, C)
    want = x * m.view(N, C, 1, 1)
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), _bf(want))
    acc = xh.clone()
    ops.chan_scale(xh, C, m.cuda(), acc, C, N, H * W_, C, accumulate=True)
    assert torch.equal(acc.float().cpu().permute(0, 3, 1, 2), _bf(x + want))
    # AvgPool2d(k, k, k // 2) * mask -> the NCHW flatten
    xr = x.clone().requires_grad_(True)
    pooled = F.avg_pool2d(xr, k, k, k // 2) * m.view(N, C, 1, 1)
    oh, ow = pooled.shape[2:]
    flat = torch.empty(N, C * oh * ow, dtype=ACT, device="cuda")
    ops.avgpool_flat_fwd(xh, C, m.cuda(), flat, N, H, W_, C, k, k, k // 2)
    assert (flat.float().cpu() - pooled.detach().reshape(N, -1)).abs().max() < 2e-2 * pooled.abs().max()
    g = _bf(torch.randn(N, C * oh * ow))
    pooled.reshape(N, -1).backward(g)
    dx = torch.empty_like(xh)
    ops.avgpool_flat_bwd(g.to(ACT).cuda(), m.cuda(), dx, C, N, H, W_, C, k, k, k // 2)
    assert (dx.float().cpu().permute(0, 3, 1, 2) - xr.grad).abs().max() < 1e-2 * xr.grad.abs().max()
    dx2 = xh.clone()
    ops.avgpool_flat_bwd(g.to(ACT).cuda(), m.cuda(), dx2, C, N, H, W_, C, k, k, k // 2, accumulate=True)
    assert (dx2.float().cpu().permute(0, 3, 1, 2) - (xr.grad + x)).abs().max() < 2e-2 * (xr.grad + x).abs().max()
    # overlapping windows (stride < kernel) exercise the general gather bounds of the backward kernel
    xr = x.clone().requires_grad_(True)
    pooled = F.avg_pool2d(xr, 4, 2, 1)
    oh, ow = pooled.shape[2:]
    flat = torch.empty(N, C * oh * ow, dtype=ACT, device="cuda")
    ops.avgpool_flat_fwd(xh, C, None, flat, N, H, W_, C, 4, 2, 1)
    assert (flat.float().cpu() - pooled.detach().reshape(N, -1)).abs().max() < 2e-2 * pooled.abs().max()
    g = _bf(torch.randn(N, C * oh * ow))
    pooled.reshape(N, -1).backward(g)
    ops.avgpool_flat_bwd(g.to(ACT).cuda(), None, dx, C, N, H, W_, C, 4, 2, 1)
    assert (dx.float().cpu().permute(0, 3, 1, 2) - xr.grad).abs().max() < 1e-2 * xr.grad.abs().max()


@pytest.mark.parametrize("hw", [(9, 12), (8, 13), (33, 45)])
def test_maxpool_ceil_mode_against_torch(hw):
    from mono_depth_estimation_amd import ops
    torch.manual_seed(5)
    N, C = 2, 16
    H, W_ = hw
    x = _bf(torch.randn(N, C, H, W_)).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1, ceil_mode=True)
    OH, OW = ops.maxpool_out_size(H, True), ops.maxpool_out_size(W_, True)
    assert (OH, OW) == tuple(y.shape[2:])
    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(ACT).cuda()
    out = torch.empty(N, OH, OW, C, dtype=ACT, device="cuda")
    idx = torch.empty(N, OH, OW, C, dtype=torch.uint8, device="cuda")
    ops.maxpool_fwd(xh, out, idx, N, H, W_, C, True)
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), y.detach())
    g = _bf(torch.randn_like(y))
    y.backward(g)
    dx = torch.empty_like(xh)
    ops.maxpool_bwd(g.permute(0, 2, 3, 1).contiguous().to(ACT).cuda(), idx, dx, N, H, W_, C, True)
    assert (dx.float().cpu().permute(0, 3, 1, 2) - x.grad).abs().max() <= 2 ** -7 * x.grad.abs().max()


@pytest.mark.parametrize("K,HW,ld", [(12, 65 * 81, 24), (68, 1000, 136), (71, 333, 144)])
def test_ordinal_head_against_torch(K, HW, ld):
    from mono_depth_estimation_amd import ops
    torch.manual_seed(7)
    N = 2
    x = _bf(torch.randn(N, 2 * K, HW, 1) * 2.0)
    x[0, 0, :7, 0] = 2e4                      # beyond the upper clamp
    x[0, 1, :7, 0] = 3e4
    xr = x.clone().requires_grad_(True)
    label, prob = nets.ordinal_layer(xr)
    xh = torch.zeros(N, HW, ld, dtype=ACT, device="cuda")
    xh[..., :2 * K] = x[..., 0].permute(0, 2, 1).to(ACT).cuda()
    ph = torch.empty(N, K, HW, device="cuda")
    lh = torch.empty(N, HW, dtype=torch.int64, device="cuda")
    ops.ordinal_fwd(xh, ld, ph, lh, N, HW, K)
    assert (ph.cpu() - prob.detach().reshape(N, K, HW)).abs().max() < 2e-6
    assert (lh.cpu() != label.reshape(N, HW)).float().mean() < 1e-3
    g = torch.randn(N, K, HW)
    prob.reshape(N, K, HW).backward(g)
    dx = torch.full((N, HW, ld), 7.0, dtype=ACT, device="cuda")
    ops.ordinal_bwd(g.cuda(), xh, ld, dx, ld, N, HW, K)
    want = xr.grad[..., 0].permute(0, 2, 1)
    got = dx.float().cpu()
    assert (got[..., :2 * K] - want).abs().max() <= 2 ** -8 * want.abs().max() + 1e-6
    if ld > 2 * K:
        assert float(got[..., 2 * K:].abs().max()) == 0.0
    assert float(want[0, :7, :2].abs().max()) == 0.0         # (the clamp's zero gradient is exercised)


def test_ord_loss_against_the_reference_and_the_oracle(golden):
    from mono_depth_estimation_amd import criteria
    g = golden("dorn_net")
    prob = torch.tensor(g["ord_prob"]).cuda().requires_grad_(True)
    loss = criteria.ordLoss()(prob, torch.tensor(g["ord_target"]).cuda())
    loss.backward()
    assert abs(float(loss) - float(g["ord_loss"])) < 2e-5 * abs(float(g["ord_loss"]))
    assert (prob.grad.cpu() - torch.tensor(g["ord_grad"])).abs().max() < 1e-6 + 1e-5 * np.abs(g["ord_grad"]).max()
    # DORN's default geometry: 68 planes at 257 x 353, the label map modules/dorn.py computes from a depth map
    torch.manual_seed(11)
    p = torch.rand(2, 68, 257, 353)
    depth = torch.rand(2, 1, 257, 353) * 9.0 + 0.05
    depth[0, 0, :3] = 0.0                       # holes: log(0) = -inf
    t = L.sid_labels(depth, 0.02, 10.0, 68)
    pr = p.clone().requires_grad_(True)
    want = L.ord_loss(pr, t)
    want.backward()
    ph = p.cuda().requires_grad_(True)
    got = criteria.ordLoss()(ph, t.cuda())
    (got * 3.0).backward()
    assert abs(float(got) - float(want)) < 1e-5 * abs(float(want))
    assert (ph.grad.cpu() - 3.0 * pr.grad).abs().max() < 1e-5 * float(pr.grad.abs().max()) * 3.0


# ---------------------------------------------------------------------------------------------- the network
@pytest.fixture(scope="module", params=[0, 1])
def setup(request):
    from mono_depth_estimation_amd.network import Dorn
    bn = request.param
    net = Dorn.DORN(types.SimpleNamespace(**dict(DORN_ARGS, batch_norm=bn)))
    sd = W.dorn_fixture_state(net, 59 + bn)
    rgb, tgt = W.synthetic_batch(59, 2, 65, 81)
    P = nets.leaf_state(sd)
    torch.manual_seed(7)
    with torch.no_grad():
        nets.dorn_forward(P, rgb, True, momentum=1.0, **DORN_KW)
    net.load_state_dict({k: v.clone() for k, v in P.items()})
    return bn, net.cuda(), P, rgb, tgt


def test_dorn_eval_against_oracle_and_reference(setup, golden):
    bn, net, P, rgb, tgt = setup
    g, pre = golden("dorn_net"), "bn%d_" % bn
    net.eval()
    with torch.no_grad():
        label, prob = net(rgb.cuda())
        lo, po = nets.dorn_forward(P, rgb, False, **DORN_KW)
        lq, pq = nets.dorn_forward(P, rgb, False, q=nets.bf16_round, **DORN_KW)
    assert label.shape == (2, 1, 65, 81) and label.dtype == torch.int64 and prob.shape == (2, 12, 65, 81) and prob.dtype == torch.float32
    ref = torch.from_numpy(g[pre + "eval_prob"])
    noise, e_o, e_q, e_ref = _rel(pq, po), _rel(prob.cpu(), po), _rel(prob.cpu(), pq), _rel(prob.cpu(), ref)
    flips_q = float((lq != lo).float().mean())
    flips = float((label.cpu() != torch.from_numpy(g[pre + "eval_label"]).long()).float().mean())
    print("DORN(bn=%d) eval: HIP vs fp32 oracle %.3e, vs bf16-rounding oracle %.3e, vs reference %.3e; rounding noise %.3e; "
          "labels differing from the reference %.4f (rounding oracle: %.4f)" % (bn, e_o, e_q, e_ref, noise, flips, flips_q))
    assert noise < 2e-2 and e_o < 1.5 * noise + 2e-3 and e_ref < 1.5 * noise + 2e-3 and e_q < 1.2 * noise + 2e-3
    assert flips <= 1.5 * flips_q + 1e-2
    assert float((label.cpu() - torch.from_numpy(g[pre + "eval_label"]).long()).abs().max()) <= 2
    # the depth the module reports (modules/dorn.py:95-100) and its AbsRel against the target
    d_hip, d_ref = L.sid_depth(label.cpu().float(), 0.02, 10.0, 12), L.sid_depth(torch.from_numpy(g[pre + "eval_label"]).float(), 0.02, 10.0, 12)
    t = tgt * 10.0
    absrel = lambda d: float(((d - t).abs() / t)[t > 0].mean())
    print("DORN(bn=%d) eval AbsRel of the decoded depth: reference %.5f, HIP %.5f" % (bn, absrel(d_ref), absrel(d_hip)))
    assert abs(absrel(d_hip) - absrel(d_ref)) < 5e-3 + 2 * abs(absrel(L.sid_depth(lq.float(), 0.02, 10.0, 12)) - absrel(d_ref))


def test_dorn_train_step_against_oracle(setup, golden):
    from mono_depth_estimation_amd import criteria
    bn, net, P0, rgb, tgt = setup
    net.train()
    net.zero_grad(set_to_none=True)
    torch.manual_seed(99)
    label, prob = net(rgb.cuda())
    y_sid = L.sid_labels(tgt * 10.0, 0.02, 10.0, 12)
    loss = criteria.ordLoss()(prob, y_sid.cuda())
    loss.backward()
    eng = next(iter(net._engines.values()))
    masks = [d.mask.cpu().clone() for d in eng.dropouts]
    assert [tuple(m.shape) for m in masks] == [(2, 2048), (2, 2560), (2, 2048)]
    for m in masks:                                  # Dropout2d(0.5): zeros and twos, about half each
        assert set(np.unique(m.numpy()).tolist()) <= {0.0, 2.0} and 0.4 < float((m == 0).float().mean()) < 0.6
    P = nets.leaf_state(P0, requires_grad=True)
    lo, po = nets.dorn_forward(P, rgb, True, masks=masks, **DORN_KW)
    want = L.ord_loss(po, y_sid)
    want.backward()
    Pq = nets.leaf_state(P0, requires_grad=True)               # the same step with the oracle's activations rounded to bf16
    _, pq = nets.dorn_forward(Pq, rgb, True, masks=masks, q=nets.bf16_round, **DORN_KW)
    L.ord_loss(pq, y_sid).backward()
    pq = pq.detach()
    noise, e_o = _rel(pq, po.detach()), _rel(prob.detach().cpu(), po.detach())
    print("DORN(bn=%d) train: ordLoss oracle %.5f, HIP %.5f; probabilities vs oracle %.3e (rounding noise %.3e)" % (bn, float(want), float(loss), e_o, noise))
    assert abs(float(loss) - float(want)) < 1e-2 * float(want)
    assert e_o < 1.5 * noise + 2e-3
    cosf = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    ratios, cosines, floor = [], {}, {}
    for k, p in net.named_parameters():
        go, gh = P[k].grad, p.grad.detach().cpu()
        assert gh.shape == go.shape and torch.isfinite(gh).all(), k
        if float(go.norm()) > 1e-9:
            ratios.append(float(gh.norm() / go.norm()))
            cosines[k], floor[k] = cosf(gh, go), cosf(Pq[k].grad, go)
    ratios = np.array(ratios)
    print("DORN gradient-norm ratios HIP / oracle, percentiles 1 10 50 90 99:", np.percentile(ratios, [1, 10, 50, 90, 99]))
    assert np.mean(np.abs(ratios - 1) < 0.15) >= 0.9
    cs, fl = np.array(list(cosines.values())), np.array(list(floor.values()))
    print("DORN gradient cosines vs the fp32 oracle, percentiles 1 10 50: HIP %s, bf16-rounding oracle %s" % (
        np.percentile(cs, [1, 10, 50]).round(3), np.percentile(fl, [1, 10, 50]).round(3)))
    worse = [k for k in cosines if cosines[k] < floor[k] - 0.1]          # direction no worse than what the rounding alone does
    assert len(worse) <= 0.02 * len(cosines), worse[:10]
    s = "SceneUnderstandingModule."
    for k in (s + "concat_process.3.weight", s + "concat_process.3.bias", s + "concat_process.1.0.weight", s + "aspp1.0.0.weight",
              s + "aspp3.0.0.weight", s + "aspp4.1.0.weight", s + "encoder.global_fc.weight", s + "encoder.global_fc.bias",
              s + "encoder.conv1.weight"):
        assert cosines[k] >= min(0.95, floor[k] - 0.03), (k, cosines[k], floor[k])
    b = "backbone.backbone."
    for k in (b + "layer4.2.conv2.weight", b + "layer3.11.conv2.weight", b + "layer2.0.downsample.0.weight", b + "layer1.0.conv1.weight",
              b + "conv3.weight", b + "conv1.weight"):
        assert cosines[k] >= min(0.85, floor[k] - 0.05), (k, cosines[k], floor[k])
    rm = net.state_dict()[b + "layer4.2.bn3.running_mean"].cpu()
    assert _rel(rm, P[b + "layer4.2.bn3.running_mean"]) < 2e-2


def test_dorn_sgd_steps_reduce_the_loss(setup):
    """modules/dorn.py:188-193: SGD(weight_decay 5e-4), backbone at 1x, scene module at 10x; through the fused flat-range step
    (the reference's optimiser has no momentum)."""
    from mono_depth_estimation_amd import criteria
    bn, net, _, rgb, tgt = setup
    crit = criteria.ordLoss()
    x, t = rgb.cuda(), L.sid_labels(tgt * 10.0, 0.02, 10.0, 12).cuda()
    net.train()
    losses = []
    eng = None
    for i in range(5):
        net.zero_grad(set_to_none=True)
        _, prob = net(x)
        if eng is None:
            eng = next(iter(net._engines.values()))
            for d in eng.dropouts:                  # the same dropout draw every step: the loss of a FIXED function must fall
                d.fixed = d.mask.clone()
        loss = crit(prob, t)
        loss.backward()
        net._store.sgd_step(1e-4, 1e-3, momentum=0.0, weight_decay=5e-4)
        losses.append(float(loss))
    for d in eng.dropouts:
        d.fixed = None
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_dorn_default_geometry_257x353():
    """The module's defaults (modules/dorn.py:205-217): 257 x 353 input, ord_num 68, kernel_size 16, pyramid 4 / 8 / 12, no
    BatchNorm in the scene module, dropout 0.5 — one training step at batch 4, checked through properties: shapes, label range,
    probabilities consistent with the labels, finite gradients for every parameter, eval mode reproducible."""
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import Dorn
    args = types.SimpleNamespace(input_size=(257, 353), kernel_size=16, ord_num=68.0, alpha=0.02, beta=10.0, discretization="SID", pretrained=0,
                                 pyramid=[4, 8, 12], batch_norm=0, dropout=0.5)
    torch.manual_seed(0)
    net = Dorn.DORN(args)
    W.dorn_fixture_state(net, 67)
    net = net.cuda().train()
    rgb, tgt = W.synthetic_batch(67, 4, 257, 353)
    x = rgb.cuda()
    t = L.sid_labels(tgt * 10.0, 0.02, 10.0, 68).cuda()
    label, prob = net(x)
    assert label.shape == (4, 1, 257, 353) and prob.shape == (4, 68, 257, 353)
    assert int(label.min()) >= 0 and int(label.max()) <= 68
    assert torch.equal(label, (prob > 0.5).sum(1, keepdim=True))
    loss = criteria.ordLoss()(prob, t)
    loss.backward()
    assert np.isfinite(float(loss))
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert float(net.SceneUnderstandingModule.encoder.global_fc.weight.grad.abs().max()) > 0
    assert float(net.backbone.backbone.conv1.weight.grad.abs().max()) > 0
    net._store.sgd_step(1e-4, 1e-3, momentum=0.0, weight_decay=5e-4)
    net.eval()
    with torch.no_grad():
        l1, p1 = net(x)
        l2, p2 = net(x)
    assert torch.equal(l1, l2) and torch.equal(p1, p2)
