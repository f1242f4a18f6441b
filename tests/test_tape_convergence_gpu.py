"""Convergence parity of a tape-run network against its CPU oracle: MiDaS (ResNeXt-101 32x8d: grouped convolutions, biased
convolutions, the in-place-ReLU fusion blocks) trained for 20 Adam steps (modules/midas.py:94-105: encoder at the lower rate) on
one fixed batch from the same state — the HIP path through its fused flat-range step, the fp32 functional oracle through
torch.optim.Adam.  Asserted: the two loss CURVES (MidasLoss(0.5, 'ssimse')) stay within 2 % of each other at every step and
0.7 % on average (measured: 0.19 % / 0.02 %), both fall, and on the state the ORACLE reached the two eval paths agree to the rounding-noise level."""
import numpy as np
import pytest
import torch

from oracle import losses as L
from oracle import nets
from oracle import weights as W

pytestmark = pytest.mark.gpu
SIZE = (64, 96)
STEPS = 20


def test_midas_loss_curves_agree_with_the_oracle():
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import MiDaS
    torch.manual_seed(0)
    net = MiDaS.MidasNet(features=256)
    sd = W.midas_fixture_state(net, 43)
    rgb, tgt = W.synthetic_batch(43, 2, *SIZE)
    P = nets.leaf_state(sd, requires_grad=True)
    with torch.no_grad():
        nets.midas_forward(P, rgb, True, momentum=1.0)
    net.load_state_dict({k: v.detach().clone() for k, v in P.items()})
    net = net.cuda().train()
    x, t = rgb.cuda(), tgt.cuda()
    crit = criteria.MidasLoss(alpha=0.5, loss="ssimse")
    lr_enc, lr_dec = 2e-5, 2e-4
    lh = []
    for _ in range(STEPS):
        net.zero_grad(set_to_none=True)
        loss = crit(net(x)[:, :1], t)
        loss.backward()
        net._store.adam_step(lr_enc, lr_dec)
        lh.append(float(loss))
    params = {k: v for k, v in P.items() if v.requires_grad}
    opt = torch.optim.Adam([{"params": [v for k, v in params.items() if k.startswith("pretrained.")], "lr": lr_enc},
                            {"params": [v for k, v in params.items() if not k.startswith("pretrained.")], "lr": lr_dec}], lr=lr_dec)
    lo = []
    for _ in range(STEPS):
        opt.zero_grad()
        loss = L.midas_loss(nets.midas_forward(P, rgb, True)[:, :1], tgt, alpha=0.5, loss="ssimse")
        loss.backward()
        opt.step()
        lo.append(float(loss))
    lh, lo = np.array(lh), np.array(lo)
    print("MidasLoss, HIP   :", np.round(lh[[0, 1, 2, 4, 9, 14, 19]], 5))
    print("MidasLoss, oracle:", np.round(lo[[0, 1, 2, 4, 9, 14, 19]], 5))
    band = np.abs(lh - lo) / lo
    print("relative gap between the curves: max %.4f mean %.4f; fall HIP %.4f oracle %.4f" % (band.max(), band.mean(), lh[-1] / lh[0], lo[-1] / lo[0]))
    assert np.isfinite(lh).all() and lh[-1] < lh[0] and lo[-1] < lo[0]
    assert band.max() < 2e-2 and band.mean() < 7e-3
    trained = {k: v.detach().clone() for k, v in P.items()}
    net.load_state_dict(trained)
    net.eval()
    with torch.no_grad():
        yh = net(x).cpu()
        yo = nets.midas_forward(trained, rgb, False)
        yq = nets.midas_forward(trained, rgb, False, q=nets.bf16_round)
    rel, noise = float((yh - yo).norm() / yo.norm()), float((yq - yo).norm() / yo.norm())
    print("trained-like state, eval: HIP vs oracle %.3e (bf16-rounding noise of the oracle %.3e)" % (rel, noise))
    assert rel < 1.5 * noise + 3e-3


def test_tape_weight_gradient_stream_changes_nothing_but_the_schedule():
    """As tests/test_fcrn_convergence_gpu.py's two-stream check, for a tape-run network: in deterministic mode a MiDaS step with the
    weight-gradient stream and one without it give BIT-identical gradients."""
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import MiDaS
    torch.manual_seed(0)
    net = MiDaS.MidasNet(features=256)
    W.midas_fixture_state(net, 43)
    net = net.cuda().train()
    net._store.set_deterministic(True)
    try:
        rgb, tgt = W.synthetic_batch(43, 2, *SIZE)
        x, t = rgb.cuda(), tgt.cuda()
        crit = criteria.MidasLoss(alpha=0.5, loss="ssimse")

        def grads():
            net.zero_grad(set_to_none=True)
            crit(net(x)[:, :1], t).backward()
            return torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None]).clone()
        grads()         # (the first backward of a plan traces who completes each BatchNorm's gradient and runs the separate
        #                  reduction passes; from the second on the sums come from the conv epilogues: graph._plan_fused_sums)
        g_two = grads()
        eng = next(iter(net._engines.values()))
        if eng.side is None:
            pytest.skip("MDE_WGRAD_STREAM=0: nothing to compare")
        side, eng.side = eng.side, None
        g_one = grads()
        eng.side = side
        assert torch.equal(g_two, g_one) and torch.equal(g_two, grads())
    finally:
        net._store.set_deterministic(False)
