"""Convergence parity of the HIP FCRN training path against the CPU oracle.

40 Adam steps (laina.py:51-57: encoder LR, decoder 10 x LR) on one fixed batch of 4 x 3 x 96 x 128 from the SAME He-initialised
state: the HIP path through its fused step, the fp32 oracle through torch.optim.Adam.  Train-mode BatchNorm on bf16
activations makes individual steps differ, so what is asserted is the CURVE: both fall by the same
factor (measured 0.9704 vs 0.9708) and stay within 2 % of each other at every step (measured max 0.8 %).  Then the 1e-4 AbsRel bound of the north-star is
checked on the state the ORACLE reached after those steps (a trained-like state instead of the gamma-scaled fixture):
identical trained weights through both eval paths."""
import numpy as np
import pytest
import torch

from oracle import fcrn as ofcrn
from oracle import losses as OL
from oracle import metrics as OM
from oracle import weights as W

pytestmark = pytest.mark.gpu
SIZE = (96, 128)
STEPS = 40


def test_loss_curves_agree_and_trained_state_absrel():
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import FCRN
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    ora = ofcrn.FCRNOracle(50, SIZE, out_channels=1)
    sd = W.fcrn_fixture_state(ora, 9)
    rgb, tgt = W.synthetic_batch(9, 4, *SIZE)
    hip = FCRN.ResNet(layers=50, output_size=SIZE, out_channels=1, pretrained=False)
    hip.load_state_dict(sd)
    hip = hip.cuda().train()
    x, t = rgb.cuda(), tgt.cuda()
    crit = criteria.silog_loss(0.85)
    lr = 2e-4
    lh = []
    for _ in range(STEPS):
        hip.zero_grad(set_to_none=True)
        loss = crit(hip(x), t)
        loss.backward()
        hip._store.adam_step(lr, 10 * lr)
        lh.append(float(loss))
    ora.train()
    opt = torch.optim.Adam([{"params": ora.get_1x_lr_params(), "lr": lr}, {"params": ora.get_10x_lr_params(), "lr": 10 * lr}], lr=lr)
    lo = []
    for _ in range(STEPS):
        opt.zero_grad()
        loss = OL.silog(ora(rgb), tgt, 0.85)
        loss.backward()
        opt.step()
        lo.append(float(loss))
    lh, lo = np.array(lh), np.array(lo)
    print("SILog, HIP   :", np.round(lh[[0, 1, 2, 4, 9, 19, 29, 39]], 4))
    print("SILog, oracle:", np.round(lo[[0, 1, 2, 4, 9, 19, 29, 39]], 4))
    assert np.isfinite(lh).all() and abs(lh[0] - lo[0]) < 1e-2 * lo[0]
    # (the synthetic target is noise: SILog can only fall to its variance floor, ~3 % below the start; what matters is that
    #  both paths trace the same curve, including the overshoot of the first Adam step)
    assert lh[-1] < lh[0] and lo[-1] < lo[0]
    assert abs(lh[-1] / lh[0] - lo[-1] / lo[0]) < 5e-3, (lh[-1] / lh[0], lo[-1] / lo[0])
    assert lh[1] > lh[0] and lo[1] > lo[0]                    # the first step's overshoot is reproduced
    band = np.abs(lh - lo) / lo
    print("relative gap between the curves: max %.4f mean %.4f" % (band.max(), band.mean()))
    assert band.mean() < 5e-3 and band.max() < 2e-2
    # ---- the trained-like state: identical weights (the oracle's, after its 40 steps) through both eval paths
    trained = {k: v.detach().clone() for k, v in ora.state_dict().items()}
    hip.load_state_dict(trained)
    hip.eval()
    ora.eval()
    with torch.no_grad():
        yh, yo = hip(x).cpu(), ora(rgb)
    ah, ao = float(OM.compute(yh, tgt)["absrel"]), float(OM.compute(yo, tgt)["absrel"])
    rel = float((yh - yo).norm() / yo.norm())
    print("trained-like state, eval: AbsRel HIP %.6f oracle %.6f (|d| %.2e), output rel L2 %.3e" % (ah, ao, abs(ah - ao), rel))
    # weights here are NOT bf16-representable (free-running Adam): bf16 weight quantisation alone moves AbsRel by ~1e-4
    # (DESIGN section 4), yet the north-star bound holds here too: measured |dAbsRel| 1.7e-5, output rel L2 1.4e-3
    assert abs(ah - ao) < 1e-4 and rel < 1e-2
