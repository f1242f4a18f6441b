#!/usr/bin/env python3
"""Time the stdepth composite criterion at the FCRN bench shape (32 x 10 x 480 x 640)."""
import sys
import types

import torch

sys.path.insert(0, ".")
from mono_depth_estimation_amd import stdepth  # noqa: E402

N, C, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 10, 480, 640
torch.manual_seed(0)
pred = (torch.rand(N, C, H, W, device="cuda") * 1.2 - 0.1)
pred[:, 8:] = pred[:, 8:].abs() + 0.05
pred.requires_grad_(True)
targ = torch.rand(N, C, H, W, device="cuda")
rgba = torch.rand(N, 4, H, W, device="cuda")
rgba[:, 3] *= (torch.rand(N, H, W, device="cuda") > 0.3)
px = N * H * W


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for loss in ("mae+composite", "silma", "mae+composite+ssim", "allssim+colorssim"):
    m = types.SimpleNamespace(loss=loss, variance_focus=0.85, depth_loss_weight=10.0, comp_loss_weight=2.0,
                              fbdiv_loss_weight=0.2, ssim_loss_weight=2.0)
    crit = stdepth.setup_criterion(m, True)
    f = timed(lambda: crit(pred.detach(), targ, rgba))
    fb = timed(lambda: torch.autograd.grad(crit(pred, targ, rgba)[0], pred))
    # algorithmic bytes without SSIM: fwd reads pred + targ + rgba; bwd reads them again and writes grad
    fwd_b, bwd_b = px * (2 * C + 4) * 4, px * (3 * C + 4) * 4
    print("%-22s fwd %7.3f ms  fwd+bwd %7.3f ms   (%.0f / %.0f GB/s on the non-SSIM byte count)"
          % (loss, f, fb, fwd_b / f / 1e6, (fwd_b + bwd_b) / fb / 1e6))
