#!/usr/bin/env python3
"""The reference FCRNModule's own default configuration through the drop-in module path (modules/laina.py:8-16,51-62,
69-72): FCRN.ResNet(output_size=(240, 320), out_channels=20), batch 16 of 240x320 crops, criterion 'mae+composite'
with single_layer, torch.optim.Adam with the 1x / 10x parameter groups, metrics every step."""
import sys
import time
import types

import torch

sys.path.insert(0, ".")
from mono_depth_estimation_amd import metrics, stdepth  # noqa: E402
from mono_depth_estimation_amd.network import FCRN  # noqa: E402

B, size = int(sys.argv[1]) if len(sys.argv) > 1 else 16, (240, 320)
torch.manual_seed(0)
net = FCRN.ResNet(output_size=size, out_channels=20, pretrained=False).cuda().train()
method = types.SimpleNamespace(loss="mae+composite", variance_focus=0.85, depth_loss_weight=10.0, comp_loss_weight=2.0,
                               fbdiv_loss_weight=0.2, ssim_loss_weight=2.0)
crit = stdepth.setup_criterion(method, single_layer=True)
opt = torch.optim.Adam([{"params": net.get_1x_lr_params(), "lr": 1e-4}, {"params": net.get_10x_lr_params(), "lr": 1e-3}], lr=1e-4)
mc = metrics.MetricComputation(['delta1', 'delta2', 'delta3', 'mse', 'mae', 'log10', 'rmse'])
x = torch.rand(B, 3, *size, device="cuda")
y = torch.rand(B, 20, *size, device="cuda")
rgba = torch.rand(B, 4, *size, device="cuda")
rgba[:, 3] *= (torch.rand(B, *size, device="cuda") > 0.3)


def step():
    opt.zero_grad(set_to_none=True)
    y_hat = net(x)
    loss, pred_full = crit(y_hat, y, rgba, return_composited=True)
    loss.backward()
    opt.step()
    mc.compute(y_hat[:, 8:10], y[:, 8:10])
    return loss


for _ in range(3):
    l0 = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("laina default (batch %d, 240x320, 20 channels, mae+composite, torch Adam, metrics): %.2f ms/step, %.0f images/s, "
      "loss %.4f -> %.4f" % (B, 1e3 * dt, B / dt, float(l0), float(l)))
