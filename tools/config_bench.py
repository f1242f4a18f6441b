#!/usr/bin/env python3
"""Training throughput of BASELINE.json configurations 3 / 4 / 5 at their per-GPU sizes on one MI355X (module path: network
forward -> drop-in criterion -> backward -> fused optimiser step), one JSON line per configuration.

    python tools/config_bench.py [vnl] [midas] [bts] [dorn] [mynet] [--steps K] [--warmup W]

GMAC figures are SURVEY.md 8a's dense-as-written forward MACs per image (x 6 = training FLOP)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 2500.0


def data(n, h, w, seed=1):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    rgb = torch.rand(n, 3, h, w, generator=g, device="cuda")
    depth = 0.05 + 0.95 * torch.rand(n, 1, h, w, generator=g, device="cuda")
    return rgb, depth.masked_fill(torch.rand(n, 1, h, w, generator=g, device="cuda") < 0.1, 0.0)


def timed(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def vnl(args):
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import VNL
    from types import SimpleNamespace
    C, dmin, dmax = 150, 0.01, 1.1
    interval = (np.log10(dmax) - np.log10(dmin)) / C
    p = SimpleNamespace(depth_min=dmin, encoder="resnext50_32x4d_body_stride16", pretrained=0, freeze_backbone=False, init_type="xavier",
                        enc_dim_in=[64, 256, 512, 1024, 2048], enc_dim_out=[512, 256, 256, 256], dec_dim_in=[512, 256, 256, 256, 256, 256],
                        dec_dim_out=[256, 256, 256, 256, 256], dec_out_c=C, focal_x=519.0, focal_y=519.0, crop_size=(480, 640), diff_loss_weight=6,
                        depth_min_log=np.log10(dmin), depth_bin_interval=interval,
                        wce_loss_weight=[[np.exp(-0.2 * (i - j) ** 2) for i in range(C)] for j in np.arange(C)],
                        depth_bin_border=np.array([np.log10(dmin) + interval * (i + 0.5) for i in range(C)]))
    torch.manual_seed(0)
    net = VNL.MetricDepthModel(p).cuda().train()
    n = args.batch or 16
    x, gt = data(n, 480, 640)
    crit = criteria.ModelLoss(p)
    bins = criteria.depth_to_bins(gt, dmin, dmax, C)

    def step():
        net.zero_grad(set_to_none=True)
        logit, prob = net(x)
        crit(criteria.bins_to_depth(prob, p.depth_bin_border), logit, bins, gt).backward()
        net._store.sgd_step(1e-4, 1e-5, momentum=0.9, weight_decay=5e-4)
    dt = timed(step, args.steps, args.warmup)
    return {"config": "VNL resnext50_32x4d stride 16, 150 bins, %dx3x480x640, ModelLoss (WCEL + 6 VNL), SGD m0.9" % n, "ms_per_step": 1e3 * dt,
            "images_per_sec": n / dt, "fwd_gmac_per_image": 348.42, "step_mfma_frac": n / dt * 348.42 * 6e9 / (PEAK * 1e12)}


def midas(args):
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import MiDaS
    torch.manual_seed(0)
    net = MiDaS.MidasNet(features=256).cuda().train()
    n = args.batch or 32
    x, gt = data(n, 384, 384)
    crit = criteria.MidasLoss(alpha=0.5, loss="ssimse")

    def step():
        net.zero_grad(set_to_none=True)
        crit(net(x)[:, :1], gt).backward()
        net._store.adam_step(1e-5, 1e-4)
    dt = timed(step, args.steps, args.warmup)
    return {"config": "MiDaS ResNeXt-101 32x8d, %dx3x384x384, MidasLoss(0.5, ssimse), Adam" % n, "ms_per_step": 1e3 * dt,
            "images_per_sec": n / dt, "fwd_gmac_per_image": 103.47, "step_mfma_frac": n / dt * 103.47 * 6e9 / (PEAK * 1e12)}


def bts(args):
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import Bts
    torch.manual_seed(0)
    net = Bts.BtsModel(max_depth=1.0, bts_size=512, encoder_version="densenet161_bts", out_channels=1).cuda().train()
    n = args.batch or 16
    x, gt = data(n, 480, 640)
    crit = criteria.silog_loss(0.85)

    def step():
        net.zero_grad(set_to_none=True)
        crit(net(x)[4], gt).backward()
        net._store.adam_step(1e-4, 1e-4, eps=1e-3, weight_decay=(1e-2, 0.0), decoupled=True)
    dt = timed(step, args.steps, args.warmup)
    return {"config": "BTS DenseNet-161, bts_size 512, %dx3x480x640, SILog, AdamW" % n, "ms_per_step": 1e3 * dt,
            "images_per_sec": n / dt, "fwd_gmac_per_image": 121.48, "step_mfma_frac": n / dt * 121.48 * 6e9 / (PEAK * 1e12)}


def plan_gmac(net):
    """Forward multiply-accumulates per image of the launch plan the module just ran (dense convs as written, grouped convs by
    their own group width), from the tape's conv ops."""
    from mono_depth_estimation_amd import graph as G
    eng = next(iter(net._engines.values()))
    macs = 0
    for op in eng.tape:
        if isinstance(op, G.Conv):
            o, c = op.out, op.conv
            macs += o.N * o.H * o.W * c.O * c.T * (getattr(c, "G", 0) or c.I)
        elif isinstance(op, G.ConvT):
            x, c = op.x, op.w
            macs += x.N * x.H * x.W * c.O * c.T * c.I
        elif isinstance(op, (G.Stem, G.ImageStem)):
            o = op.c if isinstance(op, G.Stem) else op.out
            macs += o.N * o.H * o.W * o.C * op.w.T * 3
    return macs / eng.N / 1e9


def dorn(args):
    """The DORN module's defaults (modules/dorn.py:205-217): 257 x 353, ord_num 68, SGD with weight decay, backbone 1x / scene module 10x."""
    from types import SimpleNamespace
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import Dorn
    a = SimpleNamespace(input_size=(257, 353), kernel_size=16, ord_num=68.0, alpha=0.02, beta=10.0, discretization="SID", pretrained=0,
                        pyramid=[4, 8, 12], batch_norm=0, dropout=0.5)
    torch.manual_seed(0)
    net = Dorn.DORN(a).cuda().train()
    n = args.batch or 16
    x, gt = data(n, 257, 353)
    t = 68.0 * torch.log(gt.clamp(min=1e-3) * 10.0 / 0.02) / float(np.log(10.0 / 0.02))
    crit = criteria.ordLoss()

    def step():
        net.zero_grad(set_to_none=True)
        crit(net(x)[1], t).backward()
        net._store.sgd_step(1e-4, 1e-3, momentum=0.0, weight_decay=5e-4)
    dt = timed(step, args.steps, args.warmup)
    g = plan_gmac(net)
    return {"config": "DORN dilated ResNet-101, ord_num 68, %dx3x257x353, ordLoss, SGD" % n, "ms_per_step": 1e3 * dt,
            "images_per_sec": n / dt, "fwd_gmac_per_image": g, "step_mfma_frac": n / dt * g * 6e9 / (PEAK * 1e12)}


def mynet(args):
    """The MyNet module's defaults (modules/my.py:27-36,66-70,160): 384 x 384, batch 16, MidasLoss(0.5, 'mse'), Adam 1x / 10x."""
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import MyNet
    torch.manual_seed(0)
    net = MyNet.MyModel().cuda().train()
    n = args.batch or 16
    x, gt = data(n, 384, 384)
    crit = criteria.MidasLoss(alpha=0.5, loss="mse", reduction="batch-based")

    def step():
        net.zero_grad(set_to_none=True)
        crit(net(x), gt * 10.0).backward()
        net._store.adam_step(1e-4, 1e-3)
    dt = timed(step, args.steps, args.warmup)
    g = plan_gmac(net)
    return {"config": "MyNet DenseNet-161, %dx3x384x384, MidasLoss(0.5, mse), Adam" % n, "ms_per_step": 1e3 * dt,
            "images_per_sec": n / dt, "fwd_gmac_per_image": g, "step_mfma_frac": n / dt * g * 6e9 / (PEAK * 1e12)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["vnl", "midas"])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0)
    args = ap.parse_args()
    for w in args.which:
        out = {"vnl": vnl, "midas": midas, "bts": bts, "dorn": dorn, "mynet": mynet}[w](args)
        out = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}
        print(json.dumps(out), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
