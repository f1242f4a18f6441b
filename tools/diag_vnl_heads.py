#!/usr/bin/env python3
"""Diagnostic: the VNL network's backward for a gradient on the logits only / on the softmax output only, HIP against the oracle."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nets, weights as W
from mono_depth_estimation_amd.network import VNL

SIZE = (64, 96)
params = nets.vnl_params(); params.crop_size = SIZE
torch.manual_seed(0)
net = VNL.MetricDepthModel(params)
sd = W.vnl_fixture_state(net, 41)
rgb, tgt = W.synthetic_batch(41, 2, *SIZE)
P0 = nets.leaf_state(sd)
with torch.no_grad():
    nets.vnl_forward(P0, rgb, True, momentum=1.0)
net.load_state_dict({k: v.detach().clone() for k, v in P0.items()})
net = net.cuda().train()
g = torch.Generator().manual_seed(3)
gl = torch.randn(2, 150, *SIZE, generator=g) * 1e-3
gp = torch.randn(2, 150, *SIZE, generator=g)
keys = ["depth_model.decoder_modules.topdown_predict.conv1.weight", "depth_model.decoder_modules.topdown_predict.conv1.bias",
        "depth_model.decoder_modules.topdown_fcn5.ftb.conv3.weight", "depth_model.encoder_modules.bottomup.res1.conv1.weight"]
for mode in ("logit", "prob", "both"):
    net.zero_grad(set_to_none=True)
    logit, prob = net(rgb.cuda())
    a = gl.cuda() if mode in ("logit", "both") else torch.zeros_like(logit)
    b = gp.cuda() if mode in ("prob", "both") else torch.zeros_like(prob)
    torch.autograd.backward([logit, prob], [a, b])
    P = nets.leaf_state(P0, requires_grad=True)
    lo, po = nets.vnl_forward(P, rgb, True)
    ((lo * a.cpu()).sum() + (po * b.cpu()).sum()).backward()
    named = dict(net.named_parameters())
    for k in keys:
        gh, go = named[k].grad.detach().cpu(), P[k].grad
        print("%-6s %-70s ratio %.4f cos %.4f" % (mode, k.split("modules.")[-1], float(gh.norm() / go.norm()), float((gh * go).sum() / (gh.norm() * go.norm()))))
