#!/usr/bin/env python3
"""Three backward passes of one tape network on the same batch: relative change of every gradient tensor between pass 0 and 1 and
between 1 and 2.  With the BatchNorm-backward sums fused (default) pass 0 is the trace pass with separate reductions;
MDE_FUSE_BN_RED=0 gives the spread of the unfused path.   python tools/diag_fused_sums.py midas|vnl|bts|bts_resnet"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import weights as W  # noqa: E402
from test_fused_bn_sums_gpu import SIZE, _build  # noqa: E402

kind = sys.argv[1]
net, pick = _build(kind)
rgb, _ = W.synthetic_batch(71, 2, *SIZE)
x = rgb.cuda()
wts, grads = None, []
for step in range(3):
    net.zero_grad(set_to_none=True)
    y = pick(net(x))
    if wts is None:
        wts = torch.from_numpy(np.random.default_rng(7).standard_normal(tuple(y.shape)).astype(np.float32)).cuda()
    (y * wts).mean().backward()
    grads.append({k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None})
rows = []
for k, g0 in grads[0].items():
    n = float(g0.norm()) + 1e-20
    rows.append((float((grads[1][k] - g0).norm()) / n, float((grads[2][k] - grads[1][k]).norm()) / n, k, n))
rows.sort(reverse=True)
print("%s fuse=%s: worst 0->1 %.3e, worst 1->2 %.3e; median 0->1 %.3e, 1->2 %.3e" % (
    kind, os.environ.get("MDE_FUSE_BN_RED", "1"), rows[0][0], max(r[1] for r in rows), float(np.median([r[0] for r in rows])),
    float(np.median([r[1] for r in rows]))))
for r in rows[:6]:
    print("   %.3e  %.3e  %-60s |g| %.3e" % r)
