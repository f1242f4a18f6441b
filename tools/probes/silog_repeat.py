"""Diagnostic: is the SILog criterion's gradient bit-reproducible?  Same (prediction, target) 200 times, with and without work on
another stream while the criterion runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mono_depth_estimation_amd import criteria  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1234)
n, h, w = 16, 480, 640
pred = (0.05 + 0.9 * torch.rand(n, 1, h, w, generator=g, device="cuda")).requires_grad_(True)
gt = 0.05 + 0.95 * torch.rand(n, 1, h, w, generator=g, device="cuda")
gt = gt.masked_fill(torch.rand(n, 1, h, w, generator=g, device="cuda") < 0.10, 0.0)
crit = criteria.silog_loss(0.85)
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)


def once(busy):
    pred.grad = None
    if busy:
        with torch.cuda.stream(side):
            torch.mm(a, a)
    loss = crit(pred, gt)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), pred.grad.clone()


for busy in (False, True):
    l0, g0 = once(busy)
    nl = ng = 0
    for _ in range(200):
        l, gg = once(busy)
        nl += int(l != l0)
        ng += int(not torch.equal(gg, g0))
    print("busy=%s: loss differs in %d of 200 repeats, gradient differs in %d" % (busy, nl, ng))
