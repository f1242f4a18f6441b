"""Diagnostic: as det_bts_lpg.py but WITHOUT synchronising the streams: the inputs of the plane_params convolution's backward
(its output gradient and its input activation) are cloned on the main stream right before the op runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from mono_depth_estimation_amd import graph as G  # noqa: E402

dev = torch.device("cuda:0")
net, fwd_loss, _ = bench.build_other("bts", 16, dev)
fwd_loss()
net._store.set_deterministic(True)
eng = next(iter(net._engines.values()))
target = [op for op in eng.tape if isinstance(op, G.PlaneDepth) and op.up == 2][0]
conv = eng.tape[eng.tape.index(target) - 1]
assert isinstance(conv, G.Conv), type(conv)
snaps = []
orig = G.Conv.bwd


def bwd(self):
    if self is conv:
        snaps[-1]["og"] = self.out.g.clone()
        snaps[-1]["x_t"] = self.x.t.clone()
    orig(self)
    if self is conv:
        snaps[-1]["x_g_after"] = self.x.g.clone()


G.Conv.bwd = bwd
name = "decoder.reduc2x2.reduc.plane_params.weight"
for it in range(4):
    snaps.append({})
    net.zero_grad(set_to_none=True)
    fwd_loss().backward()
    torch.cuda.synchronize()
    snaps[-1]["dW"] = dict(net.named_parameters())[name].grad.clone()
for it in range(2, 4):
    a, b = snaps[1], snaps[it]
    print("run %d vs 1:" % it, {k: ("equal" if torch.equal(a[k], b[k]) else "DIFFERS (%d, max %.3g)" % (int((a[k] != b[k]).sum()), float((a[k].float() - b[k].float()).abs().max()))) for k in a})
