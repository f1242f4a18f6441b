"""Diagnostic: BTS at the benchmark size, deterministic mode, two streams: which tensor around the local-planar-guidance heads
differs from run to run?  Snapshots, inside PlaneDepth.bwd, the incoming map gradient, the plane-parameter activations and the
gradient the kernel writes; and the weight gradient of the plane_params conv afterwards."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from mono_depth_estimation_amd import graph as G, ops  # noqa: E402

dev = torch.device("cuda:0")
net, fwd_loss, _ = bench.build_other("bts", 16, dev)
fwd_loss()
net._store.set_deterministic(True)
snaps = []
orig = G.PlaneDepth.bwd


def bwd(self):
    torch.cuda.synchronize()
    rec = {"up": self.up, "map_g_in": self.map.g.clone(), "x_t": self.x.t.clone()}
    if self.douts[0] is not None:
        rec["dout"] = self.douts[0].clone()
    orig(self)
    torch.cuda.synchronize()
    rec["x_g"] = self.x.g.clone()
    snaps[-1].append(rec)


G.PlaneDepth.bwd = bwd
for it in range(4):
    snaps.append([])
    net.zero_grad(set_to_none=True)
    fwd_loss().backward()
    torch.cuda.synchronize()
for it in range(2, 4):
    for a, b in zip(snaps[1], snaps[it]):
        print("run %d vs 1, lpg x%d:" % (it, a["up"]), {k: ("equal" if torch.equal(a[k], b[k]) else "DIFFERS (%d elements, max %.3g)" % (
            int((a[k] != b[k]).sum()), float((a[k].float() - b[k].float()).abs().max()))) for k in a if k != "up"})
