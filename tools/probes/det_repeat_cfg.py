"""Diagnostic: repeated deterministic-mode steps of BASELINE configurations 3-5 at their benchmark sizes (bench.build_other) must give
bit-identical gradients with the weight gradients on the second stream -- a cross-stream race would show as run-to-run differences.
    python tools/probes/det_repeat_cfg.py bts|midas|vnl"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

name = sys.argv[1]
dev = torch.device("cuda:0")
net, fwd_loss, _ = bench.build_other(name, bench.OTHER_CONFIGS[name]["batch"], dev)
np.random.seed(3)
fwd_loss()                                                # builds store and plan
net._store.set_deterministic(True)


def grads():
    np.random.seed(3)                                     # (VNL's point triples come from numpy's global stream)
    net.zero_grad(set_to_none=True)
    fwd_loss().backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}


runs = [grads() for _ in range(4)]                        # (run 0 traces the fused-sum plan: compare the later ones with run 1)
names = list(runs[1])
for i in range(2, 4):
    bad = [n for n in names if not torch.equal(runs[1][n], runs[i][n])]
    print("%s run %d vs run 1: %d of %d tensors differ: %s" % (name, i, len(bad), len(names), bad[:4]))
    if bad:
        last = names[max(names.index(n) for n in bad)]
        a, b = runs[1][last].float(), runs[i][last].float()
        print("   last differing (first in backward order): %s: %d of %d elements, max |diff| %.3g against max |value| %.3g" % (
            last, int((a != b).sum()), a.numel(), float((a - b).abs().max()), float(a.abs().max())))
        rel = sorted(((float((runs[1][n].float() - runs[i][n].float()).abs().max() / (runs[1][n].float().abs().max() + 1e-30)), n) for n in bad), reverse=True)[:3]
        print("   largest relative differences:", [(n[-50:], "%.2e" % r) for r, n in rel])
