"""Diagnostic: the HIP eval path's AbsRel against the fp32 oracle on the off-grid conditioned states, over several INPUT batches
(same weights, same running statistics): the spread of |dAbsRel| the 1e-4 assertion of tests/test_offgrid_gpu.py lives in."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import offgrid_states as S
from oracle import metrics as OM, nets, weights as W, losses as L

names = sys.argv[1:] or ["bts", "midas", "vnl"]
for name in names:
    (net, P), oracle, rgb0, tgt0, select = getattr(S, name)()
    net.load_state_dict({k: v.clone() for k, v in P.items()})
    net = net.cuda().eval()
    fwd = {"bts": lambda x: nets.bts_forward(P, x, False)[4], "midas": lambda x: nets.midas_forward(P, x, False)[:, :1],
           "vnl": lambda x: L.bins_to_depth(nets.vnl_forward(P, x, False)[1], torch.tensor(nets.vnl_params().depth_bin_border, dtype=torch.float32))}[name]
    rows = []
    for seed in range(100, 106):
        rgb, tgt = W.synthetic_batch(seed, rgb0.shape[0], *rgb0.shape[2:])
        tgt = tgt * (10.0 if name == "bts" else 1.0)
        with torch.no_grad():
            yo = fwd(rgb)
            d = []
            for split in (True, False):
                net._store.split_eval = split
                y = select(net(rgb.cuda()))
                y = y.cpu() if torch.is_tensor(y) else y
                d.append(float(OM.compute(y.float(), tgt)["absrel"]) - float(OM.compute(yo, tgt)["absrel"]))
        rows.append(d)
    r = np.array(rows)
    print("%-6s dAbsRel (HIP - fp32 oracle) over 6 input batches: two-term %s mean %.2e std %.2e | one-term mean %.2e std %.2e" % (
        name, np.array2string(r[:, 0], precision=1), r[:, 0].mean(), r[:, 0].std(), r[:, 1].mean(), r[:, 1].std()))
