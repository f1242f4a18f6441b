#!/usr/bin/env python3
"""CPU: what bf16 STORAGE rounding does to the fp32 oracle's own AbsRel on the conditioned fixtures, over several realisations
of the rounding (oracle/nets.rounding_draw) -- the floor under every `|dAbsRel| <= 1e-4` assertion of the GPU tests.  One
realisation says little: the rounding noise of coarse feature maps is coherent over image regions, so the shift of a mean
over pixels has a spread (and, through double roundings across constants, a systematic part) of its own.

    python tools/rounding_draws.py [bts|midas|vnl ...] [--draws 6]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import offgrid_states as S  # noqa: E402
from oracle import metrics as OM  # noqa: E402
from oracle import nets  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("nets", nargs="*", default=["bts", "midas", "vnl"])
    ap.add_argument("--draws", type=int, default=6)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    for name in a.nets:
        _, oracle, _, tgt, _ = getattr(S, name)()
        yo = oracle()
        a0 = float(OM.compute(yo, tgt)["absrel"])
        rows = []
        for k in range(a.draws):
            yq = oracle(q=nets.rounding_draw(k))
            rows.append((float((yq - yo).norm() / yo.norm()), float(OM.compute(yq, tgt)["absrel"]) - a0))
        r = np.array(rows)
        print("%-6s AbsRel %.6f | output noise %.2e | dAbsRel per draw %s | mean %.2e, spread %.2e" % (
            name, a0, r[:, 0].mean(), np.array2string(r[:, 1], precision=1), r[:, 1].mean(), r[:, 1].std()))


if __name__ == "__main__":
    main()
