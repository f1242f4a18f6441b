"""CPU: the oracles on the OFF-GRID conditioned states (tests/offgrid_states.py) against the vectors minted from the
reference's own network classes and metrics.py (tests/golden/offgrid.npz, gen_golden.py::gen_offgrid) -- the pin of the
fixtures on which tests/test_offgrid_gpu.py asserts the north-star bound |dAbsRel| <= 1e-4 -- and, on the oracle alone, what
a ONE-term 16-bit weight shadow would cost there (why the HIP path's eval forward contracts with a two-term shadow)."""
import numpy as np
import pytest
import torch

import offgrid_states as S
from oracle import metrics as OM
from oracle import nets


def _check(tag, oracle, tgt, g, out_rtol=2e-5):
    y = oracle()
    m = OM.compute(y, tgt)
    for n in ("absrel", "rmse", "log10"):
        assert abs(float(m[n]) - float(g["%s_%s" % (tag, n)])) < 3e-6, (tag, n, float(m[n]), float(g["%s_%s" % (tag, n)]))
    if tag + "_out" in g:
        ref = torch.from_numpy(g[tag + "_out"])
        assert float((y[:ref.shape[0]] - ref).norm() / ref.norm()) < out_rtol, tag
    return float(m["absrel"])


@pytest.mark.parametrize("net", ["fcrn", "bts", "vnl", "midas", "midas_ongrid"])
def test_oracle_matches_the_reference_off_the_grid(net, golden):
    g = golden("offgrid")
    torch.set_num_threads(8)
    if net == "fcrn":
        _, oracle, _, tgt, _ = S.fcrn(S.FCRN_SEEDS[0])
        tag = "fcrn_s%d" % S.FCRN_SEEDS[0]
    elif net == "midas_ongrid":
        _, oracle, _, tgt, _ = S.midas(offgrid=False)
        tag = net
    else:
        _, oracle, _, tgt, _ = getattr(S, net)()
        tag = net
    a = _check(tag, oracle, tgt, g)
    # what 16-bit WEIGHTS alone do to the fp32 oracle's AbsRel on this state (nothing on the on-grid one)
    a_w = float(OM.compute(oracle(wq=nets.bf16_round), tgt)["absrel"])
    print("%s: AbsRel %.6f; the oracle with its weights rounded to bf16: shift %.2e" % (tag, a, abs(a_w - a)))
    if net == "midas_ongrid":
        assert a_w == a
    else:
        assert a_w != a
