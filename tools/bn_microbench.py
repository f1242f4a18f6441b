#!/usr/bin/env python3
"""Streaming rate of the BatchNorm passes on the step's large sites (operands of 79-314 MB: colder than the 256 MB MALL),
against torch's elementwise add on the same bytes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_depth_estimation_amd import ops  # noqa: E402


def t(f, k=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k


for M, C in ((614400, 256), (2457600, 64), (614400, 64), (153600, 512), (38400, 1024)):
    mk = lambda: torch.randn(M, C, device="cuda").to(torch.bfloat16)
    x, r, out, dout, dx, dres = mk(), mk(), mk(), mk(), mk(), mk()
    sc, sh, mean, rstd = (torch.rand(C, device="cuda") + 0.5 for _ in range(4))
    coef = torch.rand(3, C, device="cuda") * 0.01
    part = ops.new_stat_buffer(C, "cuda")
    bits = torch.empty(M * (C // 8), dtype=torch.uint8, device="cuda")
    nb = M * C * 2
    rows = []
    dt = t(lambda: ops.bn_apply(x, C, sc, sh, out, C, M, C, True)); rows.append(("apply", 2 * nb / dt))
    dt = t(lambda: ops.bn_apply(x, C, sc, sh, out, C, M, C, True, r=r, ldr=C, relu_bits=bits)); rows.append(("apply+res+bits", (3 * nb + M * C // 8) / dt))
    dt = t(lambda: ops.bn_bwd_reduce(dout, C, None, 0, x, C, mean, rstd, M, C, True, part, mask_scale=sc, mask_shift=sh)); rows.append(("bwd reduce", 2 * nb / dt))
    dt = t(lambda: ops.bn_bwd_apply(dout, C, None, 0, x, C, mean, rstd, coef, M, C, True, dx, C, mask_scale=sc, mask_shift=sh)); rows.append(("bwd apply", 3 * nb / dt))
    dt = t(lambda: ops.bn_bwd_apply(dout, C, None, 0, x, C, mean, rstd, coef, M, C, True, dx, C, dres=dres, ldres=C, relu_bits=bits)); rows.append(("bwd apply+dres", (4 * nb + M * C // 8) / dt))
    dt = t(lambda: torch.add(x, r, out=out)); rows.append(("torch add", 3 * nb / dt))
    print("M=%d C=%d (%.0f MB/operand): " % (M, C, nb / 1e6) + "  ".join("%s %.2f" % (n, v / 1e12) for n, v in rows) + "  TB/s")
