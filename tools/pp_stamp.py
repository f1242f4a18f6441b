#!/usr/bin/env python3
"""Per-phase cycle stamps of the ping-pong conv loop (diagnostic build -DMDE_PP_STAMP, see conv_gemm.hip).  Build the diagnostic
library in-tree (gpurun ships in-tree .so files) and point the binding at it:
   cd mono_depth_estimation_amd/csrc && for f in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
       $([ $f = conv_gemm.hip ] && echo -DMDE_PP_STAMP) -c $f -o /tmp/st_${f%.hip}.o; done && \
       hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/libmde_stamp.so /tmp/st_*.o
   MDE_LIB_PATH=tools/probes/libmde_stamp.so python tools/pp_stamp.py H W N Cin Cout k"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_depth_estimation_amd import ops  # noqa: E402

H, W, N, Cin, Cout, k = (int(v) for v in sys.argv[1:7]) if len(sys.argv) >= 7 else (60, 80, 32, 256, 256, 3)
x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
w = (torch.randn(Cout, k * k, Cin, device="cuda") * 0.05).to(torch.bfloat16)
out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device="cuda")
d = ops.fwd_desc(N, H, W, Cin, Cin, x.numel() * 2, k, 1, k // 2, Cout, Cout)
stats = ops.new_stat_buffer(max(Cout, 64))
for _ in range(3):
    ops.conv_gemm(d, x, w, out, stats)
torch.cuda.synchronize()
stats.zero_()
ops.conv_gemm(d, x, w, out, stats)
torch.cuda.synchronize()
t = stats.view(-1)[:64].view(torch.int64).cpu().tolist()
for name, o in (("A (wave 0)", 0), ("B (wave 4)", 8)):
    r, rb, m, mb, v, n = t[o:o + 6]
    n = max(n, 1)
    print("%s: per K-step cycles: R body %.0f, barrier after R %.0f, M body %.0f, barrier after M %.0f, vmcnt wait %.0f | total %.0f (ideal MFMA-bound 2048)" % (
        name, r / n, rb / n, m / n, mb / n, v / n, (r + rb + m + mb + v) / n))
