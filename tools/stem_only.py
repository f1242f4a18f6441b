#!/usr/bin/env python3
"""One kernel under rocprofv3 --pmc: python tools/stem_only.py [fwd|wgrad] (stem conv at 32x3x480x640)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_depth_estimation_amd import ops  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
N, H, W = 32, 480, 640
x = torch.rand(N, 3, H, W, device="cuda")
w = torch.randn(64, 7, 7, 3, device="cuda") * 0.05
out = torch.empty(N, H // 2, W // 2, 64, dtype=torch.bfloat16, device="cuda")
part = ops.new_stat_buffer(64)
dy = torch.randn(N, H // 2, W // 2, 64, device="cuda").to(torch.bfloat16)
dw = torch.zeros(64, 7, 7, 3, device="cuda")
for _ in range(8):
    if kind == "fwd":
        ops.stem_conv_fwd(x, w, out, part)
    else:
        ops.stem_conv_wgrad(x, dy, dw)
torch.cuda.synchronize()
