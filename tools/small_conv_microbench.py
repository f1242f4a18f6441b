#!/usr/bin/env python3
"""Timing of the stem (7x7/2 on the fp32 image) and head (3x3 -> fp32) kernels at the bench shape.
python tools/small_conv_microbench.py [N H W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_depth_estimation_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 480, 640)
    dev = "cuda"
    x = torch.rand(N, 3, H, W, device=dev)
    w = torch.randn(64, 7, 7, 3, device=dev) * 0.05
    OH, OW = H // 2, W // 2
    out = torch.empty(N, OH, OW, 64, dtype=torch.bfloat16, device=dev)
    part = ops.new_stat_buffer(64)
    dy = torch.randn(N, OH, OW, 64, device=dev).to(torch.bfloat16)
    dw = torch.zeros(64, 7, 7, 3, device=dev)
    mb_in, mb_out = x.numel() * 4 / 1e6, out.numel() * 2 / 1e6
    us = timeit(lambda: ops.stem_conv_fwd(x, w, out, part))
    print("stem fwd   %8.1f us  %6.2f TB/s (image + output once)" % (us, (mb_in + mb_out) / us))
    us = timeit(lambda: ops.stem_conv_wgrad(x, dy, dw))
    print("stem wgrad %8.1f us  %6.2f TB/s" % (us, (mb_in + mb_out) / us))
    # head: 3x3, 64 -> 1 on the [N][H/2][W/2][64] feature map
    f = torch.randn(N, OH, OW, 64, device=dev).to(torch.bfloat16)
    hw = torch.randn(1, 3, 3, 64, device=dev) * 0.05
    logits = torch.empty(N, OH, OW, 1, device=dev)
    dl = torch.randn(N, OH, OW, 1, device=dev)
    df = torch.empty_like(f)
    dhw = torch.zeros_like(hw)
    us = timeit(lambda: ops.head_conv_fwd(f, hw, logits, N, OH, OW, 64, 1))
    print("head fwd   %8.1f us  %6.2f TB/s" % (us, mb_out / us))
    us = timeit(lambda: ops.head_conv_bwd(f, hw, dl, df, None, N, OH, OW, 64, 1))
    print("head dgrad %8.1f us  %6.2f TB/s" % (us, mb_out / us))
    us = timeit(lambda: ops.head_conv_bwd(f, hw, dl, None, dhw, N, OH, OW, 64, 1))
    print("head wgrad %8.1f us  %6.2f TB/s" % (us, mb_out / us))


if __name__ == "__main__":
    main()
