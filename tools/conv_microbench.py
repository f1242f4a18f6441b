#!/usr/bin/env python3
"""Microbenchmark of single conv_gemm / conv_wgrad shapes (for rocprofv3 --pmc runs and A/B
timing in one process).   python tools/conv_microbench.py [fwd|wgrad] M_h M_w N Cin Cout k [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_depth_estimation_amd import ops  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    H, W, N, Cin, Cout, k = (int(v) for v in sys.argv[2:8]) if len(sys.argv) >= 8 else (60, 80, 32, 256, 256, 3)
    reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
    dil = int(os.environ.get("MB_DIL", "1"))          # atrous: padding = dilation * (k // 2)
    pad = dil * (k // 2)
    if kind == "fwd":
        w = (torch.randn(Cout, k * k, Cin, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device=dev)
        d = ops.fwd_desc(N, H, W, Cin, Cin, x.numel() * 2, k, 1, pad, Cout, Cout, dil=dil)
        stats = ops.new_stat_buffer(Cout) if os.environ.get("MB_STATS") else None   # BN partial sums in the epilogue
        fn = lambda: ops.conv_gemm(d, x, w, out, stats)
        flops = 2.0 * N * H * W * Cout * k * k * Cin
    else:
        dy = torch.randn(N, H, W, Cout, device=dev).to(torch.bfloat16)
        dw = torch.zeros(Cout, k * k, Cin, device=dev)
        rt = Cout // (128 if Cout % 128 == 0 else 64)
        ct = Cin // (128 if Cin % 128 == 0 else 64)
        ks = int(os.environ["MB_KS"]) if os.environ.get("MB_KS") else ops.choose_ksplit(
            N * H * W, rt, ct, k * k, tile_elems=(Cout // rt) * (Cin // ct), wg_per_cu=min(8, 160 * 1024 // (256 * (Cout // rt + Cin // ct) + 512)))
        print("ksplit", ks, end="  ")
        d = ops.conv_wgrad_desc(N, H, W, Cin, Cin, x.numel() * 2, H, W, Cout, Cout, dy.numel() * 2, k, 1, pad, ks, dil=dil)
        fn = lambda: ops.conv_wgrad(d, dy, x, dw)
        flops = 2.0 * N * H * W * Cout * k * k * Cin
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%s M=%d N=%d K=%d: %.1f us  %.1f TF/s" % (kind, N * H * W, Cout, k * k * Cin, us, flops / us / 1e6))


if __name__ == "__main__":
    main()
