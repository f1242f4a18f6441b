#!/usr/bin/env python3
"""Cold-operand timing (and, with the -DMDE_SB_STAMP diagnostic library, per-phase cycle stamps) of one convolution shape on the
single-buffer 128-pixel tiles, plain against halo-tiled.  Operands rotate over enough tensor sets to exceed the 256 MB
Infinity Cache, as in the network, where no layer finds its input in cache.
   python tools/halo_probe.py H W N Cin Cout k [reps]
   MDE_LIB_PATH=tools/probes/libmde_stamp.so python tools/halo_probe.py ...      (stamps; build: tools/pp_stamp.py's recipe
   with -DMDE_SB_STAMP)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_depth_estimation_amd import ops  # noqa: E402

H, W, N, Cin, Cout, k = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 12
dil = int(os.environ.get("MB_DIL", "1"))
per_set = N * H * W * (Cin + Cout) * 2
nsets = max(2, -(-600_000_000 // per_set))
xs = [torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16) for _ in range(nsets)]
outs = [torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device="cuda") for _ in range(nsets)]
w = (torch.randn(Cout, k * k, Cin, device="cuda") * 0.05).to(torch.bfloat16)
d = ops.fwd_desc(N, H, W, Cin, Cin, xs[0].numel() * 2, k, 1, dil * (k // 2), Cout, Cout, dil=dil)
stats = ops.new_stat_buffer(2 * Cout)          # the stamps land behind the [32][2][Cout] partial sums the kernel itself writes
flops = 2.0 * N * H * W * Cout * k * k * Cin
names = ["top barrier", "DMA issue", "wait (window step)", "wait (weights step)", "barrier after wait", "reads + MFMA"]
for halo in ("0", "1", "2"):
    os.environ["MDE_CONV_HALO"] = halo
    for i in range(nsets):
        ops.conv_gemm(d, xs[i], w, outs[i], None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        ops.conv_gemm(d, xs[r % nsets], w, outs[r % nsets], None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("halo=%s  M=%d N=%d taps=%d C=%d: %.1f us  %.1f TF/s  (%d operand sets)" % (halo, N * H * W, Cout, k * k, Cin, us, flops / us / 1e6, nsets))
    if "stamp" in os.environ.get("MDE_LIB_PATH", ""):
        stats.zero_()
        ops.conv_gemm(d, xs[0], w, outs[0], stats)
        torch.cuda.synchronize()
        t = stats.view(-1)[64 * Cout:64 * Cout + 32].view(torch.int64).cpu().tolist()
        n = max(t[6], 1)
        print("   per K-step cycles (wave 0 of a mid-grid workgroup, %d steps): " % n + ", ".join("%s %.0f" % (nm, t[i] / n) for i, nm in enumerate(names)) +
              " | total %.0f" % (sum(t[:6]) / n))
        tot = t[7] + t[8] + t[9]
        print("   workgroup: prologue %d, K-loop %d, epilogue %d cycles = %.1f us at %.2f GHz (kernel %.1f us)" % (
            t[7], t[8], t[9], t[10] / 100.0, tot / max(t[10], 1) / 10.0, us))
