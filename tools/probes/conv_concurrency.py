"""Diagnostic: one conv launch repeated on the main stream while another stream keeps the chip busy -- outputs (and BatchNorm
statistics / fused backward sums) must not depend on what runs beside the launch.  MDE_CONV_DEEP_WAVES selects the form."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mono_depth_estimation_amd import ops  # noqa: E402

torch.manual_seed(0)
ACT = ops.ACT_DTYPE
N, H, W, C, O = 4, 64, 80, 64, 64
x = torch.randn(N, H, W, C, device="cuda").to(ACT)
w = (torch.randn(O, 9, C, device="cuda") * 0.05).to(ACT)
d = ops.fwd_desc(N, H, W, C, C, x.numel() * 2, 3, 1, 1, O, O)
side = torch.cuda.Stream()
big_a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
big_b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
# a BatchNorm site for the fused backward sums: its input, statistics and mask constants
sx = torch.randn(N, H, W, O, device="cuda").to(ACT)
mean, rstd = torch.randn(O, device="cuda") * 0.1, torch.rand(O, device="cuda") + 0.5
msc, msh = torch.rand(O, device="cuda") + 0.5, torch.randn(O, device="cuda") * 0.1


# co-runners for the other stream: a library GEMM, the weight-gradient kernel (64 x 64 tiles, as a 64 -> 64 3x3 layer's), a 4-wave conv
CO = os.environ.get("CO", "mm")
wx = torch.randn(8, 128, 160, 64, device="cuda").to(ACT)
wdy = torch.randn(8, 128, 160, 64, device="cuda").to(ACT)
wdw = torch.zeros(64, 9, 64, device="cuda")
wd = ops.conv_wgrad_desc(8, 128, 160, 64, 64, wx.numel() * 2, 128, 160, 64, 64, wdy.numel() * 2, 3, 1, 1, 14)
cx = torch.randn(8, 128, 160, 64, device="cuda").to(ACT)
cout = torch.empty(8, 128, 160, 64, dtype=ACT, device="cuda")
cd = ops.fwd_desc(8, 128, 160, 64, 64, cx.numel() * 2, 3, 1, 1, 64, 64)


def corun():
    if CO == "mm":
        for _ in range(3):
            torch.mm(big_a, big_b)
    elif CO == "wgrad":
        for _ in range(6):
            ops.conv_wgrad(wd, wdy, wx, wdw)
    else:
        for _ in range(6):
            ops.conv_gemm(cd, cx, w, cout)


def run(mode, busy):
    out = torch.empty(N, H, W, O, dtype=ACT, device="cuda")
    part = ops.new_stat_buffer(O)
    if busy:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            corun()
    if mode == "plain":
        ops.conv_gemm(d, x, w, out)
    elif mode == "stats":
        ops.conv_gemm(d, x, w, out, part)
    else:
        red = ops.bn_red(sx, mean, rstd, part, mask_scale=msc, mask_shift=msh)
        ops.conv_gemm(d, x, w, out, red=red)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return out.clone(), part.sum(0).clone()


for mode in ("plain", "stats", "red"):
    ref_o, ref_p = run(mode, False)
    bad_o = bad_p = 0
    worst = 0.0
    for it in range(30):
        o, p = run(mode, True)
        bad_o += int(not torch.equal(o, ref_o))
        if not torch.allclose(p, ref_p, rtol=1e-4, atol=1e-3):
            bad_p += 1
            worst = max(worst, float(((p - ref_p).abs() / (ref_p.abs() + 1e-3)).max()))
    print("[co-runner %s] %-6s: outputs differing from the quiet run in %d of 30 busy runs; sums differing (> 1e-4) in %d, worst relative %.3g" % (CO, mode, bad_o, bad_p, worst))
