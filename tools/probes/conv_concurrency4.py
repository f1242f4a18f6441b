"""Diagnostic: the OUTPUT of one conv launch repeated beside a co-runner of another stream -- how often does it differ from the
quiet run, and where (tile, rows, channels)?  tests/test_co_residency_gpu.py saw one such difference in 48 busy launches of the
4-wave deep-ring tile.  FORMS (comma list of MDE_CONV_DEEP_WAVES values), MODES (plain,stats,red), RUNS from the environment."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mono_depth_estimation_amd import ops  # noqa: E402
torch.manual_seed(0)
ACT = ops.ACT_DTYPE
N, H, W, C, O = 4, 64, 80, 64, 64
RUNS = int(os.environ.get("RUNS", 200))
CO = os.environ.get("CO", "wgrad")
x = torch.randn(N, H, W, C, device="cuda").to(ACT)
w = (torch.randn(O, 9, C, device="cuda") * 0.05).to(ACT)
d = ops.fwd_desc(N, H, W, C, C, x.numel() * 2, 3, 1, 1, O, O)
side = torch.cuda.Stream()
wx = torch.randn(8, 128, 160, 64, device="cuda").to(ACT)
wdy = torch.randn(8, 128, 160, 64, device="cuda").to(ACT)
wdw = torch.zeros(64, 9, 64, device="cuda")
wd = ops.conv_wgrad_desc(8, 128, 160, 64, 64, wx.numel() * 2, 128, 160, 64, 64, wdy.numel() * 2, 3, 1, 1, 14)
cout = torch.empty(8, 128, 160, 64, dtype=ACT, device="cuda")
cd = ops.fwd_desc(8, 128, 160, 64, 64, wx.numel() * 2, 3, 1, 1, 64, 64)
sx = torch.randn(N, H, W, O, device="cuda").to(ACT)
mean, rstd = torch.randn(O, device="cuda") * 0.1, torch.rand(O, device="cuda") + 0.5
msc, msh = torch.rand(O, device="cuda") + 0.5, torch.randn(O, device="cuda") * 0.1
out = torch.empty(N, H, W, O, dtype=ACT, device="cuda")          # ONE output buffer, poisoned before every launch


FRESH = os.environ.get("FRESH", "0") == "1"            # a new output tensor per launch, as tests/test_co_residency_gpu.py has


def run(mode, busy):
    global out
    if FRESH:
        out = torch.empty(N, H, W, O, dtype=ACT, device="cuda")
    out.fill_(float("nan"))
    part = ops.new_stat_buffer(O)
    if busy:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(6):
                if CO == "wgrad":
                    ops.conv_wgrad(wd, wdy, wx, wdw)
                else:
                    ops.conv_gemm(cd, wx, w, cout)
    if mode == "plain":
        ops.conv_gemm(d, x, w, out)
    elif mode == "stats":
        ops.conv_gemm(d, x, w, out, part)
    else:
        ops.conv_gemm(d, x, w, out, red=ops.bn_red(sx, mean, rstd, part, mask_scale=msc, mask_shift=msh))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return out.clone()


for form in os.environ.get("FORMS", "4,64").split(","):
    os.environ["MDE_CONV_DEEP_WAVES"] = form
    for mode in os.environ.get("MODES", "plain,red").split(","):
        ref = run(mode, False)
        assert bool(torch.isfinite(ref.float()).all())
        quiet_bad = sum(int(not torch.equal(run(mode, False), ref)) for _ in range(20))
        bad = 0
        for it in range(RUNS):
            o = run(mode, True)
            if torch.equal(o, ref):
                continue
            bad += 1
            if bad <= 4:
                df = (o.float() != ref.float()) | torch.isnan(o.float())
                idx = df.reshape(-1, O).nonzero()
                px, ch = idx[:, 0], idx[:, 1]
                tiles = sorted(set((px // 128).tolist()))
                rows = sorted(set((px % 128).tolist()))
                chs = sorted(set(ch.tolist()))
                a, b = o.float().reshape(-1, O)[px, ch], ref.float().reshape(-1, O)[px, ch]
                print("   form %s %s run %d: %d elements differ (%d NaN left); tiles %s; rows in tile %s; channels %s" % (
                    form, mode, it, int(df.sum()), int(torch.isnan(o.float()).sum()), tiles[:6], rows[:40], chs[:40]))
                print("      got %s\n      ref %s" % ([round(float(v), 3) for v in a[:8]], [round(float(v), 3) for v in b[:8]]))
        print("form %s, %s, co-runner %s: output differs in %d of %d busy launches (and in %d of 20 quiet ones)" % (form, mode, CO, bad, RUNS, quiet_bad))
