"""Diagnostic: the fused sums of the 8-wave 64-column launch beside a weight-gradient co-runner -- WHICH workgroup's addend is
wrong, in which channels, and by what (against the addends recomputed on the host per workgroup, per half of its rows and per
row lane).  MDE_CONV_DEEP_WAVES=64 (or d64) selects the form; N / H / W from the environment (default 4 x 64 x 80)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mono_depth_estimation_amd import ops  # noqa: E402
torch.manual_seed(0)
ACT = ops.ACT_DTYPE
N, H, W = int(os.environ.get("N", 4)), int(os.environ.get("H", 64)), int(os.environ.get("W", 80))
C = O = 64
MASK = os.environ.get("MASK", "1") == "1"
RUNS = int(os.environ.get("RUNS", 4))
x = torch.randn(N, H, W, C, device="cuda").to(ACT)
w = (torch.randn(O, 9, C, device="cuda") * 0.05).to(ACT)
d = ops.fwd_desc(N, H, W, C, C, x.numel() * 2, 3, 1, 1, O, O)
side = torch.cuda.Stream()
wx = torch.randn(8, 128, 160, 64, device="cuda").to(ACT)
wdy = torch.randn(8, 128, 160, 64, device="cuda").to(ACT)
wdw = torch.zeros(64, 9, 64, device="cuda")
wd = ops.conv_wgrad_desc(8, 128, 160, 64, 64, wx.numel() * 2, 128, 160, 64, 64, wdy.numel() * 2, 3, 1, 1, 14)
sx = torch.randn(N, H, W, O, device="cuda").to(ACT)
mean, rstd = torch.randn(O, device="cuda") * 0.1, torch.rand(O, device="cuda") + 0.5
msc, msh = torch.rand(O, device="cuda") + 0.5, torch.randn(O, device="cuda") * 0.1


def run(busy):
    out = torch.empty(N, H, W, O, dtype=ACT, device="cuda")
    part = ops.new_stat_buffer(O)
    if busy:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(6):
                ops.conv_wgrad(wd, wdy, wx, wdw)
    red = ops.bn_red(sx, mean, rstd, part, mask_scale=msc, mask_shift=msh) if MASK else ops.bn_red(sx, mean, rstd, part)
    ops.conv_gemm(d, x, w, out, red=red)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return out, part.clone()


o0, p0 = run(False)
M = N * H * W
nwg = (M + 127) // 128
g = o0.float().reshape(M, O)
sxf = sx.float().reshape(M, O)
on = ((sxf * msc + msh) > 0) if MASK else torch.ones_like(sxf, dtype=torch.bool)
ge = torch.where(on, g, torch.zeros_like(g))
ge2 = ge * ((sxf - mean) * rstd)
pad = nwg * 128 - M
if pad:
    ge = torch.cat([ge, ge.new_zeros(pad, O)])
    ge2 = torch.cat([ge2, ge2.new_zeros(pad, O)])
T1 = ge.reshape(nwg, 2, 64, O)        # [workgroup][half: rows rl / rl + 64][row lane rl][channel]
T2 = ge2.reshape(nwg, 2, 64, O)
S1, S2 = T1.sum((1, 2)), T2.sum((1, 2))
slots = p0.shape[0]
exp1 = torch.zeros(slots, O, device="cuda").index_add_(0, torch.arange(nwg, device="cuda") % slots, S1)
exp2 = torch.zeros(slots, O, device="cuda").index_add_(0, torch.arange(nwg, device="cuda") % slots, S2)
print("%d workgroups, %d slots; quiet run vs host: sum1 max |err| %.3g, sum2 %.3g" % (
    nwg, slots, float((p0[:, 0] - exp1).abs().max()), float((p0[:, 1] - exp2).abs().max())))
for it in range(RUNS):
    o, p = run(True)
    print("busy run %d: output identical: %s" % (it, torch.equal(o, o0)))
    for which, exp, T, S in ((0, exp1, T1, S1), (1, exp2, T2, S2)):
        E = p[:, which] - exp                                    # [slot][channel]
        tol = 2e-3 * (exp.abs() + 1) + 1e-2
        bad = (E.abs() > tol).nonzero().tolist()
        if not bad:
            print("   sum%d: clean" % (which + 1))
            continue
        bs = sorted(set(b[0] for b in bad))
        print("   sum%d: wrong in slots %s" % (which + 1, bs))
        for k in bs:
            ch = [b[1] for b in bad if b[0] == k]
            print("     slot %d: channels %s" % (k, ch))
            print("       error      : %s" % [round(float(E[k, c]), 3) for c in ch[:12]])
            for pi in range(k, nwg, slots):
                r = [round(float(E[k, c] / (S[pi, c] if abs(float(S[pi, c])) > 1e-6 else 1.0)), 3) for c in ch[:12]]
                # the addend of the half-tiles and of single row lanes for the first wrong channel
                c0 = ch[0]
                halves = [round(float(T[pi, h, :, c0].sum()), 3) for h in (0, 1)]
                print("       workgroup %3d: error / its addend %s; addend of channel %d by half %s, whole %.3f" % (
                    pi, r, c0, halves, float(S[pi, c0])))
            # is the error the (negated) addend of a group of row lanes of one workgroup?  8 row lanes = one wave
            c0 = ch[0]
            for pi in range(k, nwg, slots):
                wv = T[pi, :, :, c0].reshape(2, 8, 8).sum(2)      # [half][wave]
                print("       workgroup %3d channel %d by (half, wave): %s" % (pi, c0, [[round(float(v), 2) for v in row] for row in wv]))


# ---- which wave, and what happened to its register?  (one workgroup per slot needed: N=1 H=64 W=64)
if nwg <= slots:
    print("---- hypothesis search: the error over the 8 chunks of one element index against one wave's addends")
    for it in range(RUNS * 2):
        o, p = run(True)
        for which, exp, T in ((0, exp1, T1), (1, exp2, T2)):
            E = p[:, which] - exp
            tol = 2e-3 * (exp.abs() + 1) + 1e-2
            for k in range(nwg):
                for e in range(8):
                    ev = E[k, e::8]                                   # [chunk]
                    if not bool((ev.abs() > tol[k, e::8]).any()):
                        continue
                    A = T[k, :, :, e::8].reshape(2, 8, 8, 8).sum(2)   # [half][wave][chunk]
                    best = None
                    for wv in range(8):
                        cands = {"-row0": -A[0, wv], "-row1": -A[1, wv], "-both": -(A[0, wv] + A[1, wv]), "+row0": A[0, wv], "+row1": A[1, wv],
                                 "+both": A[0, wv] + A[1, wv]}
                        for name, v in cands.items():
                            r = float((ev - v).abs().max())
                            if best is None or r < best[0]:
                                best = (r, wv, name)
                    print("run %d sum%d workgroup %d element %d: error %s; best fit %s of wave %d, residual %.3g" % (
                        it, which + 1, k, e, [round(float(v), 3) for v in ev], best[2], best[1], best[0]))


# ---- single addends: is each wrong channel off by exactly one row's addend (dropped / doubled)?
if nwg <= slots:
    print("---- single-addend search (sum1): error of a channel against +- the addend of one row of the workgroup")
    for it in range(RUNS * 3):
        o, p = run(True)
        E = p[:, 0] - exp1
        tol = 2e-3 * (exp1.abs() + 1) + 1e-2
        for k, c in (E.abs() > tol).nonzero().tolist():
            col = T1[k, :, :, c].reshape(128)                     # row r = half * 64 + rl
            err = float(E[k, c])
            hits = [("-" if s < 0 else "+") + "r%d" % r for s in (-1, 1) for r in range(128) if abs(err - s * float(col[r])) < 1.5e-3 * (abs(err) + 1)]
            # pairs of rows of ONE thread (rl, rl + 64)
            hits2 = [("-" if s < 0 else "+") + "t%d" % r for s in (-1, 1) for r in range(64) if abs(err - s * float(col[r] + col[r + 64])) < 1.5e-3 * (abs(err) + 1)]
            print("run %d workgroup %d channel %d (chunk %d, element %d): error %.4f; one row: %s; one thread's two rows: %s" % (
                it, k, c, c // 8, c % 8, err, hits, hits2))
