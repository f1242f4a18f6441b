#!/usr/bin/env python3
"""SQ counters of the GEMM kernels per template instance, from the two PMC passes of tools/profile_r03.sh:
  python tools/sq_summary.py gpurun_out/prof_r03 > profiles/r03_pmc_gemm_in_network.txt
Per kernel form: MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8) -- per SIMD, as the round-2 review
normalised it --, waves waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES, LDS bank conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import glob
import re
import sys


def form(name):
    m = re.search(r"conv_gemm_nt<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        bp, bc, nt, dma, nbuf = a[0], a[1], a[2], a[3], a[4]
        pp = len(a) > 5 and a[5] == "true"
        halo = len(a) > 6 and a[6] == "true"
        red = a[7] if len(a) > 7 else "0"
        tag = "halo " if halo else "ping-pong " if pp else "single-buffer " if nbuf == "1" else "%s-deep ring " % nbuf if nbuf not in ("2",) else ""
        return "conv_gemm_nt %s%sx%s%s" % (tag, bp, bc, {"0": "", "1": " + BatchNorm-backward sums", "2": " + sums of a join"}[red])
    if "conv_wgrad_tn" in name:
        return "conv_wgrad_tn"
    if "wgrad_reduce_k" in name:
        return "wgrad_reduce_k"
    return None


def main():
    root = sys.argv[1]
    tot = collections.defaultdict(collections.Counter)
    for d in ("pmc_sq1", "pmc_sq2"):
        f = glob.glob(root + "/" + d + "/**/*counter_collection.csv", recursive=True)[0]
        for r in csv.DictReader(open(f)):
            k = form(r["Kernel_Name"])
            if k:
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("%-62s %10s %10s %14s %12s" % ("kernel form", "MFMA busy", "waiting", "LDS conflicts", "GPU cycles"))
    for k in sorted(tot, key=lambda k: -tot[k]["GRBM_GUI_ACTIVE"]):
        c = tot[k]
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / max(c["GRBM_GUI_ACTIVE"] / 8.0, 1.0)
        print("%-62s %9.1f%% %9.1f%% %13.1f%% %12.3g" % (k, 100 * busy, 100 * c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
                                                     100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), c["GRBM_GUI_ACTIVE"]))
    print()
    for k in sorted(tot):
        print(k + ": " + "  ".join("%s %.4g" % kv for kv in sorted(tot[k].items())))


if __name__ == "__main__":
    main()
