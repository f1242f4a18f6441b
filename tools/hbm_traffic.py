#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes over bench.py into profiles/r03_hbm_traffic.json (tagged with the hash of the kernel
sources it was collected on: bench.py quotes it only for those).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-timing
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py ... (same)
  python tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write <steps incl. warm-up> <conv calls/step> <wgrad calls/step> > profiles/r03_hbm_traffic.json

Counters are in KiB; FETCH_SIZE is doubled on gfx950 (it tallies 128-byte read requests as 64 bytes,
MI355X_MICROARCH.md, HBM/rocprofv3 section); WRITE_SIZE is exact.  Bytes are reported per conv CALL (the
unit bench.py times: a call is two kernel launches when the pixel range is split into full rounds + tail)."""
import collections
import csv
import glob
import json
import sys


def totals(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for k, names in (("conv_gemm_nt", ("conv_gemm_nt",)), ("conv_wgrad_tn", ("conv_wgrad_tn", "conv_wgrad_win", "wgrad_reduce_k"))):
            if any(nm in r["Kernel_Name"] for nm in names):     # (the windowed kernel and the two-stage reduction's second kernel count towards the weight-gradient call)
                tot[k] += float(r["Counter_Value"])
                n[k] += 1
    return tot, n


def main():
    fetch_dir, write_dir, steps, cg, cw = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    ft, fn = totals(fetch_dir, "FETCH_SIZE")
    wt, _ = totals(write_dir, "WRITE_SIZE")
    out = {}
    for k, calls_per_step in (("conv_gemm_nt", cg), ("conv_wgrad_tn", cw)):
        calls = calls_per_step * steps
        raw = ft[k] * 1024 / calls
        wr = wt[k] * 1024 / calls
        out[k] = {"kernel_launches_profiled": fn[k], "layer_calls_profiled": calls,
                  "fetch_size_bytes_per_call_raw": round(raw), "fetch_bytes_per_call_corrected_x2": round(2 * raw),
                  "write_bytes_per_call": round(wr), "hbm_bytes_per_launch": round(2 * raw + wr)}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps %d --warmup 1` "
                    "(batch 32); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests as 64 B); "
                    "WRITE_SIZE exact; counters are in KiB. Bytes are per conv CALL (= bench.py launch unit). "
                    "Produced by tools/hbm_traffic.py." % (steps - 1))
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["_source_hash"] = bench.kernel_source_hash()
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
