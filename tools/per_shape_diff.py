#!/usr/bin/env python3
"""Compare two `bench.py --per-shape` tables (stderr of the runs): per GEMM shape, time per launch in A and in B.
    python tools/per_shape_diff.py gpurun_out/a.txt gpurun_out/b.txt [--kind conv_gemm_nt]"""
import re
import sys


def load(path):
    tab = {}
    for line in open(path):
        m = re.match(r"(conv_gemm_nt|conv_wgrad_tn)\s+(.*?)\s+x(\d+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+([\d.]+) ms/step", line)
        if m:
            k, desc, cnt, us, tf, ms = m.groups()
            tab[(k, " ".join(desc.split()))] = (int(cnt), float(us), float(tf))
    return tab


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    kind = sys.argv[sys.argv.index("--kind") + 1] if "--kind" in sys.argv else None
    rows = []
    for key in a:
        if key in b and (kind is None or key[0] == kind):
            n, ua, tfa = a[key]
            _, ub, tfb = b[key]
            rows.append((n * (ub - ua) / 1e3, key, n, ua, ub, tfa, tfb))
    rows.sort()
    tot_a = sum(r[2] * r[3] for r in rows) / 1e3
    tot_b = sum(r[2] * r[4] for r in rows) / 1e3
    for dms, key, n, ua, ub, tfa, tfb in rows:
        print("%-14s %-44s x%-2d %7.1f -> %7.1f us (%+5.1f%%)  %6.0f -> %6.0f TF/s  %+6.3f ms/step" % (
            key[0], key[1], n, ua, ub, 100.0 * (ub / ua - 1.0), tfa, tfb, dms))
    print("total %.2f -> %.2f ms/step; sum of wins %.2f, sum of losses %.2f" % (
        tot_a, tot_b, sum(r[0] for r in rows if r[0] < 0), sum(r[0] for r in rows if r[0] > 0)))


if __name__ == "__main__":
    main()
