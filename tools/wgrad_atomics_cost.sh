#!/bin/bash
# Upper bound of what the weight-gradient kernel's fp32-atomic epilogue costs: per-shape in-network time with the epilogue skipped.
cd "$(dirname "$0")/.."
for ns in 0 1; do
  MDE_WGRAD_NOSTORE=$ns python bench.py --steps 6 --warmup 2 --no-cpu-baseline --per-shape 2> gpurun_out/wa_$ns.txt > gpurun_out/wa_$ns.json
done
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"(conv_wgrad_tn)\s+(.*?)\s+x(\d+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+([\d.]+) ms/step", l)
        if m: d[m.group(2)]=(int(m.group(3)), float(m.group(4)))
    return d
a,b=load("gpurun_out/wa_0.txt"),load("gpurun_out/wa_1.txt")
ta=tb=0
for k,(n,us) in sorted(a.items(), key=lambda kv: -(kv[1][0]*(kv[1][1]-b[kv[0]][1]))):
    print("%-46s x%-2d with %7.1f  without %7.1f  (%4.1f%%)"%(k,n,us,b[k][1],100*(us-b[k][1])/us))
    ta+=n*us; tb+=n*b[k][1]
print("total us/step: with %.0f without %.0f"%(ta,tb))
PY
