#!/bin/bash
cd "$(dirname "$0")/.."
for t in default 128x64s; do
  if [ $t = default ]; then env=""; else env="MDE_CONV_TILE=$t"; fi
  env $env python bench.py --steps 4 --warmup 2 --no-cpu-baseline --per-shape 2> gpurun_out/t64_$t.txt > gpurun_out/t64_$t.json || { tail -5 gpurun_out/t64_$t.txt; exit 1; }
done
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"(conv_gemm_nt)\s+(.*?)\s+x(\d+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+([\d.]+) ms/step", l)
        if m: d[m.group(2)]=(int(m.group(3)), float(m.group(4)))
    return d
a,b=load("gpurun_out/t64_default.txt"),load("gpurun_out/t64_128x64s.txt")
ta=tb=best=0
for k,(n,us) in sorted(a.items()):
    if " N=64 " in k: continue
    flag = "<--" if b[k][1] < 0.97*us else ""
    print("%-46s x%-2d default %7.1f  128x64s %7.1f %s"%(k,n,us,b[k][1],flag))
    ta+=n*us; tb+=n*b[k][1]; best+=n*min(us,b[k][1])
print("total us/step: default %.0f  128x64s %.0f  best-of %.0f"%(ta,tb,best))
PY
