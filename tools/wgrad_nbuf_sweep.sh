#!/bin/bash
# In-network per-shape timing of the weight-gradient kernel with its 2-deep ring (default) and the single-buffer form.
cd "$(dirname "$0")/.."
for nb in 2 1; do
  MDE_WGRAD_NBUF=$nb python bench.py --steps 6 --warmup 2 --no-cpu-baseline --per-shape 2> gpurun_out/wn_$nb.txt > gpurun_out/wn_$nb.json || { tail -5 gpurun_out/wn_$nb.txt; exit 1; }
  cut -c1-130 gpurun_out/wn_$nb.json
done
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"(conv_wgrad_tn)\s+(.*?)\s+x(\d+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+([\d.]+) ms/step", l)
        if m: d[m.group(2)]=(int(m.group(3)), float(m.group(4)))
    return d
a,b=load("gpurun_out/wn_2.txt"),load("gpurun_out/wn_1.txt")
ta=tb=best=0
for k,(n,us) in sorted(a.items()):
    print("%-46s x%-2d ring %7.1f  single %7.1f"%(k,n,us,b[k][1]))
    ta+=n*us; tb+=n*b[k][1]; best+=n*min(us,b[k][1])
print("total us/step: ring %.0f single %.0f best-of %.0f"%(ta,tb,best))
PY
