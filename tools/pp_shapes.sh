#!/bin/bash
cd "$(dirname "$0")/.."
for pp in 0 1; do
  MDE_CONV_PP=$pp python bench.py --steps 6 --warmup 2 --no-cpu-baseline --per-shape 2> gpurun_out/shape_pp$pp.txt > gpurun_out/bench_pp$pp.json
done
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"(conv_gemm_nt)\s+(.*?)\s+x(\d+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+([\d.]+) ms/step", l)
        if m: d[m.group(2)]=(int(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6)))
    return d
a,b=load("gpurun_out/shape_pp0.txt"),load("gpurun_out/shape_pp1.txt")
tot=0
rows=[]
for k in a:
    if k in b:
        rows.append((b[k][3]-a[k][3], k, a[k], b[k]))
rows.sort()
for dlt,k,x,y in rows[:12]+rows[-12:]:
    print("%+.3f ms  %-44s x%-2d  %7.1f -> %7.1f us  (%6.1f -> %6.1f TF/s)"%(dlt,k,x[0],x[1],y[1],x[2],y[2]))
print("sum delta ms/step: %.3f"%sum(r[0] for r in rows))
PY
