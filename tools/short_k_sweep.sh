#!/bin/bash
# In-network timing of the 1x1 convolutions under forced tiles (the step's other convs run on the forced tile too: only the
# per-shape lines of the 1x1 shapes are compared).
cd "$(dirname "$0")/.."
for t in default 128x128 128x128s 128x256s 256x256; do
  if [ $t = default ]; then env=""; else env="MDE_CONV_TILE=$t"; fi
  env $env python bench.py --steps 4 --warmup 2 --no-cpu-baseline --per-shape 2> gpurun_out/sk_$t.txt > gpurun_out/sk_$t.json || { tail -5 gpurun_out/sk_$t.txt; exit 1; }
done
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"(conv_gemm_nt)\s+(.*?)\s+x(\d+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+([\d.]+) ms/step", l)
        if m: d[m.group(2)]=(int(m.group(3)), float(m.group(4)))
    return d
tiles=["default","128x128","128x128s","128x256s","256x256"]
T={t:load("gpurun_out/sk_%s.txt"%t) for t in tiles}
print("%-46s %4s "%("shape","x")+" ".join("%9s"%t for t in tiles))
tot={t:0.0 for t in tiles}; best=0.0
for k,(n,us) in sorted(T["default"].items()):
    if "taps=1 " not in k or " N=64 " in k: continue
    row=[T[t].get(k,(0,float('nan')))[1] for t in tiles]
    print("%-46s x%-3d "%(k,n)+" ".join("%9.1f"%v for v in row))
    for t,v in zip(tiles,row): tot[t]+=n*v
    best+=n*min(row)
print("total us/step: "+"  ".join("%s %.0f"%(t,tot[t]) for t in tiles)+"  best-of %.0f"%best)
PY
