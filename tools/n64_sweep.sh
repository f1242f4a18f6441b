#!/bin/bash
cd "$(dirname "$0")/.."
for t in default 128x64s; do
  if [ $t = default ]; then env=""; else env="MDE_CONV_TILE=$t"; fi
  env $env python bench.py --steps 4 --warmup 2 --no-cpu-baseline --per-shape 2> gpurun_out/n64_$t.txt > gpurun_out/n64_$t.json || { tail -5 gpurun_out/n64_$t.txt; exit 1; }
  grep "conv_gemm_nt" gpurun_out/n64_$t.txt | grep " N=64 " 
  echo ---
done
