#!/bin/bash
# A/B of the ping-pong (MDE_CONV_PP=1) vs lockstep (0) 8-wave conv loop: microbench shapes, then bench.py.
set -e
cd "$(dirname "$0")/.."
for shape in "60 80 32 256 256 3" "30 40 32 512 512 3" "120 160 32 128 128 3" "15 20 32 1024 1024 3" "60 80 32 512 256 1" "120 160 32 256 256 1"; do
  for pp in 0 1; do
    echo -n "PP=$pp  "
    MDE_CONV_PP=$pp MDE_CONV_TILE=256x256 python tools/conv_microbench.py fwd $shape 30
  done
done
for pp in 0 1; do
  echo "bench PP=$pp"
  MDE_CONV_PP=$pp python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['roofline']['achieved'], j['roofline']['frac'])"
done
