#!/bin/bash
cd "$(dirname "$0")/.."
for r in 0 0.95 1.05 1.15 1.3 1.6; do
  echo -n "rate_s=$r  fcrn: "
  MDE_CONV_RATE_S=$r python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])"
done
for r in 0 1.05 1.3; do
  echo "rate_s=$r"
  MDE_CONV_RATE_S=$r python tools/config_bench.py vnl midas dorn 2>/dev/null | grep config | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['config'][:30], d['ms_per_step'])"
done
