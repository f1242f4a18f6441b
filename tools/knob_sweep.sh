#!/bin/bash
# bench.py throughput under a list of environment settings, REPS alternating rounds.   tools/knob_sweep.sh "A=1" "B=2 C=3" ...
cd "$(dirname "$0")/.."
REPS=${REPS:-2}
for rep in $(seq $REPS); do
  for cfg in "" "$@"; do
    v=$(env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-launch-timing 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "[$cfg] $v"
  done
done
