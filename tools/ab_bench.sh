#!/bin/bash
# A/B of one environment switch on the FCRN step, alternating runs on the same box:  tools/ab_bench.sh VAR A B [rounds] [bench args...]
# prints ms/step of every run; output under gpurun_out/ab_<VAR>.log
VAR=$1; A=$2; B=$3; R=${4:-2}; shift 4 || true
mkdir -p gpurun_out
LOG=gpurun_out/ab_${VAR}.log
: > $LOG
for r in $(seq 1 $R); do
  for v in $A $B; do
    echo "== $VAR=$v round $r" >> $LOG
    env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>>$LOG | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$VAR=$v', 'ms/step', d['ms_per_step'], 'img/s', d['value'], 'conv frac', r['frac'], 'conv us', r.get('avg_launch_us'), 'wgrad', r.get('wgrad_kernel',{}).get('achieved'))
" | tee -a $LOG
  done
done
