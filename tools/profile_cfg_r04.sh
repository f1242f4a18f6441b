#!/bin/bash
# rocprofv3 kernel-time summary of `bench.py --config $1` (the driver's own entry point for the other configurations):
# per-kernel totals, launches per step, and the sum of kernel time against the measured step.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
w=$1
O=gpurun_out/prof_r04_$w
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 bench.py --config $w --steps 4 --warmup 2 --no-cpu-baseline --no-launch-timing > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
tail -1 $O/run.log | cut -c1-200
f=$(find $O -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r04_cfg_${w}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
calls=sum(int(r['Calls']) for r in rows)
print("kernel time per step: %.2f ms, %d launches per step (6 steps profiled)" % (tot/1e6/6, calls/6))
for r in rows[:40]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']).split('(')[0][:70]
    print("%-72s %6s %8.3f ms/step %5.1f%% avg %8.1f us"%(n,r['Calls'],float(r['TotalDurationNs'])/1e6/6,100*float(r['TotalDurationNs'])/tot,float(r['AverageNs'])/1e3))
PY
