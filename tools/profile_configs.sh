#!/bin/bash
# Kernel-time summaries (rocprofv3 --kernel-trace --stats) of the tape networks' training steps at their configuration sizes.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for w in "$@"; do
  O=gpurun_out/prof_cfg_$w
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 tools/config_bench.py $w --steps 3 --warmup 1 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
  tail -1 $O/run.log
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per step: %.2f ms (4 steps profiled)" % (tot/1e6/4))
for r in rows[:28]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']).split('(')[0][:62]
    print("%-64s %6s %8.3f ms/step %5.1f%% avg %8.1f us"%(n,r['Calls'],float(r['TotalDurationNs'])/1e6/4,100*float(r['TotalDurationNs'])/tot,float(r['AverageNs'])/1e3))
PY
done
