#!/bin/bash
# Re-fit check of the conv tile cost model's relative rates (MDE_CONV_RATES=r256,r192) with whole-step timings.
for r in "1.15,1.10" "1.30,1.20" "1.00,1.00" "1.15,1.25" "1.30,1.10" "1.45,1.30"; do
    echo -n "rates $r  "
    MDE_CONV_RATES=$r python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-launch-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
