#!/bin/bash
# The kernel sequence of one FCRN step with what precedes every __amd_rocclr_copyBuffer / fillBuffer launch (which host call is
# behind them?): one-stream run, kernel trace, the last step's launches in start order.  Output: gpurun_out/kseq/sequence.txt
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MDE_WGRAD_STREAM=0
O=gpurun_out/kseq
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-launch-timing > $O/run.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kseq/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last_adam = max(i for i, n in enumerate(names) if "adam_k" in n)
prev_adam = max(i for i, n in enumerate(names[:last_adam - 2]) if "adam_k" in n)
step = rows[prev_adam + 1:last_adam + 1]
with open("gpurun_out/kseq/sequence.txt", "w") as w:
    t0 = int(step[0]["Start_Timestamp"])
    prev_end = t0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        w.write("%9.1f us  gap %6.1f  dur %7.1f  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:110]))
        prev_end = max(prev_end, e)
    w.write("step span %.1f us, kernels %d, sum of gaps %.1f us\n" % ((prev_end - t0) / 1e3, len(step), 0))
print(open("gpurun_out/kseq/sequence.txt").read()[-600:])
PY
