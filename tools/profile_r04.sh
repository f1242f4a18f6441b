#!/bin/bash
# Round-4 evidence on one MI355X box: kernel-time summary of bench.py, HBM traffic (two PMC passes), SQ counters of the GEMM kernels.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-timing"
# (a) the default command: weight-gradient GEMMs on a second stream, per-kernel durations include the overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -o p -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-timing > $O/stats2.log 2>&1
# (b) everything on one stream: each duration is one kernel alone (what bench.py's instrumented step measures); the counter
#     passes below use the same setting
export MDE_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-timing > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 $B > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq1 -o p -- python3 $B > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- python3 $B > $O/sq2.log 2>&1
python tools/hbm_traffic.py $O/pmc_fetch $O/pmc_write 4 143 61 > $O/hbm_traffic.json
python tools/sq_summary.py $O > $O/pmc_gemm_in_network.txt; head -24 $O/pmc_gemm_in_network.txt
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; head -12 $O/kernel_stats.csv
f=$(find $O/stats2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_two_streams.csv
cat $O/hbm_traffic.json
