#!/bin/bash
# SQ / LDS counters of one cold-operand convolution shape, plain single-buffer tile against the halo-tiled form (tools/halo_probe.py)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/halo_pmc
rm -rf $O; mkdir -p $O
SHP="${SHP:-120 160 32 128 128 3}"
for pass in 1 2 3; do
  case $pass in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE";;
    2) C="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU";;
    3) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL";;
  esac
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$pass -o p -- python3 tools/halo_probe.py $SHP 4 > $O/p$pass.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
O="gpurun_out/halo_pmc"
tot=collections.defaultdict(collections.Counter); cnt=collections.Counter()
for d in ("p1","p2","p3"):
    for f in glob.glob(O+"/"+d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "conv_gemm_nt" not in k: continue
            key="halo" if k.rstrip(">) ").endswith("true") or ", true>" in k.split("conv_gemm_nt")[1][-12:] else "plain"
            key += " " + k.split("conv_gemm_nt")[1][:40]
            tot[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key,r["Counter_Name"])] += 1
with open(O+"/summary.txt","w") as w:
    for key in sorted(tot):
        w.write(key+"\n")
        for c,v in sorted(tot[key].items()):
            w.write("   %-28s %.4g per launch (%d launches)\n"%(c, v/cnt[(key,c)], cnt[(key,c)]))
print(open(O+"/summary.txt").read())
PY
