#!/bin/bash
# where a wave of the matrix-core mat-mat spends its cycles: SQ counters per kernel (perf_mmvq.py shapes, few launches)  -> gpurun_out/exp_mmq_pmc.log
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/exp_mmq_pmc; rm -rf $OUT; mkdir -p $OUT
SH=${1:-"q4_K:8192:8192,q4_K:57344:8192,q4_K:8192:28672,q6_K:8192:28672"}
P=$PWD/tools_dev/perf_mmvq.py
cd /tmp
NTS=${NTS:-1,6,16} NLAUNCH=12 timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -o a -- python3 $P $SH > $OUT/a.log 2>&1
NTS=${NTS:-1,6,16} NLAUNCH=12 timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- python3 $P $SH > $OUT/b.log 2>&1
cd - > /dev/null
python3 - $OUT <<'PY' > gpurun_out/exp_mmq_pmc.log
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_mm" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Grid_Size"] if "Grid_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +4M -delete
tail -30 gpurun_out/exp_mmq_pmc.log
