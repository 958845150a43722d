#!/bin/bash
# matrix-core mat-mat: (waves per row group, row groups per workgroup) sweep per shape through PI_MMQ_CFG, and the cache-resident (NCOPY=1) compute-only bound
# usage (GPU box): tools_dev/exp_mmq_cfg.sh [lib.so]   -> gpurun_out/exp_mmq_cfg.log
cd "$(dirname "$0")/.."
OUT=gpurun_out/exp_mmq_cfg.log; : > $OUT
[ -n "$1" ] && export PIPEINFER_LIB=$PWD/$1
SH="q4_K:8192:8192,q4_K:9216:8192,q4_K:28672:8192,q4_K:57344:8192,q4_K:8192:28672,q6_K:8192:28672"
echo "== default dispatch" >> $OUT
NTS=1,4,6,8,12,16 python3 tools_dev/perf_mmvq.py $SH >> $OUT 2>&1
echo "== default dispatch, one copy (cache-resident where it fits)" >> $OUT
NCOPY=1 NTS=1,6,16 python3 tools_dev/perf_mmvq.py $SH >> $OUT 2>&1
for c in 17 18 14 24 22 43 42 41; do
  echo "== PI_MMQ_CFG=$c" >> $OUT
  PI_MMQ_CFG=$c NTS=6,16 python3 tools_dev/perf_mmvq.py $SH >> $OUT 2>&1
done
tail -3 $OUT
