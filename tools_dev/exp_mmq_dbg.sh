#!/bin/bash
# what bounds the matrix-core mat-mat: PI_MMQ_DBG toggles of a -DPI_MMQ_DBG_BUILD library (1 no compute, 2 no staging, 4 no barrier, 8 no chains; wrong results)
# usage (GPU box): tools_dev/exp_mmq_dbg.sh   -> gpurun_out/exp_mmq_dbg.log
cd "$(dirname "$0")/.."
OUT=gpurun_out/exp_mmq_dbg.log; : > $OUT
SH="q4_K:8192:8192,q4_K:28672:8192,q4_K:57344:8192,q4_K:8192:28672,q6_K:8192:28672"
echo "== product library" >> $OUT
NTS=1,4,6,8,12,16 python3 tools_dev/perf_mmvq.py $SH >> $OUT 2>&1
for d in 0 1 2 3 4 6 8 10; do
  echo "== dbg library PI_MMQ_DBG=$d" >> $OUT
  PIPEINFER_LIB=$PWD/pipeinfer_amd/libpipeinfer_dbg.so PI_MMQ_DBG=$d NTS=6,16 python3 tools_dev/perf_mmvq.py $SH >> $OUT 2>&1
done
tail -5 $OUT
