#!/bin/bash
# A/B of one environment switch of the matrix-core mat-mat, interleaved processes: tools_dev/exp_mmq_ab.sh VAR A B [shapes] [nts]  -> gpurun_out/exp_mmq_ab_$VAR.log
cd "$(dirname "$0")/.."
VAR=$1; A=$2; B=$3
SH=${4:-"q4_K:8192:8192,q4_K:9216:8192,q4_K:8192:28672,q5_K:8192:8192"}
export NTS=${5:-"4,6,8,16"} NLAUNCH=${NLAUNCH:-1500}
OUT=gpurun_out/exp_mmq_ab_$VAR.log; : > $OUT
for rep in 1 2; do for v in $A $B; do
  echo "== $VAR=$v (rep $rep)" >> $OUT
  env $VAR=$v python3 tools_dev/perf_mmvq.py $SH >> $OUT 2>&1
done; done
python3 - "$OUT" <<'PY'
import sys,re,collections
cur=None; tab=collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1]):
    if l.startswith('=='): cur=l.split('(')[0].strip('= \n'); continue
    m=re.match(r'(\S+)\s+M=\s*(\d+) K=\s*(\d+) NT=(\d+):\s+([\d.]+) us',l)
    if m and cur: tab[(m.group(1),m.group(2),m.group(3),m.group(4))][cur].append(float(m.group(5)))
for k,v in tab.items(): print(k, {c: x for c,x in v.items()})
PY
