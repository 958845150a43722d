#!/bin/bash
# On the GPU box: the default line's value under different caps on the HIP hardware queues per process (GPU_MAX_HW_QUEUES; the runtime's default is 4).
# usage: bash tools_dev/n1_hwq.sh <tag> [runs]
tag=${1:-n1hwq}; runs=${2:-3}
out=$PWD/gpurun_out/$tag; mkdir -p $out
for q in 4 2 8 4 8; do
    for i in $(seq 1 $runs); do
        GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --no-other-schedules --steps 3 --warmup 1 --detail $out/q${q}_$i.detail.json > $out/q${q}_$i.json 2> $out/q${q}_$i.err
        python3 - $out/q${q}_$i.detail.json $q $i <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("hwq", sys.argv[2], "run", sys.argv[3], "value", d["value"], "frac", r["frac"], {k: d.get(k) for k in ("encode_tps", "draft_ms_per_token", "target_only_tps")})
except Exception as e:
    print("hwq", sys.argv[2], sys.argv[3], "failed", e)
P
    done
done
