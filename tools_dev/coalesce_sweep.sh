#!/bin/bash
# widest pass the target's feeder composes (PIPEINFER_COALESCE): 16 is the default
mkdir -p gpurun_out/r3co
for c in 16 8 12 24 16; do
  PIPEINFER_COALESCE=$c timeout 900 python bench.py --no-cpu-baseline --no-other-schedules --steps 6 --warmup 1 > gpurun_out/r3co/b_$c.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3co/b_$c.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("coalesce=$c value", d["value"], d.get("lossless_vs_target_only"), r.get("target_passes"), "all calls ms", r["target_all_calls"]["gpu_ms_per_call"])
PY
done
