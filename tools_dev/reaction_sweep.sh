#!/bin/bash
# PIPEINFER_REACTION_US (how long a coalescing feeder waits for the caller's reaction to a finished pass) on a configuration with a SLOW draft (Falcon-7B, 4.7 ms per step)
w=${1:-falcon40b}
mkdir -p gpurun_out/r3react
for us in ${SWEEP:-3000 0 1000 6000 12000}; do
  PIPEINFER_REACTION_US=$us timeout 900 python bench.py --workload $w --no-cpu-baseline --no-other-schedules --steps 4 --warmup 1 > gpurun_out/r3react/b_$us.json 2> gpurun_out/r3react/b_$us.err || tail -5 gpurun_out/r3react/b_$us.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3react/b_$us.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$w reaction_us=$us value", d["value"], d.get("lossless_vs_target_only"), r.get("target_passes"), "draft call", r["draft_decode_call"]["gpu_ms_per_call"])
PY
done
