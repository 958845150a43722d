#!/bin/bash
# how much of the one-GPU PipeInfer schedule hangs on the draft's latency?  the default 70B bench with the priority (draft) context slowed by a fixed delay per call
out=gpurun_out/r3ds; mkdir -p $out
for d in 0 500 1000 -500 -1000; do
  PIPEINFER_PRIORITY_DELAY_US=$d timeout 900 python bench.py --no-cpu-baseline --no-other-schedules --steps 4 --warmup 1 > $out/b_$d.json 2> $out/b_$d.err
  python - <<PY
import json
d=json.loads(open("$out/b_$d.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("delay_us=$d value", d["value"], "ms/step", d["ms_per_step"], "draft call ms", r["draft_decode_call"]["gpu_ms_per_call"], "draft calls", r["draft_decode_call"]["calls"], "target passes", r.get("target_passes"), "lossless", d.get("lossless"))
PY
done
