#!/bin/bash
# layers per target segment (PIPEINFER_SEG_LAYERS; default a quarter of the model) with the 10 ms reaction window
mkdir -p gpurun_out/r3seg
for sl in 20 10 27 40 80 20; do
  PIPEINFER_SEG_LAYERS=$sl timeout 900 python bench.py --no-cpu-baseline --no-other-schedules --steps 5 --warmup 1 > gpurun_out/r3seg/b_$sl.json 2> gpurun_out/r3seg/b_$sl.err || tail -3 gpurun_out/r3seg/b_$sl.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3seg/b_$sl.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("seg_layers=$sl value", d["value"], d.get("lossless_vs_target_only"), r.get("target_passes"), "draft call", r["draft_decode_call"]["gpu_ms_per_call"])
PY
done
