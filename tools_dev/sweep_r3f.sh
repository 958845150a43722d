#!/bin/bash
mkdir -p gpurun_out/r3g
show() {
python - "$1" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r3g/{name}.json"))
    r = d["roofline"]
    print(name, "VALUE", d["value"], "lossless", d["lossless_vs_target_only"], d["acceptance"]["spec_runs"], d["acceptance"]["rate"], r["target_passes"], r["draft_decode_call"]["calls"], r["draft_decode_call"]["gpu_ms_per_call"], r["target_all_calls"]["gpu_ms_per_call"])
except Exception as e:
    print(name, "failed", e)
PY
}
run() { local name=$1; shift
    env "$@" timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedules > gpurun_out/r3g/$name.json 2> gpurun_out/r3g/$name.err
    show $name
}
run seg20_all
run seg20_y3 PIPEINFER_YIELDS_PER_PASS=3
run seg20_y2 PIPEINFER_YIELDS_PER_PASS=2
run seg10_y3 PIPEINFER_SEG_LAYERS=10 PIPEINFER_YIELDS_PER_PASS=3
run seg10_y4 PIPEINFER_SEG_LAYERS=10 PIPEINFER_YIELDS_PER_PASS=4
run seg8_y4 PIPEINFER_SEG_LAYERS=8 PIPEINFER_YIELDS_PER_PASS=4
run seg10_y5 PIPEINFER_SEG_LAYERS=10 PIPEINFER_YIELDS_PER_PASS=5
run seg5_y6 PIPEINFER_SEG_LAYERS=5 PIPEINFER_YIELDS_PER_PASS=6
