#!/bin/bash
# round 3, second GPU pass: dead-run elimination + abort word: tests, then the default bench at several segment sizes
mkdir -p gpurun_out/r3b
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_ops.py -x -q -k "dead_run or abort or cancel or topk or lossless" > gpurun_out/r3b/tests.log 2>&1
tail -5 gpurun_out/r3b/tests.log
show() {
python - "$1" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r3b/{name}.json"))
    r = d["roofline"]
    print(name, "VALUE", d["value"], "lossless", d["lossless_vs_target_only"], {k: (v["tokens_per_s"], v.get("lossless_vs_target_only")) for k, v in d["other_schedules"].items()}, d["acceptance"], r["target_cancelled_runs"], r["target_segments"], r["draft_decode_call"], r["target_decode_call"], "frac", r["frac"])
except Exception as e:
    print(name, "failed", e)
PY
}
run() { # name, env...
    local name=$1; shift
    env "$@" timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedules > gpurun_out/r3b/$name.json 2> gpurun_out/r3b/$name.err
    show $name
}
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r3b/seg40_full.json 2> gpurun_out/r3b/seg40_full.err
show seg40_full
run seg20 PIPEINFER_SEG_LAYERS=20
run seg10 PIPEINFER_SEG_LAYERS=10
run seg80 PIPEINFER_SEG_LAYERS=80
run seg40_prio0 PIPEINFER_DRAFT_PRIORITY=0
