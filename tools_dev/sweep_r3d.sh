#!/bin/bash
# round 3, fourth GPU pass: coalescing with cell hand-over detection: tests, then the default bench at several segment sizes
mkdir -p gpurun_out/r3d
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_ops.py -x -q -k "coalesc or two_weight_types or topk or attention_vs_oracle or three_processes" > gpurun_out/r3d/tests.log 2>&1
tail -6 gpurun_out/r3d/tests.log
show() {
python - "$1" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r3d/{name}.json"))
    r = d["roofline"]
    print(name, "VALUE", d["value"], "lossless", d["lossless_vs_target_only"], {k: (v["tokens_per_s"], v.get("lossless_vs_target_only")) for k, v in d["other_schedules"].items()}, d["acceptance"], r["target_cancelled_runs"], r["target_passes"], r["target_segments"], r["draft_decode_call"], r["target_decode_call"], "frac", r["frac"])
except Exception as e:
    print(name, "failed", e)
PY
}
run() { # name, env...
    local name=$1; shift
    env "$@" timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedules > gpurun_out/r3d/$name.json 2> gpurun_out/r3d/$name.err
    show $name
}
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r3d/default_full.json 2> gpurun_out/r3d/default_full.err
show default_full
run seg20 PIPEINFER_SEG_LAYERS=20
run seg10_c8 PIPEINFER_SEG_LAYERS=10 PIPEINFER_COALESCE=8
run seg16 PIPEINFER_SEG_LAYERS=16
run seg20_c8 PIPEINFER_SEG_LAYERS=20 PIPEINFER_COALESCE=8
