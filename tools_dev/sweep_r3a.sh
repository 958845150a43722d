#!/bin/bash
# round 3, first GPU pass: the new tests, then the default bench at several segment sizes with and without the abort word
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_ops.py -x -q -k "abort or cancel or bench_starts" > gpurun_out/r3a/tests.log 2>&1
tail -5 gpurun_out/r3a/tests.log
run() { # name, env...
    local name=$1; shift
    env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedules > gpurun_out/r3a/$name.json 2> gpurun_out/r3a/$name.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3a/$name.json"))
    r = d["roofline"]
    print("$name", d["value"], d["acceptance"], r["target_cancelled_runs"], r["target_segments"], r["draft_decode_call"], r["target_decode_call"])
except Exception as e:
    print("$name failed", e)
PY
}
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/seg40_full.json 2> gpurun_out/r3a/seg40_full.err
python -c "
import json; d=json.load(open('gpurun_out/r3a/seg40_full.json')); r=d['roofline']; print('seg40_full', d['value'], d['other_schedules'], d['acceptance'], r['target_cancelled_runs'], r['target_segments'], r['draft_decode_call'], r['target_decode_call'], r['frac'])"
run seg40_noabort PI_NO_ABORT_WORD=1
run seg10 PIPEINFER_SEG_LAYERS=10
run seg5 PIPEINFER_SEG_LAYERS=5
run seg10_noabort PIPEINFER_SEG_LAYERS=10 PI_NO_ABORT_WORD=1
