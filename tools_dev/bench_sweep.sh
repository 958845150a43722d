#!/bin/bash
# bench.py under a list of environment settings, one summary line each (replaces the per-round sweep scripts: segment size, coalescing width, reaction window,
# draft latency sensitivity, abort word on / off, ...).
#   tools_dev/bench_sweep.sh <tag> [-- bench.py arguments] -- NAME[:VAR=V[,VAR=V...]] ...
# e.g. tools_dev/bench_sweep.sh seg -- --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedules -- seg20 seg10:PIPEINFER_SEG_LAYERS=10 seg5:PIPEINFER_SEG_LAYERS=5
#      tools_dev/bench_sweep.sh react -- --workload falcon40b --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedules -- r3ms:PIPEINFER_REACTION_US=3000 r10ms
#      tools_dev/bench_sweep.sh draft -- --steps 4 --no-cpu-baseline --no-other-schedules -- d0 d500:PIPEINFER_PRIORITY_DELAY_US=500 dm500:PIPEINFER_PRIORITY_DELAY_US=-500
# switches that exist: PIPEINFER_SEG_LAYERS, PIPEINFER_COALESCE, PIPEINFER_REACTION_US, PIPEINFER_PRIORITY_DELAY_US, PI_NO_ABORT_WORD, PI_NO_GRAPHS, PI_MMQ_G2, PI_MMQ_PIPE,
# PI_MMQ_MIN_NT, PI_MMQ_CFG, PIPEINFER_DRAFT_PRIORITY, PIPEINFER_LIB (a second build of the library)
cd "$(dirname "$0")/.."
tag=$1; shift
[ "$1" == "--" ] && shift
args=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
[ "$1" == "--" ] && shift
out=gpurun_out/sweep_$tag; mkdir -p $out
for spec in "$@"; do
    name=${spec%%:*}; envs=(); [ "$spec" != "$name" ] && IFS=, read -ra envs <<< "${spec#*:}"
    env "${envs[@]}" timeout 1200 python3 bench.py "${args[@]}" --detail $out/$name.detail.json > $out/$name.json 2> $out/$name.err || tail -3 $out/$name.err
    python3 - "$out/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1].replace(".json", ".detail.json"))); r = d["roofline"]
    print(sys.argv[2], "value", d["value"], "lossless", d.get("lossless_vs_target_only"), {k: v["tokens_per_s"] for k, v in d.get("other_schedules", {}).items()},
          "passes", r.get("target_passes"), "target call ms", r["target_decode_call"]["gpu_ms_per_call"], "draft call ms", r["draft_decode_call"]["gpu_ms_per_call"],
          "all calls ms", r["target_all_calls"]["gpu_ms_per_call"], "family", r.get("family"), r.get("frac"), "cancelled", r["target_cancelled_runs"]["cancelled_after_start"])
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done
