#!/bin/bash
out=gpurun_out/r3final; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "coalesc or cancel or segmented or lossless_under_timing or lossless_with_run" 2>&1 | tail -3
timeout 900 python bench.py --steps 10 --warmup 1 > $out/bench_70b.json 2> $out/bench_70b.err; tail -c 200 $out/bench_70b.json
timeout 600 python bench.py --workload 7b --steps 10 --warmup 1 > $out/bench_7b.json 2> $out/bench_7b.err
rm -rf /tmp/pipeinfer_models/*
timeout 1500 python bench.py --workload falcon40b --steps 4 --warmup 1 > $out/bench_falcon40b.json 2> $out/bench_falcon40b.err
rm -rf /tmp/pipeinfer_models/*
timeout 1500 python bench.py --workload goliath --n-draft 8 --steps 3 --warmup 1 > $out/bench_goliath.json 2> $out/bench_goliath.err
python - <<PY
import json
for w in ("70b","7b","falcon40b","goliath"):
    try:
        d=json.loads(open("$out/bench_%s.json" % w).read().strip().splitlines()[-1])
        print(w, d["value"], d.get("lossless_vs_target_only"), {k:(v["tokens_per_s"], v["lossless_vs_target_only"]) for k,v in d["other_schedules"].items()}, d["roofline"]["frac"], d["roofline"]["traffic"])
    except Exception as e: print(w, "failed", e)
PY
