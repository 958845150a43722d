#!/bin/bash
out=gpurun_out/r3tails; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "tails or attention_vs_oracle or abort_word" 2>&1 | tail -12 > $out/ops.log; cat $out/ops.log
timeout 1500 python -m pytest tests/test_gpu_engine.py -x -q -k "scenario_logits or greedy_continuation or every_run_of_a_speculative or (pipeinfer_driver_is_lossless and tiny_llama) or coalesc or segmented or cancel" 2>&1 | tail -12 > $out/engine.log; cat $out/engine.log
for t in 0 1; do
  if [ $t = 1 ]; then export PI_NO_TAILS=1; else unset PI_NO_TAILS; fi
  timeout 600 python bench.py --workload 1b --no-cpu-baseline --no-other-schedules --mode 2 --steps 4 --warmup 1 > $out/b1b_$t.json 2> $out/b1b_$t.err
  timeout 900 python bench.py --no-cpu-baseline --no-other-schedules --steps 5 --warmup 1 > $out/b70_$t.json 2> $out/b70_$t.err
  python - <<PY
import json
for f in ("$out/b1b_$t.json", "$out/b70_$t.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("PI_NO_TAILS=${PI_NO_TAILS:-unset}", d["config"]["workload"], "value", d["value"], "lossless", d.get("lossless_vs_target_only"), "target call", r.get("target_decode_call"), "draft call", r.get("draft_decode_call"))
    except Exception as e:
        print(f, "failed", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
done
