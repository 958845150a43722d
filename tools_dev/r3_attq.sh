#!/bin/bash
out=gpurun_out/r3attq; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -6 > $out/ops.log; cat $out/ops.log
timeout 1500 python -m pytest tests/test_gpu_engine.py -x -q -k "scenario_logits or greedy_continuation or every_run or (pipeinfer_driver_is_lossless and tiny_llama) or coalesc or segmented" 2>&1 | tail -6 > $out/engine.log; cat $out/engine.log
for t in 0 1; do
  if [ $t = 1 ]; then export PI_NO_ATT_QUANT=1; else unset PI_NO_ATT_QUANT; fi
  timeout 600 python bench.py --workload 1b --no-cpu-baseline --no-other-schedules --mode 2 --steps 4 --warmup 1 > $out/b1b_$t.json 2> $out/b1b_$t.err
  timeout 900 python bench.py --no-cpu-baseline --no-other-schedules --mode 2 --steps 4 --warmup 1 > $out/b70t_$t.json 2> $out/b70t_$t.err
  timeout 900 python bench.py --no-cpu-baseline --no-other-schedules --steps 6 --warmup 1 > $out/b70_$t.json 2> $out/b70_$t.err
  python - <<PY
import json
for f in ("$out/b1b_$t.json", "$out/b70t_$t.json", "$out/b70_$t.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("PI_NO_ATT_QUANT=${PI_NO_ATT_QUANT:-unset}", f.split("/")[-1], "value", d["value"], "lossless", d.get("lossless_vs_target_only"), "target call", r.get("target_decode_call", {}).get("gpu_ms_per_call"), "draft call", r.get("draft_decode_call", {}).get("gpu_ms_per_call"))
    except Exception as e:
        print(f, "failed", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
done
