#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -k "prologue or scenario_logits or greedy_continuation or lossless or test_mul_mat_vs_oracle or bench_starts or segmented_context" > gpurun_out/r3f/tests.log 2>&1
tail -6 gpurun_out/r3f/tests.log
show() {
python - "$1" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r3f/{name}.json"))
    r = d["roofline"]
    print(name, "VALUE", d["value"], "lossless", d["lossless_vs_target_only"], {k: (v["tokens_per_s"], v.get("lossless_vs_target_only")) for k, v in d["other_schedules"].items()}, d["acceptance"], r["target_passes"], r["draft_decode_call"], r["target_decode_call"], "frac", r["frac"])
except Exception as e:
    print(name, "failed", e)
PY
}
run() { local name=$1; shift
    env "$@" timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedules > gpurun_out/r3f/$name.json 2> gpurun_out/r3f/$name.err
    show $name
}
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r3f/default_full.json 2> gpurun_out/r3f/default_full.err
show default_full
run nofuse PIPEINFER_FUSE_MAX_K=0
run seg10_c8 PIPEINFER_SEG_LAYERS=10 PIPEINFER_COALESCE=8
timeout 600 python bench.py --workload 1b --mode 2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedules > gpurun_out/r3f/draft_alone.json 2> gpurun_out/r3f/draft_alone.err
python -c "
import json; d=json.load(open('gpurun_out/r3f/draft_alone.json')); print('draft alone fused', d['value'], d['roofline']['target_decode_call'])"
PIPEINFER_FUSE_MAX_K=0 timeout 600 python bench.py --workload 1b --mode 2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedules > gpurun_out/r3f/draft_alone_nofuse.json 2> gpurun_out/r3f/draft_alone_nofuse.err
python -c "
import json; d=json.load(open('gpurun_out/r3f/draft_alone_nofuse.json')); print('draft alone unfused', d['value'], d['roofline']['target_decode_call'])"
