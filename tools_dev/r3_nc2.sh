#!/bin/bash
# round 3: the multi-workgroup norm + quantise kernel and the 32-token passes of the matrix-core kernel — parity, then their time
out=gpurun_out/r3nc; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "mul_mat_vs_oracle or qkv_rope or norm or quant or integer_sums" 2>&1 | tail -5 > $out/ops.log; cat $out/ops.log
timeout 1200 python -m pytest tests/test_gpu_engine.py -x -q -k "scenario or greedy or every_run or quantiser_writes or full_size or lossless" 2>&1 | tail -5 > $out/engine.log; cat $out/engine.log
for nc in 1 2; do
  echo "== PI_MMQ_CHUNKS=$nc"; PI_MMQ_CHUNKS=$nc NTS=16,32,64 timeout 300 python tools_dev/perf_mmvq.py q4_K:8192:8192,q4_K:57344:8192,q4_K:28672:8192,q4_K:8192:28672,q6_K:8192:28672,q5_K:1024:8192,q4_K:10240:8192 2>&1 | tee $out/perf_nc$nc.log
done
for nc in 1 2; do
  PI_MMQ_CHUNKS=$nc timeout 600 python bench.py --no-cpu-baseline --no-other-schedules --mode 2 --steps 3 --warmup 1 > $out/bench_t_nc$nc.json 2> $out/bench_t_nc$nc.err
  python - <<PY
import json
d=json.loads(open("$out/bench_t_nc$nc.json").read().strip().splitlines()[-1])
print("chunks $nc target-only", d["value"], "encode_tps", d.get("encode_tps"), d["roofline"].get("target_decode_call"))
PY
done
for g in 1 0; do
  if [ $g = 1 ]; then export PI_NORM_WGS=1; else unset PI_NORM_WGS; fi
  timeout 600 python bench.py --no-cpu-baseline --no-other-schedules --mode 2 --steps 4 --warmup 1 > $out/bench_norm$g.json 2> $out/bench_norm$g.err
  python - <<PY
import json
d=json.loads(open("$out/bench_norm$g.json").read().strip().splitlines()[-1])
print("PI_NORM_WGS=${PI_NORM_WGS:-default} target-only", d["value"], d["roofline"].get("target_decode_call"), d["roofline"].get("frac"))
PY
done
