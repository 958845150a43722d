cd /root/repo
PI_MMQ_S1NC2=1 timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mul_mat_vs_oracle or qkv_rope" 2>&1 | tail -3 > gpurun_out/exp14_tests.log
NLAUNCH=600 bash tools_dev/exp_mmq_ab.sh PI_MMQ_S1NC2 0 1 "q4_K:57344:8192,q4_K:28672:8192,q6_K:32000:8192,q4_K:44032:4096" "24,32" > gpurun_out/exp14_ab.log 2>&1
tail -8 gpurun_out/exp14_ab.log
bash tools_dev/bench_sweep.sh nc2 -- --mode 2 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedules -- s1nc2off s1nc2on:PI_MMQ_S1NC2=1 > gpurun_out/exp14_sweep.log 2>&1
python3 - <<'PY' >> gpurun_out/exp14_sweep.log
import json
for n in ("s1nc2off","s1nc2on"):
    d=json.loads([l for l in open(f"gpurun_out/sweep_nc2/{n}.json") if l.startswith("{")][-1]); print(n, "encode_tps", d["encode_tps"], "ttft", d["ttft_s"])
PY
tail -3 gpurun_out/exp14_sweep.log
