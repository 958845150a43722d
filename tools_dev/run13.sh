cd /root/repo
df -h /tmp | tail -1 > gpurun_out/exp13_misc.log; free -g | head -2 >> gpurun_out/exp13_misc.log
PIPEINFER_ALLOW_SHARED_GPU=1 timeout 1500 python3 bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_70b_n8_one_device.json 2> gpurun_out/bench_70b_n8_one_device.err; echo "gpus8 rc=$?" >> gpurun_out/exp13_misc.log; tail -c 1500 gpurun_out/bench_70b_n8_one_device.err >> gpurun_out/exp13_misc.log
df -h /tmp | tail -1 >> gpurun_out/exp13_misc.log
timeout 1200 python3 bench.py --workload falcon40b --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_falcon40b_n1.json 2> gpurun_out/bench_falcon40b_n1.err; echo "falcon rc=$?" >> gpurun_out/exp13_misc.log; tail -c 800 gpurun_out/bench_falcon40b_n1.err >> gpurun_out/exp13_misc.log
rm -f /tmp/pipeinfer_models/falcon-40b* /tmp/pipeinfer_models/llama-2-70b-q4_k_m-bigram-s24301-g1.0-n0.25.gguf
timeout 1500 python3 bench.py --workload goliath --n-draft 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_goliath_n1.json 2> gpurun_out/bench_goliath_n1.err; echo "goliath rc=$?" >> gpurun_out/exp13_misc.log; tail -c 800 gpurun_out/bench_goliath_n1.err >> gpurun_out/exp13_misc.log
df -h /tmp | tail -1 >> gpurun_out/exp13_misc.log
bash tools_dev/soak.sh 3 "three_processes or coalesc or cancel or bench_starts" > gpurun_out/exp13_soak.log 2>&1
tail -5 gpurun_out/exp13_soak.log
