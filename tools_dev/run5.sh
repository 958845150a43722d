cd /root/repo
NLAUNCH=1500 bash tools_dev/exp_mmq_ab.sh PI_MMQ_PIPE 0 1 "q4_K:8192:8192,q4_K:8192:28672" "6,16" > gpurun_out/exp5_ab.log 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "coalesced or headline" -s 2>&1 | tail -40 > gpurun_out/exp5_tests.log
timeout 900 python3 bench.py --steps 3 --warmup 1 > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err
tail -c 600 gpurun_out/bench_r04a.err
