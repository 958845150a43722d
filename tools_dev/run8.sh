cd /root/repo
timeout 120 tools_dev/cumask_probe 32 > gpurun_out/cumask_probe.log 2>&1
timeout 120 tools_dev/cumask_probe 64 >> gpurun_out/cumask_probe.log 2>&1
timeout 900 python3 bench.py --steps 3 --warmup 1 > gpurun_out/bench_r04b.json 2> gpurun_out/bench_r04b.err
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/exp8_tests.log
tail -3 gpurun_out/exp8_tests.log
