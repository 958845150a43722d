cd /root/repo
timeout 1200 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "norm or quantis or two_row_groups or topk" 2>&1 | tail -5 > gpurun_out/exp11_tests.log
timeout 900 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "scenario_logits or greedy_continuation or every_run_of_a_speculative" 2>&1 | tail -5 >> gpurun_out/exp11_tests.log
bash tools_dev/bench_sweep.sh mw -- --mode 2 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedules -- mw1 mw0:PI_NORMQ_MW=0 mw1b mw0b:PI_NORMQ_MW=0 > gpurun_out/exp11_sweep.log 2>&1
tail -5 gpurun_out/exp11_sweep.log
