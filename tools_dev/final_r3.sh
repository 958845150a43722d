#!/bin/bash
# full GPU suite (with -s output of the transport / coalescing tests kept), then the profile round
mkdir -p gpurun_out/r3e
timeout 2400 python -m pytest tests -m gpu -q --durations=8 -x > gpurun_out/r3e/gpu_tests.log 2>&1
tail -15 gpurun_out/r3e/gpu_tests.log
timeout 300 python -m pytest tests/test_gpu_engine.py -q -s -k "three_processes or bench_starts or coalesced_runs_equal or kernel_granularity" > gpurun_out/r3e/verbose_tests.log 2>&1
grep -E "stage [0-9]+:|bench --gpus|tiny:|70b |7b |segments=" gpurun_out/r3e/verbose_tests.log | head -30
bash tools_dev/profile_round.sh r3prof2 > gpurun_out/r3prof2.log 2>&1
tail -4 gpurun_out/r3prof2.log
