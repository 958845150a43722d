#!/bin/bash
# repeat the concurrency-heavy tests (feeder + coalescing + cancels + transport) to catch rare stalls / races
mkdir -p gpurun_out/r3soak
fail=0
for i in $(seq 1 12); do
    timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -k "coalesc or cancel or segmented or priority_context or lossless_under_timing or three_processes or bench_starts" > gpurun_out/r3soak/run_$i.log 2>&1 || { fail=$((fail+1)); echo "iteration $i FAILED"; tail -20 gpurun_out/r3soak/run_$i.log; }
    tail -1 gpurun_out/r3soak/run_$i.log
done
echo "soak: $fail failing iteration(s) of 12"
