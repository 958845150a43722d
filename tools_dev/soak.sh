#!/bin/bash
# repeat the concurrency-heavy GPU tests (feeder + coalescing + cancels + transport) to catch rare stalls / races; a fatal signal leaves its C stack in the log
#   tools_dev/soak.sh [iterations=12] [pytest -k expression]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/soak
n=${1:-12}; k=${2:-"coalesc or cancel or segmented or priority_context or lossless_under_timing or three_processes or bench_starts or headline"}
fail=0
export PIPEINFER_CRASH_TRACE=1
for i in $(seq 1 $n); do
    timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_engine.py -x -q -s -k "$k" > gpurun_out/soak/run_$i.log 2>&1 || { fail=$((fail+1)); echo "iteration $i FAILED"; tail -60 gpurun_out/soak/run_$i.log; }
    tail -1 gpurun_out/soak/run_$i.log
done
echo "soak: $fail failing iteration(s) of $n"
