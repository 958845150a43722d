#!/bin/bash
# usage: tools_dev/gpu.sh <timeout-seconds> '<command>'   — retries while no GPU slot / box is free (nothing is charged for those attempts)
T=$1; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@" > /tmp/gpurun_last.log 2>&1
    rc=$?
    if grep -q "status=transient" /tmp/gpurun_last.log || [ $rc -eq 3 ]; then sleep 45; continue; fi
    break
done
tail -60 /tmp/gpurun_last.log
exit $rc
