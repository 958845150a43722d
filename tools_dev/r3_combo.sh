#!/bin/bash
out=gpurun_out/r3combo; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_engine.py -x -q -k "(pipeinfer_driver_is_lossless or every_run_of_a_speculative) and (mpt or bloom or starcoder or stablelm or refact or baichuan)" 2>&1 | tail -5 > $out/engine_archs.log; cat $out/engine_archs.log
bash tools_dev/draft_sensitivity.sh 2>&1 | tee $out/sensitivity.log
bash tools_dev/soak_r3.sh 16 2>&1 | tail -30
