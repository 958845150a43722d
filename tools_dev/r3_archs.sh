#!/bin/bash
out=gpurun_out/r3arch; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "alibi or partial_rotary or bias_and_clamp or rope_kv_store or attention_vs_oracle" 2>&1 | tail -8 > $out/ops.log; cat $out/ops.log
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "scenario_logits or greedy_continuation" 2>&1 | tail -12 > $out/engine1.log; cat $out/engine1.log
timeout 1500 python -m pytest tests/test_gpu_engine.py -x -q -k "(pipeinfer_driver_is_lossless or every_run_of_a_speculative) and (mpt or bloom or starcoder or stablelm or refact or baichuan)" 2>&1 | tail -12 > $out/engine2.log; cat $out/engine2.log
PIPEINFER_CRASH_TRACE=1 timeout 900 python -X faulthandler tools_dev/stress_segmented.py 120 > $out/stress.log 2> $out/stress.err; echo "stress rc=$?"; tail -3 $out/stress.log; tail -40 $out/stress.err
