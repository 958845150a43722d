#!/bin/bash
out=gpurun_out/r3arch; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "fused_norm or mul_mat_vs_oracle or alibi or partial_rotary or bias_and_clamp" 2>&1 | tail -8 > $out/ops2.log; cat $out/ops2.log
timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "scenario_logits or greedy_continuation" 2>&1 | tail -6 > $out/engine1b.log; cat $out/engine1b.log
timeout 2400 python -m pytest tests/test_gpu_engine.py -q -k "(pipeinfer_driver_is_lossless or every_run_of_a_speculative) and (mpt or bloom or starcoder or stablelm or refact or baichuan)" 2>&1 | tail -15 > $out/engine2b.log; cat $out/engine2b.log
