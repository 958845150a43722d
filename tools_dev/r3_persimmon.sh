#!/bin/bash
out=gpurun_out/r3pers; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "persimmon or fused_norm or partial_rotary" 2>&1 | tail -5 > $out/ops.log; cat $out/ops.log
timeout 1800 python -m pytest tests/test_gpu_engine.py -q -k "persimmon or pipeline_stages" 2>&1 | tail -12 > $out/engine.log; cat $out/engine.log
