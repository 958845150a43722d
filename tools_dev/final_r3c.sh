#!/bin/bash
# the 70B line again now that profiles/ holds the PMC pass of THIS kernel source (roofline.traffic), and the Goliath configuration
out=gpurun_out/r3final; mkdir -p $out
timeout 900 python bench.py --steps 10 --warmup 1 > $out/bench_70b.json 2> $out/bench_70b.err; tail -c 400 $out/bench_70b.json
rm -rf /tmp/pipeinfer_models/*
timeout 1500 python bench.py --workload goliath --n-draft 8 --steps 3 --warmup 1 > $out/bench_goliath.json 2> $out/bench_goliath.err; tail -c 300 $out/bench_goliath.json; tail -3 $out/bench_goliath.err
