#!/bin/bash
# end of round 3: the whole GPU suite on the final tree, then the profiles and the bench lines that go under profiles/
out=gpurun_out/r3final; mkdir -p $out
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $out/gpu_tests.log; cat $out/gpu_tests.log
bash tools_dev/profile_round.sh r3prof 2>&1 | tail -12
timeout 900 python bench.py --steps 10 --warmup 1 > $out/bench_70b.json 2> $out/bench_70b.err; tail -c 600 $out/bench_70b.json
timeout 600 python bench.py --workload 7b --steps 10 --warmup 1 > $out/bench_7b.json 2> $out/bench_7b.err; tail -c 300 $out/bench_7b.json
rm -rf /tmp/pipeinfer_models/llama-2-7b* /tmp/pipeinfer_models/tinyllama*
df -h /tmp | tail -1
