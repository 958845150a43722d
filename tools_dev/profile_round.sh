#!/bin/bash
# On the GPU box: kernel-trace stats of the default bench command and a separate PMC FETCH_SIZE pass; results under gpurun_out/$1.
# usage: bash tools_dev/profile_round.sh <tag>         (then copy the summaries into profiles/, see the end of this file)
tag=${1:-prof}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
export PI_NO_GRAPHS=1      # rocprofv3 on this image segfaults inside hipGraphLaunch when it traces graph replay: the profiled runs launch kernel by kernel
B=$PWD/bench.py
cd /tmp
# 1. per-kernel time of the bench command (cpu_baseline leg off: it is CPU work; the other schedules on: they are part of the default run)
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $B --no-cpu-baseline --steps 2 --warmup 1 --detail $out/stats_bench_detail.json > $out/stats_bench.json 2> $out/stats_bench.err
echo "stats rc=$?"
# 1b. the target alone (mode 2), for the per-kernel breakdown of one decode call
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_t -o target -- python3 $B --no-cpu-baseline --no-other-schedules --mode 2 --steps 1 --warmup 0 --n-predict 64 --detail $out/stats_target_detail.json > $out/stats_target.json 2> $out/stats_target.err
echo "stats target-only rc=$?"
# 2. HBM bytes per launch (own pass, counters only); kernels are serialised under --pmc, so a short generation — of the DEFAULT schedule (mode 0, run coalescing on): the
#    matrix-core launches then have the pass widths `value` really runs on (3..8 tokens and the 32-token prompt chunks).  The profiled bench prints its own launch accounting
#    (roofline.by_family: algorithmic bytes per launch of every kernel family in THAT run); pmc_to_json.py divides the fetched bytes by it.
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc -o pmc -- python3 $B --no-cpu-baseline --no-other-schedules --steps 1 --warmup 0 --n-predict 48 --detail $out/pmc_bench_detail.json > $out/pmc_bench.json 2> $out/pmc_bench.err
echo "pmc rc=$?"
cd - > /dev/null
find $out -name "*kernel_stats.csv" | head; find $out -name "*counter_collection.csv" | head -2
python3 tools_dev/pmc_to_json.py $out/pmc "PI_NO_GRAPHS=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-other-schedules --steps 1 --warmup 0 --n-predict 48" $out/pmc_bench_detail.json > $out/pmc_fetch_size.json
# keep what travels back small: the raw counter csv can be large
find $out -name "*.db" -delete
find $out/pmc -name "*counter_collection.csv" -size +8M -delete
find $out -name "*kernel_trace.csv" -size +8M -delete
find $out -name "*agent_info.csv" -delete
du -sh $out
head -c 1500 $out/pmc_fetch_size.json
