#!/bin/bash
# end of a round on the GPU box: full GPU suite, the profile round (rocprofv3 kernel stats of the default and the target-only schedule + the PMC FETCH_SIZE pass), and the
# bench lines of BASELINE's configurations.   tools_dev/round_end.sh <round tag, e.g. r04> [skip-tests]
#   -> gpurun_out/<tag>z/*; then (here) copy what is to be judged into profiles/: tools_dev/round_end.sh prints the cp lines
cd "$(dirname "$0")/.."
tag=${1:-r05}; out=gpurun_out/${tag}z; mkdir -p $out
if [ "$2" != "skip-tests" ]; then
    PIPEINFER_SLOW_TESTS=1 timeout 3300 python3 -m pytest tests -m gpu -q --durations=5 > $out/gpu_tests.log 2>&1; tail -6 $out/gpu_tests.log      # (with the long oracle replays the driver's `-m gpu` run skips)
fi
bash tools_dev/profile_round.sh ${tag}prof > $out/profile_round.log 2>&1; tail -3 $out/profile_round.log | cut -c1-300
line() { # name, bench args...
    local name=$1; shift
    timeout 1500 python3 bench.py "$@" --detail $out/bench_${name}_detail.json > $out/bench_$name.json 2> $out/bench_$name.err || tail -3 $out/bench_$name.err
    python3 -c "
import json,sys; d=json.load(open('$out/bench_${name}_detail.json')); r=d['roofline']
print('$name', d['value'], d['lossless_vs_target_only'], {k:(v['tokens_per_s'], v['lossless_vs_target_only']) for k,v in d['other_schedules'].items()}, r['family'], r['frac'], r['traffic'], r['target_decode_call'], r['draft_decode_call'], d.get('cpu_baseline', {}).get('value'))"
}
# (the GPU box's /tmp holds 79 GB: the synthetic GGUFs of one configuration at a time)
cache=${PIPEINFER_CACHE:-/tmp/pipeinfer_models}
line 70b_n1 --steps 10 --warmup 2
line 7b_n1 --workload 7b --steps 5 --warmup 1
rm -f $cache/llama-2-70b-* $cache/llama-2-7b-*
line falcon40b_n1 --workload falcon40b --steps 3 --warmup 1 --no-cpu-baseline
rm -f $cache/falcon-*
line goliath_n1 --workload goliath --n-draft 8 --steps 3 --warmup 1 --no-cpu-baseline
rm -f $cache/goliath-*
# one-device runs of the N > 1 job (every rank on the one GPU): the 70B model's file is generated once and mapped by every rank
python3 -c "import bench as B; B.ensure_model('$cache', 'llama-2-70b', 'q4_k_m', 0x5EED, 1.0, 0.25)" 2>> $out/bench_70b_one_device.err
for n in 2 4 8; do
    PIPEINFER_ALLOW_SHARED_GPU=1 timeout 1500 python3 bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline --detail $out/bench_70b_n${n}_one_device_detail.json > $out/bench_70b_n${n}_one_device.json 2> $out/bench_70b_n${n}_one_device.err; echo "$n ranks on one device: rc=$? $(cut -c1-160 $out/bench_70b_n${n}_one_device.json)"
done
echo "copy (in the build container):"
echo "  cp gpurun_out/${tag}prof/pmc_fetch_size.json profiles/${tag}_pmc_fetch_size_70b.json"
echo "  for f in 70b_n1 7b_n1 falcon40b_n1 goliath_n1 70b_n2_one_device 70b_n4_one_device 70b_n8_one_device; do cp $out/bench_\${f}_detail.json profiles/${tag}_bench_\$f.json; done"
echo "  cp gpurun_out/${tag}prof/stats/*kernel_stats.csv profiles/${tag}_bench_70b_default_eager_kernel_stats.csv; cp gpurun_out/${tag}prof/stats_t/*kernel_stats.csv profiles/${tag}_bench_70b_target_only_eager_kernel_stats.csv"
