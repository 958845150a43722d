#!/bin/bash
# end of round 3: full GPU suite, profile round (rocprofv3 stats + PMC), bench lines of the BASELINE configurations
mkdir -p gpurun_out/r3z
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r3z/gpu_tests.log 2>&1
tail -9 gpurun_out/r3z/gpu_tests.log
bash tools_dev/profile_round.sh r3profz > gpurun_out/r3profz.log 2>&1
tail -3 gpurun_out/r3profz.log | cut -c1-200
timeout 1200 python bench.py --steps 10 --warmup 2 > gpurun_out/r3z/bench_70b.json 2> gpurun_out/r3z/bench_70b.err
python -c "
import json; d=json.load(open('gpurun_out/r3z/bench_70b.json')); r=d['roofline']; print('70b', d['value'], d['lossless_vs_target_only'], {k:(v['tokens_per_s'], v['lossless_vs_target_only']) for k,v in d['other_schedules'].items()}, r['frac'], r['traffic'], r['target_decode_call'], r['draft_decode_call'], d.get('cpu_baseline'))"
timeout 900 python bench.py --workload 7b --steps 5 --warmup 1 > gpurun_out/r3z/bench_7b.json 2> gpurun_out/r3z/bench_7b.err
python -c "
import json; d=json.load(open('gpurun_out/r3z/bench_7b.json')); r=d['roofline']; print('7b', d['value'], d['lossless_vs_target_only'], {k:(v['tokens_per_s'], v['lossless_vs_target_only']) for k,v in d['other_schedules'].items()}, r['frac'], r['target_decode_call'], r['draft_decode_call'])"
export TMPDIR=/tmp
( cd /tmp && PI_NO_GRAPHS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3z/stats7b -o t7b -- python3 $OLDPWD/bench.py --workload 7b --no-cpu-baseline --no-other-schedules --mode 2 --steps 1 --warmup 0 --n-predict 64 > $OLDPWD/gpurun_out/r3z/stats7b.json 2> $OLDPWD/gpurun_out/r3z/stats7b.err )
find gpurun_out/r3z/stats7b -name "*.db" -delete; find gpurun_out/r3z/stats7b -name "*kernel_trace.csv" -delete; find gpurun_out/r3z/stats7b -name "*agent_info.csv" -delete
timeout 900 python bench.py --workload falcon40b --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3z/bench_falcon40b.json 2> gpurun_out/r3z/bench_falcon40b.err
python -c "
import json; d=json.load(open('gpurun_out/r3z/bench_falcon40b.json')); print('falcon40b', d['value'], d['lossless_vs_target_only'], {k:(v['tokens_per_s'], v['lossless_vs_target_only']) for k,v in d['other_schedules'].items()})"
timeout 1200 python bench.py --workload goliath --n-draft 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3z/bench_goliath.json 2> gpurun_out/r3z/bench_goliath.err
python -c "
import json; d=json.load(open('gpurun_out/r3z/bench_goliath.json')); print('goliath', d['value'], d['lossless_vs_target_only'], {k:(v['tokens_per_s'], v['lossless_vs_target_only']) for k,v in d['other_schedules'].items()})"
