#!/bin/bash
# (a) per-kernel cost of the abort word: target-only decode under rocprofv3 with and without it; (b) segments-ahead knob
mkdir -p gpurun_out/r3x
export TMPDIR=/tmp
R=$PWD
( cd /tmp && PI_NO_GRAPHS=1 PI_NO_ABORT_WORD=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3x/noabort -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-schedules --mode 2 --steps 1 --warmup 0 --n-predict 48 > $R/gpurun_out/r3x/noabort.json 2> $R/gpurun_out/r3x/noabort.err )
( cd /tmp && PI_NO_GRAPHS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3x/abort -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-schedules --mode 2 --steps 1 --warmup 0 --n-predict 48 > $R/gpurun_out/r3x/abort.json 2> $R/gpurun_out/r3x/abort.err )
find gpurun_out/r3x -name "*.db" -delete; find gpurun_out/r3x -name "*kernel_trace.csv" -delete; find gpurun_out/r3x -name "*agent_info.csv" -delete
for v in noabort abort; do echo "== $v"; python3 tools_dev/prof_summary.py gpurun_out/r3x/$v/t_kernel_stats.csv | head -12; done
show() {
python3 - "$1" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r3x/{name}.json")); r = d["roofline"]
    print(name, "VALUE", d["value"], d["lossless_vs_target_only"], r["target_passes"], r["draft_decode_call"], r["target_all_calls"]["gpu_ms_per_call"], r["target_decode_call"]["gpu_ms_per_call"])
except Exception as e:
    print(name, "failed", e)
PY
}
run() { local name=$1; shift
    env "$@" timeout 900 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedules > gpurun_out/r3x/$name.json 2> gpurun_out/r3x/$name.err
    show $name
}
run ahead1 PIPEINFER_SEGS_AHEAD=1
run ahead1_seg10 PIPEINFER_SEGS_AHEAD=1 PIPEINFER_SEG_LAYERS=10
run base
run mode2 
