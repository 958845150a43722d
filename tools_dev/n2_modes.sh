#!/bin/bash
# On the GPU box: the 2-rank line on ONE device, several times, with and without a cap on the HIP hardware queues per process — which of the two states (about 58 / 105 tok/s)
# each start lands in.  Round 5's answer: the cap does not decide it; the state persists over consecutive starts on a box (profiles/r05_n2_one_device_states.log).  usage: bash tools_dev/n2_modes.sh <tag> [runs]
tag=${1:-n2modes}; runs=${2:-4}
out=$PWD/gpurun_out/$tag; mkdir -p $out
export PIPEINFER_ALLOW_SHARED_GPU=1
one() {   # label, env...
    local label=$1; shift
    for i in $(seq 1 $runs); do
        env "$@" python3 bench.py --gpus ${G:-2} --steps 2 --warmup 1 --no-cpu-baseline --detail $out/${label}_$i.detail.json > $out/${label}_$i.json 2> $out/${label}_$i.err
        python3 - $out/${label}_$i.detail.json $label $i <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; s = r["stages"][0]
    print(sys.argv[2], sys.argv[3], "value", d["value"], "stage_ms_per_call", r["slowest_stage_decode_call"]["gpu_ms_per_call"], "compute_ms/run", round(s["compute_ms"] / max(1, s["runs"]), 2),
          "skipped", s["runs_skipped"], "several", s.get("passes_of_several_runs"))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
P
    done
}
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -4
if [ -n "$N_SWEEP" ]; then      # 2, 4 and 8 ranks on the device
    one default PI_DUMMY=0
    G=4 one n4 PI_DUMMY=0
    G=8 one n8 PI_DUMMY=0
else
    one plain PI_DUMMY=0
    one hwq2 GPU_MAX_HW_QUEUES=2
    one hwq8 GPU_MAX_HW_QUEUES=8
fi
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -4
rm -f $out/*.err.tmp
