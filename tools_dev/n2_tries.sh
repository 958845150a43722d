#!/bin/bash
# On the GPU box: the two-rank line on ONE device, a few starts in a row (it has two states, see profiles/r05_n2_one_device_states.log).  usage: bash tools_dev/n2_tries.sh <out dir> [tries]
out=${1:-gpurun_out/n2tries}; tries=${2:-5}; mkdir -p $out
python3 -c 'import bench as B; B.ensure_model("/tmp/pipeinfer_models", "llama-2-70b", "q4_k_m", 0x5EED, 1.0, 0.25)' 2> $out/ensure.err
for i in $(seq 1 $tries); do
    PIPEINFER_ALLOW_SHARED_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --detail $out/n2_try${i}_detail.json > $out/n2_try$i.json 2> $out/n2_try$i.err
    echo "try $i: rc=$? $(cut -c80-130 $out/n2_try$i.json)"
done
