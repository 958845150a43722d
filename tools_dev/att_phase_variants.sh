#!/bin/bash
# builds tools_dev/lib_P1.so / lib_P2.so: the library with k_attention returning after phase 1 / phase 2 (phase timing by difference)
set -e
cd "$(dirname "$0")/../pipeinfer_amd/csrc"
for v in P1 P2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DATT_STOP_AFTER_$v -c kernels/attention.hip -o /tmp/k_att_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/k_runtime.o build/k_elementwise.o build/k_mmvq.o /tmp/k_att_$v.o build/e_*.o -o ../../tools_dev/lib_$v.so -lpthread
done
