set -x
timeout 900 python3 tools_dev/persist_beside.py 70b 1 > gpurun_out/persist_beside_70b_nt1.log 2>&1
tail -12 gpurun_out/persist_beside_70b_nt1.log
timeout 600 python3 tools_dev/persist_beside.py 70b 4 > gpurun_out/persist_beside_70b_nt4.log 2>&1
tail -10 gpurun_out/persist_beside_70b_nt4.log
