cd /root/repo
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mul_mat" 2>&1 | tail -3 > gpurun_out/exp7_tests.log
NLAUNCH=1000 bash tools_dev/exp_mmq_ab.sh PI_MMQ_B2 0 1 "q4_K:8192:8192,q4_K:9216:8192,q4_K:8192:28672,q6_K:8192:28672,q5_K:8192:8192" "4,6,8" > gpurun_out/exp7_ab.log 2>&1
NLAUNCH=1000 bash tools_dev/exp_mmq_ab.sh PI_MMQ_G2_MIN 100000 1024 "q4_K:28672:8192,q4_K:22016:4096,q4_K:32768:8192" "4,6,8" > gpurun_out/exp7_g2min.log 2>&1
tail -3 gpurun_out/exp7_ab.log
