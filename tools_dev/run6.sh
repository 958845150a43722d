cd /root/repo
PI_MMQ_G2_MIN=1 timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mul_mat" 2>&1 | tail -8 > gpurun_out/exp6_tests.log
NLAUNCH=1000 bash tools_dev/exp_mmq_ab.sh PI_MMQ_G2 0 1 "q4_K:57344:8192,q4_K:44032:4096,q4_K:28672:8192" "3,4,6,8" > gpurun_out/exp6_ab.log 2>&1
tail -5 gpurun_out/exp6_ab.log
