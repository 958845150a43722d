cd /root/repo
timeout 900 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -s -k "three_processes or bench_starts or cli_driver_layer_split" 2>&1 | tail -25 > gpurun_out/exp10_tests.log
bash tools_dev/bench_sweep.sh tcu -- --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedules -- tlo16:PIPEINFER_TARGET_CU_LO=16 tlo32:PIPEINFER_TARGET_CU_LO=32 tlo8:PIPEINFER_TARGET_CU_LO=8 nog2:PI_MMQ_G2=0 > gpurun_out/exp10_sweep.log 2>&1
tail -5 gpurun_out/exp10_sweep.log
