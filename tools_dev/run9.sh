cd /root/repo
timeout 900 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -s -k "three_processes or bench_starts or cli_driver_layer_split" 2>&1 | tail -25 > gpurun_out/exp9_tests.log
bash tools_dev/bench_sweep.sh cus -- --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedules -- base cus32p1:PIPEINFER_DRAFT_CUS=32 cus32p0:PIPEINFER_DRAFT_CUS=32,PIPEINFER_DRAFT_PRIORITY=0 cus64p0:PIPEINFER_DRAFT_CUS=64,PIPEINFER_DRAFT_PRIORITY=0 cus16p0:PIPEINFER_DRAFT_CUS=16,PIPEINFER_DRAFT_PRIORITY=0 cus48p0:PIPEINFER_DRAFT_CUS=48,PIPEINFER_DRAFT_PRIORITY=0 > gpurun_out/exp9_sweep.log 2>&1
tail -8 gpurun_out/exp9_sweep.log
