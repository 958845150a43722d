cd /root/repo
timeout 1200 python3 bench.py --steps 10 --warmup 2 > gpurun_out/bench_r04_final.json 2> gpurun_out/bench_r04_final.err
tail -c 300 gpurun_out/bench_r04_final.err
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_r04_final.json') if l.startswith('{')][-1]); r=d['roofline']; print(d['value'], r['family'], r['frac'], r['traffic'], r['traffic_fetch_over_alg'], r['traffic_source'])"
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r04.log 2>&1; tail -2 gpurun_out/smoke_r04.log
bash tools_dev/soak.sh 4 > gpurun_out/soak_r04b.log 2>&1; tail -3 gpurun_out/soak_r04b.log
