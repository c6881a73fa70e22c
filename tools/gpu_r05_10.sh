python3 -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; tail -3 gpurun_out/t_all.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --workload S-C --seqs 512 --steps 30 --warmup 4 --cpu-frames 60 --host-steps 0 --sustained-frames 0 --latency-frames 0 > gpurun_out/bench_config_c.json 2> gpurun_out/bench_config_c.err
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_like.json 2> gpurun_out/bench_driver_like.err
python3 -c "
import json
for f in ('bench_config_c','bench_driver_like'):
    d=json.loads(open('gpurun_out/'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('traffic'), r.get('traffic_stale'), (r.get('heaviest_by_instructions') or {}).get('kernel'), (r.get('heaviest_by_instructions') or {}).get('frac'))
"
