set -u
O=gpurun_out
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-style bench rc=$?"
python -c "
import json
for f in ('$O/bench_default.json','$O/bench_driver.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    v=d['roofline']['valu']
    print(d['value'], d['steps'], d['ms_per_step'], d['value_host_fed'], d['value_sustained'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us'], v['path_insts_per_frame'], v['path_frac'], v['kernel_frac'])
    print({k: v['insts_per_frame'][k] for k in list(v['insts_per_frame'])[:8]})
"
grep -E "CPUs busy|page faults" $O/bench_default.err | tail -2
