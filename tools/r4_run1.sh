set -u
O=gpurun_out
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['value_host_fed'], d['value_sustained'], d['roofline']['kernel'], d['roofline']['frac'])
"
