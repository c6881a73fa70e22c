set -u
O=gpurun_out
python -m pytest tests/test_gpu_tracker.py tests/test_gpu_long.py -x -q -m gpu > $O/t_fast.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_fast.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['steps'], d['ms_per_step'], d['value_host_fed'], d['value_sustained'], d['roofline']['kernel'], d['roofline']['frac'])
print({k:v for k,v in d['sustained'].items() if k in ('host_rss_gb','keyframes_per_sequence','hbm_used_gb','value')})
"
grep -E "CPUs busy|page faults" $O/bench_default.err | tail -4
