set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
run() { n=$1; shift
  python bench.py --steps 30 --warmup 5 --cpu-frames 0 --sustained-frames 0 --host-steps 0 "$@" > gpurun_out/exp/t_$n.json 2> gpurun_out/exp/t_$n.err
  python - $n <<'PY'
import json,sys
d=json.loads(open('gpurun_out/exp/t_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('%-24s resident %7d  cpus %.1f  stages %s'%(sys.argv[1], d['value'], d['host_cpu']['cpus_busy'], {k:v for k,v in d['host_stage_ms_per_group_step'].items() if v>0.3}), flush=True)
PY
}
run base
run g16_t2 --groups 16 --threads 2
run g16_t3 --groups 16 --threads 3
run base2
run g16_t2b --groups 16 --threads 2
