set -u
O=gpurun_out
run() { env "$@" python bench.py --steps 60 --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f' % (d['value']/1e3), end=' ')"; }
for i in 1 2 3; do
  echo -n "none: "; run SDVL_BENCH_NO_KERNEL_TIMING=1; echo -n " | dominant only: "; run SDVL_X=1; echo -n " | all: "; run SDVL_BENCH_TIME_ALL=1; echo
done
