#!/bin/bash
# tools/ab_r05.sh "<ENV=VAL ... | ->" "<extra bench args for B | ->" PAIRS STEPS : alternating A (default) / B runs of the resident leg only
ENVB="$1"; ARGB="$2"; PAIRS=${3:-3}; STEPS=${4:-60}
[ "$ENVB" = "-" ] && ENVB=""
[ "$ARGB" = "-" ] && ARGB=""
COMMON="--steps $STEPS --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 --latency-frames 0"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f fps  %.3f ms/step  host cpus %.1f' % (d['value'], d['ms_per_step'], d['host_cpu']['cpus_busy']))"; }
for i in $(seq $PAIRS); do
  echo -n "A default            : "; python3 bench.py $COMMON $AB_COMMON_ARGS 2> gpurun_out/ab_a.err | val
  echo -n "B $ENVB $ARGB : "; env $ENVB python3 bench.py $COMMON $AB_COMMON_ARGS $ARGB 2> gpurun_out/ab_b.err | val
done
