#!/bin/bash
# A/B of one environment switch on the resident leg and the lone camera:  tools/ab_env.sh VAR [bench args]   (runs: unset, VAR=1, unset, VAR=1)
VAR=$1; shift
SHORT="--cpu-frames 0 --host-steps 0 --sustained-frames 0 --latency-frames 0 --lost-mix-steps 0"
EXE=slam-sdvl_amd/host/track_sequence
ARGS="--synthetic 300 --texture camera --size 640 480 --cam 517.3 516.5 318.6 255.3 --seed 20260001 --prerender --quiet --json --trackers 1"
for rep in 1 2; do
  for v in "" 1; do
    if [ -z "$v" ]; then unset $VAR; else export $VAR=$v; fi
    python3 bench.py "$@" $SHORT 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v bench', d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items() if 'align' in k})"
    $EXE $ARGS | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v lone ', d['frames_per_s'], d['ms_per_frame_p50'], d['tracked'])"
  done
done
