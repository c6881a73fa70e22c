#!/bin/bash
# A/B of two builds of libsdvl_hip.so (ab_tmp/libsdvl_hip_A.so, _B.so; not committed) on the resident leg, alternating:  tools/ab_lib.sh [reps] [bench args]
REPS=${1:-2}; shift
SHORT="--cpu-frames 0 --host-steps 0 --sustained-frames 0 --latency-frames 0 --lost-mix-steps 0"
cp slam-sdvl_amd/csrc/libsdvl_hip.so ab_tmp/libsdvl_hip_orig.so
for rep in $(seq $REPS); do
  for v in A B; do
    cp ab_tmp/libsdvl_hip_$v.so slam-sdvl_amd/csrc/libsdvl_hip.so
    python3 bench.py "$@" $SHORT 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$v', d['value'], d['ms_per_step'], {n: round(k[n],2) for n in ('filter_select','pose_refine','select_pack','track_project','image_align_pre') if n in k})"
  done
done
cp ab_tmp/libsdvl_hip_orig.so slam-sdvl_amd/csrc/libsdvl_hip.so
