#!/bin/bash
# Kernel timeline of ONE camera through SDVL::HandleFrame (host/track_sequence): rocprofv3 --kernel-trace of the B = 1 latency leg.
#   tools/lone_camera_trace.sh TAG [texture] [extra track_sequence args]   ->  gpurun_out/TAG/
set -e
TAG=${1:-lone}; TEX=${2:-camera}; shift; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
EXE=slam-sdvl_amd/host/track_sequence
ARGS="--synthetic 120 --texture $TEX --size 640 480 --cam 517.3 516.5 318.6 255.3 --seed 20260001 --prerender --quiet --json --trackers 1 $*"
$EXE $ARGS > $OUT/plain.json 2> $OUT/plain.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $EXE $ARGS > $OUT/traced.json 2> $OUT/traced.err
find $OUT/trace -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/trace
cat $OUT/plain.json
