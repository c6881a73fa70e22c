#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/profile_round_ab.sh  -> gpurun_out/ab_round4.txt (copied to profiles/r04/)
set -u
O=gpurun_out
# round 4: the A/B lines DESIGN quotes, alternating runs on this box (tools/r4_ab.sh: resident leg only, 60 steps)
{
  echo "== kernel timing: default (every dispatch timed in the warm-up, the roofline's kernel only in the timed region) vs events on every dispatch"
  bash tools/r4_ab.sh "SDVL_BENCH_TIME_ALL=1" 3 60
  echo "== the search-pose-commit chain as a HIP graph (SDVL_STEP_GRAPH=1), host microseconds per group-step in the chain's submission"
  SDVL_STEP_GRAPH_STATS=1 bash tools/r4_ab.sh "SDVL_STEP_GRAPH=1" 3 60
  grep -h "chain submission" $O/ab_a.err | head -2
  grep -h "chain submission" $O/ab_b.err | head -2
  echo "== keyframes as objects (round 3) vs flat keyframes + appended table rows"
  bash tools/r4_ab.sh "SDVL_KEYFRAME_OBJECTS=1" 3 60
  grep -h "host CPU\|host memory" $O/ab_a.err | tail -2
  grep -h "host CPU\|host memory" $O/ab_b.err | tail -2
  echo "== image alignment: round-3 workgroup kernel (SDVL_IA_WAVE=0) vs one wave per job + precompute launch"
  bash tools/r4_ab.sh "SDVL_IA_WAVE=0" 3 60
  echo "== fused launches: track_align_prep + search_prepare as launches of their own (round 3)"
  bash tools/r4_ab.sh "SDVL_TRACK_ALIGN_RECORDS=1 SDVL_TRACK_SEPARATE_PREPARE=1" 2 60
  echo "== pyr_down: plain 3-D grid (round 3) vs all tiles of a frame on one XCD"
  bash tools/r4_ab.sh "SDVL_PYR_GRID3D=1" 2 60
} > $O/ab_round4.txt 2>&1
