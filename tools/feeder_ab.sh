#!/bin/bash
# A/B of the three changes to the host-fed path, all eight combinations on ONE box, 32 host-fed steps each:
#   own queue     = the feeder's copy stream in its own priority class (off: SDVL_FEED_NORMAL_PRIORITY=1)
#   host wait     = a group submits its step once its images have arrived (off: SDVL_RING_DEVICE_WAIT=1, hipStreamWaitEvent only)
#   two in flight = at most two image transfers queued on the device (off: SDVL_FEED_IN_FLIGHT=64, i.e. a ring's worth)
# Run on the GPU box from the repo root:  bash tools/feeder_ab.sh > gpurun_out/feeder_ab.txt
set -u
O=gpurun_out/feeder_ab
mkdir -p $O
run() {
  n=$1; shift
  env "$@" python3 bench.py --steps 6 --warmup 3 --cpu-frames 0 --sustained-frames 0 --host-steps 32 > $O/$n.json 2> $O/$n.err
  python3 - $O/$n.json "$n" <<'PY'
import json, sys
h = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["host_fed"]
f = h["feeder"]
print("%-28s %7d tracked frames/s  %5.1f GB/s H2D  group-steps that began before their images had arrived: %3d of %d" %
      (sys.argv[2], h["value"], h["pcie_h2d_gb_per_s"], f["group_steps_begun_before_their_images_arrived"], f["group_steps"]), flush=True)
PY
}
P=SDVL_FEED_NORMAL_PRIORITY=1; T=SDVL_FEED_IN_FLIGHT=64; D=SDVL_RING_DEVICE_WAIT=1
run none_of_the_three $P $T $D
run own_queue_only $T $D
run host_wait_only $P $T
run two_in_flight_only $P $D
run own_queue+host_wait $T
run own_queue+two_in_flight $D
run host_wait+two_in_flight $P
run all_three SDVL_NOTHING=1
run none_of_the_three_again $P $T $D
