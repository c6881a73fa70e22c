T=./slam-sdvl_amd/host/track_sequence
A="--synthetic 300 --texture camera --prerender --quiet --json"
echo "b1"; $T $A
echo "b1 graph"; SDVL_STEP_GRAPH=1 $T $A
echo "b16 threads"; $T $A --trackers 16
echo "b16 threads hwq16"; GPU_MAX_HW_QUEUES=16 $T $A --trackers 16
echo "b16 batch"; $T $A --trackers 16 --batch
echo "b64 batch"; $T $A --trackers 64 --batch
echo "b4 threads"; $T $A --trackers 4
echo "b4 threads hwq16"; GPU_MAX_HW_QUEUES=16 $T $A --trackers 4
