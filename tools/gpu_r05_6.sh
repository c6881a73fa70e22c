T=./slam-sdvl_amd/host/track_sequence
A="--synthetic 300 --prerender --quiet --json --texture camera"
echo "b1 default"; $T $A
echo "b1 ia1"; SDVL_IA_SMALL_WAVES=1 $T $A
echo "b1 nohelpers"; SDVL_POSE_SMALL_HELPERS=0 $T $A
echo "b1 neither"; SDVL_IA_SMALL_WAVES=1 SDVL_POSE_SMALL_HELPERS=0 $T $A
echo "b16 batch"; $T $A --trackers 16 --batch
$T --synthetic 300 --texture camera --prerender --quiet --profile 2>&1 >/dev/null | head -22
python -m pytest tests/test_gpu_camera_texture.py tests/test_gpu_tracker.py -x -q -m gpu 2>&1 | tail -3
