T=./slam-sdvl_amd/host/track_sequence
A="--synthetic 300 --prerender --quiet --json"
for t in camera plane; do
echo "b1 fork $t"; $T $A --texture $t
echo "b1 nofork $t"; SDVL_DETECT_FORK=0 $T $A --texture $t
done
echo "b16 batch fork"; $T $A --texture camera --trackers 16 --batch
echo "b16 batch nofork"; SDVL_DETECT_FORK=0 $T $A --texture camera --trackers 16 --batch
$T --synthetic 300 --texture camera --prerender --quiet --profile 2>&1 >/dev/null | head -24
python -m pytest tests/test_gpu_camera_texture.py -x -q -m gpu -k "handleframe or chunks" 2>&1 | tail -3
