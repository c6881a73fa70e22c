T=./slam-sdvl_amd/host/track_sequence
A="--synthetic 300 --prerender --quiet --json"
for t in camera plane; do
echo "b1 fork $t"; $T $A --texture $t
done
echo "b16 batch"; $T $A --texture camera --trackers 16 --batch
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
