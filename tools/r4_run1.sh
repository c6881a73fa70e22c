set -u
O=gpurun_out
python -m pytest tests/test_gpu_tracker.py tests/test_gpu_long.py -x -q -m gpu > $O/t_flat.log 2>&1; echo "tracker+long tests rc=$?"; tail -4 $O/t_flat.log
bash tools/r4_ab.sh "SDVL_KEYFRAME_OBJECTS=1" 3 60
grep -h "host CPU\|host memory" $O/ab_a.err $O/ab_b.err
