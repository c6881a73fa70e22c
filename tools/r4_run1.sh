set -u
O=gpurun_out
python -m pytest tests/test_gpu_tracker.py -x -q -m gpu -k "look_ahead" > $O/t_fast.log 2>&1; echo "tests rc=$?"; tail -12 $O/t_fast.log | cut -c1-200
