set -u
O=gpurun_out
python -m pytest tests/test_gpu_tracker.py -x -q -m gpu -k "look_ahead or transient" > $O/t_fast.log 2>&1; echo "tests rc=$?"; tail -5 $O/t_fast.log
