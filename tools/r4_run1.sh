set -u
O=gpurun_out
python -m pytest tests/test_gpu_tracker.py tests/test_gpu_long.py -x -q -m gpu -k "bench_size or min_align_level_0 or 512_trackers or transient_ring" --durations=5 > $O/t_new.log 2>&1; echo "new tests rc=$?"; tail -15 $O/t_new.log
