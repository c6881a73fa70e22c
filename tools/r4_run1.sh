set -u
O=gpurun_out
slam-sdvl_amd/host/frontend_link_check > $O/flc.txt 2>&1; echo "frontend_link_check rc=$?"; cat $O/flc.txt
python -m pytest tests/test_gpu_tracker.py -x -q -m gpu -k "cpp" > $O/t_cpp.log 2>&1; echo "cpp tests rc=$?"; tail -3 $O/t_cpp.log
