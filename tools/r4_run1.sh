set -u
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py -x -q -m gpu -k "not other_forms" > $O/t_fast.log 2>&1; echo "tests rc=$?"; tail -2 $O/t_fast.log
python tools/kernel_bench.py 256 10 > $O/kb_v.log 2>&1; echo "rc=$? $(grep -E '^  fast_cells|^  search_points|^  pyr_down|select_cells' $O/kb_v.log)"
bash tools/r4_ab.sh "SDVL_FAST_PAIRS=1" 3 40
