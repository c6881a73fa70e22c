set -u
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "search or depth_filter" > $O/t_s.log 2>&1; echo "tests rc=$?"; tail -2 $O/t_s.log
python tools/kernel_bench.py 256 6 > $O/kb_s.txt 2>&1; grep -E "search" $O/kb_s.txt
