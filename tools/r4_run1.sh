set -u
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_long.py -x -q -m gpu > $O/t_pre.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_pre.log
python tools/kernel_bench.py 256 5 > $O/kb_pre.txt 2>&1; grep -E "image_align" $O/kb_pre.txt
SDVL_IA_PRE=0 python tools/kernel_bench.py 256 5 > $O/kb_nopre.txt 2>&1; grep -E "image_align" $O/kb_nopre.txt
bash tools/r4_ab.sh "SDVL_IA_PRE=0" 3 60
