set -u
O=gpurun_out
bash tools/r4_ab.sh "SDVL_IA_WAVES=2" 2
bash tools/r4_ab.sh "SDVL_IA_WAVES=4" 2
python -m pytest tests -x -q -m gpu > $O/gpu_all.log 2>&1; echo "gpu suite rc=$?"; tail -3 $O/gpu_all.log
