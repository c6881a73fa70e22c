set -u
O=gpurun_out
python -m pytest tests -x -q -m gpu > $O/gpu_all.log 2>&1; echo "gpu suite rc=$?"; tail -3 $O/gpu_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
