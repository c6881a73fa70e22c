set -u
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py -x -q -m gpu > $O/t_a.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_a.log
python tools/kernel_bench.py 256 6 > $O/kb_p.txt 2>&1; grep -E "pyr_down" $O/kb_p.txt
for i in 1 2 3; do python bench.py --steps 40 --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_step']; print('%.1f k' % (d['value']/1e3), 'search', k['search_points'], 'fast', k['fast_cells'], 'ia', k['image_align'], 'pyr', k['pyr_down'], 'shi', k['shi_tomasi'])"; done
