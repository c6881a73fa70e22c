set -u
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fast_cells or detect or corner" > $O/t_fast.log 2>&1; echo "tests rc=$?"; tail -2 $O/t_fast.log
python tools/kernel_bench.py 256 10 > $O/kb_v.log 2>&1; echo "rc=$? $(grep -E '^  fast_cells' $O/kb_v.log)"
for cfg in "512 8" "512 16" "1024 16" "1024 8"; do
set -- $cfg
r=$(timeout -k 10 300 python bench.py --workload S-C --seqs $1 --groups $2 --steps 30 --warmup 4 --cpu-frames 0 --host-steps 0 --sustained-frames 0 2>$O/ab_a.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k  %.2f ms/step' % (d['value']/1e3, d['ms_per_step']))")
echo "S-C seqs $1 groups $2: $r  $(grep -o '= [0-9.]* CPUs busy' $O/ab_a.err | tail -1)"
done
