set -u
O=gpurun_out
for cfg in "8192 32 2" "8192 32 1" "6144 24 2" "4096 16 1"; do
set -- $cfg
r=$(timeout -k 10 400 python bench.py --seqs $1 --groups $2 --fibers $3 --steps 40 --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 2>$O/ab_a.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k  %.2f ms/step workers=%s' % (d['value']/1e3, d['ms_per_step'], d['config'].get('host_worker_threads')))")
echo "seqs $1 groups $2 fibers $3: $r $(grep -o '= [0-9.]* CPUs busy' $O/ab_a.err | tail -1) $(grep -o 'throttling in the timed region: [0-9]* periods, [0-9.]* ms' $O/ab_a.err | tail -1)"
done
