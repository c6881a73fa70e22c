set -u
O=gpurun_out
for v in "X=1" "SDVL_EXPERIMENT_EXTRA_DETECT=1" "X=1" "SDVL_EXPERIMENT_EXTRA_DETECT=1"; do
r=$(env $v timeout -k 10 300 python bench.py --steps 40 --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 2>$O/ab_a.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('%.1f k  %.2f ms/step fast=%s pyr=%s search=%s' % (d['value']/1e3, d['ms_per_step'], k.get('fast_cells'), k.get('pyr_down'), k.get('search_points')))")
echo "$v: $r $(grep -o '= [0-9.]* CPUs busy' $O/ab_a.err | tail -1)"
tail -2 $O/ab_a.err | cut -c1-200
done
