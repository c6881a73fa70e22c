set -u
O=gpurun_out
for cfg in "16 1" "16 2" "16 3" "12 2" "16 1" "16 2"; do
set -- $cfg
r=$(python bench.py --groups $1 --threads $2 --steps 40 --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 2>$O/ab_a.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); h=d['host_stage_ms_per_group_step']; print('%.1f k  %.2f ms/step  mapping=%.2f search=%.2f finish=%.2f total=%.2f' % (d['value']/1e3, d['ms_per_step'], h['mapping'], h['search'], h['finish'], h['total']))")
echo "groups $1 threads $2: $r $(grep -o '= [0-9.]* CPUs busy' $O/ab_a.err | tail -1) $(grep -o 'throttling in the timed region: [0-9]* periods' $O/ab_a.err | tail -1)"
done
