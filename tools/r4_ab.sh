# A/B of one environment switch on the resident leg of bench.py, alternating runs on one box:  bash tools/r4_ab.sh "VAR=1" [runs] [steps]
set -u
O=gpurun_out
V="$1"; N=${2:-3}; S=${3:-40}
for i in $(seq 1 $N); do
  a=$(python bench.py --steps $S --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 2>$O/ab_a.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k  ia_ms_per_step=%s pyr=%s' % (d['value']/1e3, d.get('kernel_ms_per_step',{}).get('image_align'), d.get('kernel_ms_per_step',{}).get('pyr_down')))")
  b=$(env $V python bench.py --steps $S --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 2>$O/ab_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k  ia_ms_per_step=%s pyr=%s' % (d['value']/1e3, d.get('kernel_ms_per_step',{}).get('image_align'), d.get('kernel_ms_per_step',{}).get('pyr_down')))")
  echo "default: $a    |   $V: $b"
done
