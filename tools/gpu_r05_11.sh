C="--steps 10 --warmup 5 --cpu-frames 0 --host-steps 16 --sustained-frames 0 --latency-frames 0"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident %.0f  host-fed %.0f (%.1f GB/s)  queues %s' % (d['value'], d['value_host_fed'], d['roofline']['link']['achieved'], d['config']['gpu_max_hw_queues']))"; }
echo "== host-fed leg: 16 hardware queues (bench.py's default) vs the runtime's 4 (GPU_MAX_HW_QUEUES=4)"
for i in 1 2 3; do
  echo -n "A 16 queues: "; python3 bench.py $C 2>/dev/null | val
  echo -n "B  4 queues: "; GPU_MAX_HW_QUEUES=4 python3 bench.py $C 2>/dev/null | val
done
