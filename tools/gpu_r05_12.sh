val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'], end=' ')"; }
echo "== resident leg, tracked frames/s by number of timed steps (warm-up 5), two runs each"
for K in 10 20 40 80; do
  for Q in 16 4; do
    echo -n "steps $K queues $Q: "
    for i in 1 2; do GPU_MAX_HW_QUEUES=$Q python3 bench.py --steps $K --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 --latency-frames 0 2>/dev/null | val; done
    echo
  done
done
echo "== camera texture"
for K in 20 40; do
  for Q in 16 4; do
    echo -n "steps $K queues $Q: "
    for i in 1 2; do GPU_MAX_HW_QUEUES=$Q python3 bench.py --texture camera --steps $K --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 --latency-frames 0 2>/dev/null | val; done
    echo
  done
done
