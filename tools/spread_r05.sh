val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'], end=' ')"; }
S="--cpu-frames 0 --host-steps 0 --sustained-frames 0 --latency-frames 0"
echo "== run-to-run spread on ONE box: six consecutive runs of the resident leg (40 timed steps, warm-up 5), tracked frames/s"
echo -n "S-A plane : "; for i in 1 2 3 4 5 6; do python3 bench.py $S 2>/dev/null | val; done; echo
echo -n "S-A camera: "; for i in 1 2 3 4 5 6; do python3 bench.py --texture camera $S 2>/dev/null | val; done; echo
echo -n "S-B camera: "; for i in 1 2 3; do python3 bench.py --workload S-B --texture camera $S 2>/dev/null | val; done; echo
echo -n "driver-like (--steps 20 --warmup 5), plane: "; for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 $S 2>/dev/null | val; done; echo
