python3 -m pytest tests/test_golden_round5.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('%.0f fps  %.3f ms/step  %dx%d  host cpus %.1f' % (d['value'], d['ms_per_step'], c['groups_per_gpu'], c['sequences_per_group'], d['host_cpu']['cpus_busy']))"; }
S="--cpu-frames 0 --host-steps 0 --sustained-frames 0 --latency-frames 0"
echo "S-C 64"; for i in 1 2; do python3 bench.py --workload S-C --steps 30 --warmup 4 $S 2>/dev/null | tee gpurun_out/bench_config_c_64seq.json | val; done
echo "S-C 512"; python3 bench.py --workload S-C --seqs 512 --steps 30 --warmup 4 --cpu-frames 60 --host-steps 0 --sustained-frames 0 --latency-frames 0 2>/dev/null | tee gpurun_out/bench_config_c.json | val
echo "mapper"; for i in 1 2 3; do python3 bench.py --mapper --steps 20 --warmup 4 $S 2>/dev/null | tee gpurun_out/bench_mapper.json | val; done
echo "mapper camera"; python3 bench.py --mapper --texture camera --steps 20 --warmup 4 $S 2>/dev/null | val
