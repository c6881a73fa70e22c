C="--mapper --steps 20 --warmup 4 --cpu-frames 0 --host-steps 0 --latency-frames 0"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('%.0f fps  %.2f ms/step  %dx%d fibers %d  host cpus %.1f' % (d['value'], d['ms_per_step'], c['groups_per_gpu'], c['sequences_per_group'], c['group_steps_per_worker'], d['host_cpu']['cpus_busy']))"; }
echo "default"; for i in 1 2 3; do python3 bench.py $C 2>/dev/null | val; done
echo "16 groups x 256, 1 per worker"; python3 bench.py $C --groups 16 --fibers 1 2>/dev/null | val
echo "16 groups x 256, workers 8 fibers 2"; python3 bench.py $C --groups 16 --fibers 2 --workers 8 2>/dev/null | val
echo "48 groups, fibers 3"; python3 bench.py $C --groups 48 --fibers 3 --workers 16 --seqs 3072 2>/dev/null | val
echo "64 groups x 64, fibers 4"; python3 bench.py $C --groups 64 --fibers 4 --workers 16 2>/dev/null | val
echo "32 groups, workers 14"; python3 bench.py $C --groups 32 --fibers 2 --workers 14 2>/dev/null | val
echo "camera texture default"; python3 bench.py $C --texture camera 2>/dev/null | val
