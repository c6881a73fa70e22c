set -u
O=gpurun_out
python bench.py > $O/bench_now.json 2> $O/bench_now.err; echo rc=$?
tail -3 $O/bench_now.err
