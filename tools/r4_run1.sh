set -u
O=gpurun_out
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
bash tools/profile_round_ab.sh
cat $O/ab_round4.txt
