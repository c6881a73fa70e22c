set -u
O=gpurun_out
SDVL_PROFILE=1 python bench.py --steps 60 --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/prof.json 2> $O/prof.err; echo rc=$?
grep -n "^----" $O/prof.err | tail -3
awk '/^---- charged/{c++} c>=2' $O/prof.err | head -75
