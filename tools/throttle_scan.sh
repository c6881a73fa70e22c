#!/bin/bash
# CFS throttling of the container during a bench run, for several group counts (Bg = 64)
cat /sys/fs/cgroup/cpu.max
python bench.py --steps 40 2>/dev/null | python tools/show_bench.py | head -1
for g in 10 12 14 16; do
  echo "== groups $g"
  grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
  python bench.py --steps 100 --cpu-frames 0 --groups $g --seqs $((g*64)) 2>/dev/null | python tools/show_bench.py | head -1
  grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
done
