#!/bin/bash
# stage times of one group-step as the number of groups sharing the GPU grows (Bg = 64 sequences per group throughout)
python bench.py --steps 40 2>/dev/null | python tools/show_bench.py | head -1
for g in 1 2 4 8 16; do
  echo "== groups $g"
  python bench.py --steps 60 --cpu-frames 0 --groups $g --seqs $((g*64)) 2>/dev/null | python tools/show_bench.py | head -3
done
