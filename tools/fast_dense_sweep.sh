#!/bin/bash
# fast_cells per 256 frames as the share of probed pixels that sends a cell down the dense path varies (SDVL_FAST_DENSE_NUM of 64):
# 0 = every cell dense, 64 = every cell through the candidate list; the default is where the sum over the camera-like inputs is least
for n in 0 8 12 16 20 24 32 48 64; do
  echo "== SDVL_FAST_DENSE_NUM=$n"
  SDVL_FAST_DENSE_NUM=$n python3 tools/fast_density_probe.py 256 2>&1 | grep "fast_cells"
done
