#!/usr/bin/env python3
"""Host-to-device rate of this box's link, to put the host-fed bench number in proportion: one 1 GiB pinned buffer copied to
HBM with hipMemcpyAsync (DMA engine), best of 5.  Prints GB/s."""
import time
import torch

n = 1 << 30
src = torch.empty(n, dtype=torch.uint8, pin_memory=True)
src.fill_(7)
dst = torch.empty(n, dtype=torch.uint8, device="cuda:0")
best = 0.0
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    best = max(best, n / (time.perf_counter() - t0) / 1e9)
print("pinned host -> HBM, 1 GiB, DMA: %.1f GB/s" % best)
