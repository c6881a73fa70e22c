#!/usr/bin/env python3
"""Times TrackerFarm.run in chunks of a few steps: separates the fixed cost of a run() call (thread start, first touches)
from the per-step cost, with and without per-kernel event timing.  python tools/short_run_probe.py [chunk] [n_chunks]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
timing_from = int(sys.argv[3]) if len(sys.argv) > 3 else n_chunks // 2

torch.cuda.set_device(0)
pkg = importlib.import_module("slam-sdvl_amd")
trk = importlib.import_module("slam-sdvl_amd.tracker")
shard = importlib.import_module("slam-sdvl_amd.shard")
trk.bind_to_gpu_numa_node(0)
trk.configure()
B, G = 1024, 16
Bg = B // G
W, H = bench.W_IMG, bench.H_IMG
farm = trk.TrackerFarm(0, G, Bg, W, H, bench.TUM_CAM, host_threads_per_group=1)
ctxs = [bench.CtxView(pkg, farm.ctx_handle(g)) for g in range(G)]
ctx = ctxs[0]
n_frames = 1 + chunk * n_chunks
fb = W * H
buf = ctx.malloc(B * n_frames * fb)
seqs = shard.sequences_for_rank(0, 1, B)
for k in range(n_frames):
    views = [bench.make_view(pkg, bench.se3_exp(shard.sequence_twist(g) * k), shard.sequence_seed(g), k) for g in seqs]
    ctx.render(views, buf + k * B * fb)
ptrs = (buf + (np.arange(n_frames, dtype=np.uint64)[:, None] * B + np.arange(B, dtype=np.uint64)[None, :]) * fb).astype(np.uint64)
farm.reserve(Bg * (4 + n_frames // 3))
# busy the host for a while first (fresh-box start-up activity)
t_end = time.time() + 15
x = 0
while time.time() < t_end:
    x += 1
t0 = time.perf_counter()
farm.run(ptrs[:1], G)
print("bootstrap frame: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for c in range(n_chunks):
    if c == timing_from:
        for cv in ctxs:
            cv.timing(True)
        print("-- kernel event timing on")
    stats_buf = farm.alloc_stats(chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = farm.run(ptrs[1 + c * chunk:1 + (c + 1) * chunk], G, stats_buf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kf = sum(s.keyframe for s in st)
    print("chunk %d (%d steps): %.1f ms  = %.2f ms/step  %.0f fps  keyframes %d" % (c, chunk, dt * 1e3, dt * 1e3 / chunk, B * chunk / dt, kf))
