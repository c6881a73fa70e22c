#!/usr/bin/env python3
"""Do the latency-bound kernels (pose_hypotheses / pose_refine: one workgroup or wave per frame) hide behind the GPU-filling
ones (fast_cells / select_corners) when they run on different streams?  Two contexts, two host threads: each loop alone,
then both together.  Perfect overlap: together ~ max(alone); none: together ~ sum."""
import ctypes as C
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

sdvl = importlib.import_module("slam-sdvl_amd")
shard = importlib.import_module("slam-sdvl_amd.shard")
import bench as B  # noqa: E402

n, reps = 256, 20
W, H = 640, 480
cA, cB = sdvl.Context(0), sdvl.Context(0)
buf = cA.device_malloc(n * W * H)
views = [B.make_view(sdvl, B.se3_exp(shard.sequence_twist(i) * 0), shard.sequence_seed(i), 0) for i in range(n)]
cA.synth_render(views, W, H, buf)
frames = [sdvl.Frame(cA, W, H) for _ in range(n)]
for i, f in enumerate(frames):
    f.set_image_device(buf + i * W * H)
cA.pyramid_build(frames)
arr = (C.c_void_p * n)(*[f.h for f in frames])
dp = sdvl.default_detect_params()
prng = np.random.default_rng(7)
nf = 190
pose_jobs = []
for j in range(n):
    P3 = np.stack([prng.uniform(-1.2, 1.2, nf), prng.uniform(-0.9, 0.9, nf), prng.uniform(1.5, 3.0, nf)], 1)
    a = P3[:, :2] / P3[:, 2:3] + prng.normal(0, 0.4 / 517.3, (nf, 2))
    obs = np.concatenate([a, P3, prng.integers(0, 3, nf)[:, None].astype(np.float64)], 1)
    pose_jobs.append((obs, [1, 0, 0, 0, 0, 0, 0], prng.integers(0, 2**31 - 1, 100)))


def detect():
    for _ in range(reps):
        cA._check(cA.lib.sdvl_detect_corners(cA.h, n, arr, C.byref(dp), 1000))
    cA.synchronize()


def pose():
    for _ in range(reps):
        cB.pose_from_matches(pose_jobs, fx=517.3)


def timed(fns):
    ts = [threading.Thread(target=f) for f in fns]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return (time.perf_counter() - t0) * 1e3


detect(); pose()
a, b, both = timed([detect]), timed([pose]), timed([detect, pose])
print("detect alone %.1f ms, pose alone %.1f ms, together %.1f ms (sum %.1f, max %.1f)" % (a, b, both, a + b, max(a, b)))
