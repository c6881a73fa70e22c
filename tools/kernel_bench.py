#!/usr/bin/env python3
"""Per-kernel micro-benchmark through the C-ABI (one stream, no tracker): dispatch time of every kernel on a batch of
n synthetic 640x480 frames, with the algorithmic bytes of SURVEY §8(d) -> GB/s.  Used to tune kernels in isolation.
    python tools/kernel_bench.py [n_frames] [reps]          (SDVL_KB_TEXTURE=camera: the camera-like texture)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

sdvl = importlib.import_module("slam-sdvl_amd")
shard = importlib.import_module("slam-sdvl_amd.shard")
import bench as B  # noqa: E402  (se3_exp, make_view)

B.TEXTURE = B.TEXTURES[os.environ.get("SDVL_KB_TEXTURE", "plane")]   # SDVL_KB_TEXTURE=camera: the camera-like texture
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H = 640, 480
ctx = sdvl.Context(0)
buf = ctx.device_malloc(2 * n * W * H)
views = [B.make_view(sdvl, B.se3_exp(shard.sequence_twist(i) * k), shard.sequence_seed(i), k) for k in (0, 3) for i in range(n)]
ctx.synth_render(views, W, H, buf)
fr0 = [sdvl.Frame(ctx, W, H) for _ in range(n)]
fr3 = [sdvl.Frame(ctx, W, H) for _ in range(n)]
for i in range(n):
    fr0[i].set_image_device(buf + i * W * H)
    fr3[i].set_image_device(buf + (n + i) * W * H)
dp = sdvl.default_detect_params()
cam = sdvl.Camera(W, H, *B.TUM_CAM)
import ctypes as C
arr0 = (C.c_void_p * n)(*[f.h for f in fr0])
arr3 = (C.c_void_p * n)(*[f.h for f in fr3])
lib = ctx.lib
ctx.pyramid_build(fr0); ctx.pyramid_build(fr3)
ctx._check(lib.sdvl_detect_corners(ctx.h, n, arr3, C.byref(dp), 1000))
ctx._check(lib.sdvl_orb_describe(ctx.h, n, arr3, 4096, None))
# alignment features on the plane z=2 seen from frame 0
nf = 190
rng = np.random.default_rng(1)
feats = (sdvl.AlignFeature * (nf * n))()
for j in range(n):
    px = np.stack([rng.uniform(48, W - 48, nf), rng.uniform(48, H - 48, nf)], 1)
    ray = np.stack([(px[:, 0] - B.TUM_CAM[2]) / B.TUM_CAM[0], (px[:, 1] - B.TUM_CAM[3]) / B.TUM_CAM[1], np.ones(nf)], 1)
    b = ray / np.linalg.norm(ray, axis=1, keepdims=True)
    for i in range(nf):
        f = feats[j * nf + i]
        f.px, f.py = px[i]; f.fx, f.fy, f.fz = b[i]; f.depth = 2.0 / b[i, 2]; f.valid = 1
jobs = [(fr0[j], fr3[j], j * nf, (j + 1) * nf, [1, 0, 0, 0, 0, 0, 0]) for j in range(n)]
ap = sdvl.default_align_params()
if os.environ.get('SDVL_KB_MAX_ITS'):
    ap.max_its = int(os.environ['SDVL_KB_MAX_ITS'])
ctx.timing_enable(True)
ctx.timing_reset()
its = 0
for _ in range(reps):
    ctx.pyramid_build(fr0)
    ctx._check(lib.sdvl_detect_corners(ctx.h, n, arr0, C.byref(dp), 1000))
    ctx._check(lib.sdvl_orb_describe(ctx.h, n, arr0, 4096, None))
    res = ctx.image_align(jobs, feats, cam, ap)
    its += sum(r.iters_run for r in res)
if hasattr(lib, 'sdvl_debug_ia_stamps'):
    st = (C.c_ulonglong * 8)()
    lib.sdvl_debug_ia_stamps(st)
    print('ia stamps job0 (ticks): prologue %d precompute %d pass1 %d reduce %d rebuildH %d solve %d evals %d' % tuple(st[:7]), 'slowest job: %d ticks, %d evals' % (st[7] >> 8, st[7] & 255))
# input stage: cv::undistort of n raw frames already in HBM (TUM fr1 coefficients) into the frames' level 0
import ctypes as _C
dist = sdvl.Distortion((_C.c_double * 5)(0.2624, -0.9531, -0.0054, 0.0026, 1.1633))
raw = (C.c_void_p * n)(*[buf + i * W * H for i in range(n)])
und = [sdvl.Frame(ctx, W, H) for _ in range(n)]
arru = (C.c_void_p * n)(*[f.h for f in und])
for _ in range(reps):
    ctx._check(lib.sdvl_frames_upload_undistorted(ctx.h, n, arru, raw, W, 1, C.byref(cam), C.byref(dist)))
# SearchPoint: per pair (frame 0 -> frame 3) nf fixed points on corners of frame 0, depth from the plane z = 2
c0s = ctx.detect_corners(fr0, dp, 1000)
d0s = ctx.orb_describe(fr0)
sreqs = (sdvl.SearchReq * (nf * n))()
nreq = 0
for j in range(n):
    T0 = B.se3_exp(shard.sequence_twist(j) * 0)
    T3 = B.se3_exp(shard.sequence_twist(j) * 3)
    R0, t0 = B.quat_to_R(T0[:4]), T0[4:]
    R3, t3 = B.quat_to_R(T3[:4]), T3[4:]
    Rw, tw = R0.T, -R0.T @ t0
    cs = c0s[j]
    sel = rng.choice(len(cs), size=min(nf, len(cs)), replace=False)
    for ci in sel:
        x, y, l = [int(q) for q in cs[ci]]
        px = np.array([x * (1 << l), y * (1 << l)], np.float64)
        ray = np.array([(px[0] - B.TUM_CAM[2]) / B.TUM_CAM[0], (px[1] - B.TUM_CAM[3]) / B.TUM_CAM[1], 1.0])
        bearing = ray / np.linalg.norm(ray)
        rw = Rw @ bearing
        sdepth = (2.0 - tw[2]) / rw[2]
        pc = R3 @ (Rw @ (bearing * sdepth) + tw) + t3
        r = sreqs[nreq]; nreq += 1
        r.cur, r.ref = fr3[j].h.value, fr0[j].h.value
        for k in range(7):
            r.cur_pose[k], r.ref_pose[k] = float(T3[k]), float(T0[k])
        r.px[0], r.px[1] = px
        r.bearing[0], r.bearing[1], r.bearing[2] = bearing
        r.idepth, r.idepth_std = 1.0 / sdepth, 0.05 / sdepth
        r.px0[0] = B.TUM_CAM[2] + B.TUM_CAM[0] * pc[0] / pc[2] + rng.normal() * 0.7
        r.px0[1] = B.TUM_CAM[3] + B.TUM_CAM[1] * pc[1] / pc[2] + rng.normal() * 0.7
        r.level, r.fixed = l, 1
        for k in range(32):
            r.desc[k] = int(d0s[j][ci, k])
sreqs = (sdvl.SearchReq * nreq).from_buffer(sreqs)
sp = sdvl.default_search_params()
for _ in range(max(1, reps // 3)):
    sres = ctx.search_points(sreqs, cam, sp)
n_found = sum(r.found for r in sres)
lk = sum(r.lk_its for r in sres) / max(1, n_found)
print("search: requests=%d found=%d lk_its/found=%.2f" % (nreq, n_found, lk))
# the same search on frames WITHOUT descriptors: every wave computes the descriptors of the corners it compares
ctx.synchronize()
t_a = ctx.timing_get().get("search_points", (0.0, 0))
ctx._check(lib.sdvl_detect_corners(ctx.h, n, arr3, C.byref(dp), 1000))   # new corner lists: descriptors invalid
if hasattr(lib, 'sdvl_debug_search_stamps'):
    lib.sdvl_debug_search_stamps(None, 1)
for _ in range(max(1, reps // 3)):
    sres2 = ctx.search_points(sreqs, cam, sp)
ctx.synchronize()
if hasattr(lib, 'sdvl_debug_search_stamps'):   # -DSDVL_SEARCH_STAMPS build
    st = (C.c_ulonglong * 8)()
    lib.sdvl_debug_search_stamps(st, 0)
    nw = max(1, st[7])
    print('search stamps per request (ticks): head %.1f patch %.1f scan %.1f descriptors %.1f lk %.1f | lk its %.2f descriptors %.2f | requests %d' %
          (st[0] / nw, st[1] / nw, st[2] / nw, st[3] / nw, st[4] / nw, st[5] / nw, st[6] / nw, nw))
t_b = ctx.timing_get().get("search_points", (0.0, 0))
assert os.environ.get('SDVL_KB_NOASSERT') or ([r.found for r in sres2] == [r.found for r in sres] and [tuple(r.px) for r in sres2] == [tuple(r.px) for r in sres])
print("search_points, descriptors on demand: %.1f us/launch (descriptors in HBM: %.1f us/launch)" %
      ((t_b[0] - t_a[0]) / max(1, t_b[1] - t_a[1]) * 1e3, t_a[0] / max(1, t_a[1]) * 1e3))
# tolerance-class LK sums (sdvl_search_params.lk_tree_sums)
sp_tree = sdvl.default_search_params()
sp_tree.lk_tree_sums = 1
ctx.synchronize()
t_c = ctx.timing_get().get("search_points", (0.0, 0))
for _ in range(max(1, reps // 3)):
    sres3 = ctx.search_points(sreqs, cam, sp_tree)
ctx.synchronize()
t_d = ctx.timing_get().get("search_points", (0.0, 0))
print("search_points, LK sums as a wave butterfly (tolerance class): %.1f us/launch; found flags that differ from the sequential sums: %d of %d" %
      ((t_d[0] - t_c[0]) / max(1, t_d[1] - t_c[1]) * 1e3, sum(a.found != b.found for a, b in zip(sres2, sres3)), nreq))
# pose stage: n jobs of 190 matches (20 % gross outliers) of a small camera motion; rand() draws from numpy (timing only)
prng = np.random.default_rng(7)
pose_jobs = []
true_pose = B.se3_exp(np.array([0.03, -0.02, 0.01, 0.004, -0.006, 0.003]))
qw, qx, qy, qz = true_pose[:4]
Rm = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
               [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
               [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
for j in range(n):
    P3 = np.stack([prng.uniform(-1.2, 1.2, nf), prng.uniform(-0.9, 0.9, nf), prng.uniform(1.5, 3.0, nf)], 1)
    pc = P3 @ Rm.T + true_pose[4:]
    a = pc[:, :2] / pc[:, 2:3] + prng.normal(0, 0.4 / 517.3, (nf, 2))
    bad = prng.random(nf) < 0.2
    a[bad] += prng.uniform(-40, 40, (int(bad.sum()), 2)) / 517.3
    obs = np.concatenate([a, P3, prng.integers(0, 3, nf)[:, None].astype(np.float64)], 1)
    pose_jobs.append((obs, [1, 0, 0, 0, 0, 0, 0], prng.integers(0, 2**31 - 1, 100)))
for _ in range(max(1, reps // 3)):
    pres = ctx.pose_from_matches(pose_jobs, fx=B.TUM_CAM[0])
ctx.synchronize()
t = ctx.timing_get()
print("pose: draws/job=%.1f inliers/job=%.1f" % (np.mean([r["n_draws"] for r in pres]), np.mean([len(r["inliers"]) for r in pres])))
counts = np.zeros(n, np.int32)
ctx._check(lib.sdvl_frames_corner_counts(ctx.h, n, arr0, counts.ctypes.data_as(C.POINTER(C.c_int32))))
nc = float(counts.mean())
i_ia = its / (reps * n)
P = [(W >> l) * (H >> l) for l in range(5)]
alg = {"pyr_down": (sum(P[:4]) + sum(P[1:])) / 4.0, "fast_cells": sum(P[:3]) + 16 * nc, "select_cells": 4 * 10000 + 6 * nc + 1600, "select_pack": 1600 + 30 * nc,
       "orb_describe": 993 * nc, "image_align": 147 * nf + 25 * nf * i_ia, "pose_hypotheses": 48 * nf + 6400, "undistort": 2 * W * H,
       "search_points": 12 * nc + nf * (121 + 164 + lk * 81), "search_prepare": nf * (120 + 80),
       "pose_refine": 52 * nf + 6480}
print("n_frames=%d reps=%d corners/frame=%.0f GN evaluations/job=%.1f" % (n, reps, nc, i_ia))
for k, (ms, launches) in sorted(t.items()):
    us = ms / launches * 1e3
    a = alg.get(k)
    gbs = (a * n / (us * 1e-6) / 1e9) if a else float("nan")
    print("  %-16s %8.1f us/launch  %7.2f us/frame  %8.1f GB/s algorithmic  (%.4f of 8 TB/s)" % (k, us, us / n, gbs, gbs / 8000))
