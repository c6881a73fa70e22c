#!/usr/bin/env python3
"""Generate tests/golden/front_end_golden.npz.

The reference (JdeRobot/slam-SDVL) has no tests, fixtures or golden vectors and cannot be compiled in this image
(needs OpenCV + Eigen + Pangolin), so these vectors are outputs of the CPU restatement (oracle/) on seeded
synthetic frames — they FREEZE the oracle (regressions show up as diffs) and give the HIP path fixed targets.
The independent numpy / scipy / libc cross-checks in tests/test_oracle_independent.py are what pins the oracle
itself.  Inputs are not stored: they are re-rendered by the seeded generator (slam-sdvl_amd/csrc/sdvl_synth.h) and
verified by SHA-256.

    python tests/golden/make_golden.py        # rewrites the fixture (run only when the oracle changes on purpose)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oraclelib import Oracle, Synth, TUM_CAM, TUM_DIST, quat_rot, trajectory_pose  # noqa: E402

W, H = 640, 480
FRAMES = [0, 3]


def inputs(orc, syn):
    return [syn.render(trajectory_pose(orc, k), TUM_CAM, W, H, seed=20260001, frame_id=k) for k in FRAMES]


def align_features(n, seed):
    rng = np.random.default_rng(seed)
    px = np.stack([rng.uniform(48, W - 48, n), rng.uniform(48, H - 48, n)], 1)
    ray = np.stack([(px[:, 0] - TUM_CAM[2]) / TUM_CAM[0], (px[:, 1] - TUM_CAM[3]) / TUM_CAM[1], np.ones(n)], 1)
    bearing = ray / np.linalg.norm(ray, axis=1, keepdims=True)
    depth = 2.0 / bearing[:, 2]
    valid = np.ones(n, np.uint8)
    valid[::17] = 0
    return px, bearing, depth, valid


def pose_matches(orc, n=150, seed=20260400):
    """matches of a small camera motion with 20 % gross outliers: rows ax, ay, px, py, pz, level; plus 100 rand() draws"""
    rng = np.random.default_rng(seed)
    true_pose = orc.se3_exp(np.array([0.04, -0.03, 0.015, 0.006, -0.008, 0.004]))
    P = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-0.9, 0.9, n), rng.uniform(1.5, 3.0, n)], 1)
    pc = P @ quat_rot(true_pose[:4]).T + true_pose[4:]
    a = pc[:, :2] / pc[:, 2:3] + rng.normal(0, 0.4 / TUM_CAM[0], (n, 2))
    bad = rng.random(n) < 0.2
    a[bad] += rng.uniform(-40, 40, (int(bad.sum()), 2)) / TUM_CAM[0]
    obs = np.concatenate([a, P, rng.integers(0, 3, n)[:, None].astype(np.float64)], 1)
    return obs, orc.se3_exp(np.zeros(6)), orc.rand_stream(100, seed=7)


def compute(orc, syn):
    img0, img3 = inputs(orc, syn)
    out = {"sha256_frame0": hashlib.sha256(img0.tobytes()).hexdigest(), "sha256_frame3": hashlib.sha256(img3.tobytes()).hexdigest()}
    pyr = orc.pyramid(img0, 5)
    out["pyr_sha256"] = np.array([hashlib.sha256(p.tobytes()).hexdigest() for p in pyr])
    out["pyr_level4"] = pyr[4]
    kps, offs, ran = orc.fast_cells(img0)
    out["fast_level0_kps"] = kps.astype(np.int16)
    out["fast_level0_offsets"] = offs
    corners = orc.detect_pyramid(img0)
    out["corners"] = corners.astype(np.int16)
    sel = corners[::31]
    out["shi_tomasi"] = np.array([orc.shi_tomasi(pyr[l], x, y) for x, y, l in sel])
    out["filtered"] = orc.filter_corners(img0, corners)
    descs, angs = [], []
    for x, y, l in sel:
        d, a = orc.orb_describe(pyr[l], [[x, y]])
        descs.append(d[0]); angs.append(a[0])
    out["orb_desc"] = np.array(descs, np.uint8)
    out["orb_angle"] = np.array(angs, np.float32)
    px, bearing, depth, valid = align_features(200, 20260200)
    r = orc.image_align(img0, img3, TUM_CAM, px, bearing, depth, valid, [1, 0, 0, 0, 0, 0, 0])
    out["align_T"] = r["T"]; out["align_n"] = r["n"]; out["align_error"] = r["error"]; out["align_its"] = r["its"]
    c0 = corners[corners[:, 2] == 0][:64]
    rng = np.random.default_rng(20260300)
    uv0 = c0[:, :2] + rng.uniform(-2, 2, (len(c0), 2))
    res = [orc.align_patch(img0, img0[y - 5:y + 5, x - 5:x + 5].reshape(-1), img0[y - 4:y + 4, x - 4:x + 4].reshape(-1), uv0[i])
           for i, (x, y, _) in enumerate(c0)]
    out["lk_uv0"] = uv0; out["lk_conv"] = np.array([r[0] for r in res], np.uint8); out["lk_uv"] = np.array([r[1] for r in res])
    trk = orc.tracker(W, H, TUM_CAM)
    poses, counts = [], []
    for k in range(6):
        st = trk.handle_frame(syn.render(trajectory_pose(orc, k), TUM_CAM, W, H, seed=20260001, frame_id=k))
        poses.append(list(st.pose)); counts.append([st.matches, st.attempts, st.inliers, st.n_corners, st.keyframe])
    trk.close()
    out["track_pose"] = np.array(poses); out["track_counts"] = np.array(counts, np.int32)
    # input stage: cv::undistort with the TUM fr1 coefficients
    und = orc.undistort(img0, TUM_CAM, TUM_DIST)
    out["undistort_sha256"] = hashlib.sha256(und.tobytes()).hexdigest()
    out["undistort_row240"] = und[240]
    # pose from matches: RANSAC + Tukey Gauss-Newton + rescue
    obs, guess, _ = pose_matches(orc)
    r = orc.pose_from_matches(TUM_CAM, obs, guess, rand_seed=7)
    out["pose_T"] = r["pose"]; out["pose_draws"] = r["n_draws"]
    out["pose_inliers"] = r["inliers"].astype(np.int32); out["pose_outliers"] = r["outliers"].astype(np.int32)
    # closed loop with the reference's mapper (sequential mode)
    trk = orc.tracker(W, H, TUM_CAM)
    trk.use_mapper(True)
    poses, counts, mstats = [], [], []
    for k in range(14):
        st = trk.handle_frame(syn.render(trajectory_pose(orc, k), TUM_CAM, W, H, seed=20260001, frame_id=k))
        ms = trk.map_stats()
        poses.append(list(st.pose)); counts.append([st.matches, st.attempts, st.inliers, st.n_corners, st.keyframe])
        mstats.append([ms[q] for q in ("candidates", "converged", "initialized", "linked", "connected", "keyframes")])
    trk.close()
    out["mapper_pose"] = np.array(poses); out["mapper_counts"] = np.array(counts, np.int32); out["mapper_stats"] = np.array(mstats, np.int32)
    return out


if __name__ == "__main__":
    data = compute(Oracle(), Synth())
    path = os.path.join(HERE, "front_end_golden.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes")
