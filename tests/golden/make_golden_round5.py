#!/usr/bin/env python3
"""Generate tests/golden/round5_golden.npz: the round-5 inputs frozen the same way front_end_golden.npz freezes round 1's —
the camera-like texture (csrc/sdvl_synth.h, SDVL_TEXTURE_CAMERA: SHA-256 of rendered frames, so that the generator cannot drift),
what the oracle detects and tracks on it at S-A's and S-B's geometry (seeds 20260010..13, config_euroc.cfg), and the
plane-map stub's max_keyframes policy.  Outputs of the CPU restatement (parity unpinned: the reference has no vectors of its own).

    python tests/golden/make_golden_round5.py      # rewrites the fixture (run only when the oracle or the texture change on purpose)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oraclelib import EUROC_CAM, Oracle, Synth, TUM_CAM, trajectory_pose  # noqa: E402

N_TRACK, N_SB, N_CULL = 10, 6, 60


def frame(orc, syn, k, cam=TUM_CAM, w=640, h=480, seed=20260001):
    return syn.render(trajectory_pose(orc, k), cam, w, h, seed=seed, frame_id=k, texture=1)


def counts(st):
    return [st.matches, st.attempts, st.inliers, st.outliers, st.n_corners, st.keyframe, st.quality]


def compute(orc, syn):
    out = {}
    img0 = frame(orc, syn, 0)
    out["sha256_camera_frame0"] = hashlib.sha256(img0.tobytes()).hexdigest()
    out["sha256_camera_frame7"] = hashlib.sha256(frame(orc, syn, 7).tobytes()).hexdigest()
    pyr = orc.pyramid(img0, 5)
    out["fast_counts"] = np.array([len(orc.fast_cells(pyr[l])[0]) for l in range(3)], np.int32)
    out["fast_level1_kps"] = orc.fast_cells(pyr[1])[0].astype(np.int16)
    out["corners"] = orc.detect_pyramid(img0).astype(np.int16)
    trk = orc.tracker(640, 480, TUM_CAM)
    tc, tp = [], []
    for k in range(N_TRACK):
        st = trk.handle_frame(frame(orc, syn, k))
        tc.append(counts(st)); tp.append(list(st.pose[:]))
    trk.close()
    out["track_counts"], out["track_pose"] = np.array(tc, np.int32), np.array(tp)
    # S-B: the four chunks at EuRoC's geometry, min_matches 5
    old = orc.params.min_matches
    orc.params.min_matches = 5
    try:
        sb_sha, sb_c, sb_p = [], [], []
        for seed in (20260010, 20260011, 20260012, 20260013):
            f0 = frame(orc, syn, 0, EUROC_CAM, 752, 480, seed)
            sb_sha.append(hashlib.sha256(f0.tobytes()).hexdigest())
            trk = orc.tracker(752, 480, EUROC_CAM)
            c, p = [], []
            for k in range(N_SB):
                st = trk.handle_frame(frame(orc, syn, k, EUROC_CAM, 752, 480, seed))
                c.append(counts(st)); p.append(list(st.pose[:]))
            trk.close()
            sb_c.append(c); sb_p.append(p)
        out["sb_sha256_frame0"] = np.array(sb_sha)
        out["sb_counts"], out["sb_pose"] = np.array(sb_c, np.int32), np.array(sb_p)
    finally:
        orc.params.min_matches = old
    # the plane-map stub with SDVL.max_keyframes 8 (PlaneLimitKeyframes): decisions over 60 frames, ~11 keyframes made
    trk = orc.tracker(640, 480, TUM_CAM)
    trk.set_max_keyframes(8)
    cc, cp = [], []
    for k in range(N_CULL):
        st = trk.handle_frame(frame(orc, syn, k))
        cc.append(counts(st)); cp.append(list(st.pose[:]))
    trk.close()
    out["cull_counts"], out["cull_pose"] = np.array(cc, np.int32), np.array(cp)
    return out


if __name__ == "__main__":
    res = compute(Oracle(), Synth())
    np.savez_compressed(os.path.join(HERE, "round5_golden.npz"), **res)
    print("wrote round5_golden.npz:", {k: np.asarray(v).shape for k, v in res.items()})
