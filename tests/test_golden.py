"""Golden vectors (tests/golden/front_end_golden.npz, written by tests/golden/make_golden.py): the CPU oracle must keep
reproducing them (CPU suite), and the HIP path must hit the same targets through the C-ABI (GPU suite)."""
import hashlib
import importlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402
from oraclelib import TUM_CAM  # noqa: E402


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "front_end_golden.npz"), allow_pickle=False)


def check(got, gold, tol_keys=("align_T", "align_error", "track_pose", "pose_T", "mapper_pose")):
    for k in gold.files:
        g, w = np.asarray(got[k]), gold[k]
        if k in tol_keys:
            assert np.allclose(g, w, rtol=0, atol=1e-9), k
        else:
            assert np.array_equal(g, w), k


def test_oracle_reproduces_golden(orc, synth, golden):
    check(mg.compute(orc, synth), golden)


@pytest.mark.gpu
def test_hip_path_hits_golden(orc, synth, golden):
    sdvl = importlib.import_module("slam-sdvl_amd")
    trk = importlib.import_module("slam-sdvl_amd.tracker")
    img0, img3 = mg.inputs(orc, synth)
    assert hashlib.sha256(img0.tobytes()).hexdigest() == str(golden["sha256_frame0"])
    ctx = sdvl.Context(0)
    f0, f3 = ctx.frame(img0), ctx.frame(img3)
    for l in range(5):
        assert hashlib.sha256(f0.level(l).tobytes()).hexdigest() == str(golden["pyr_sha256"][l])
    (kps, offs), = ctx.fast_cells([f0], sdvl.default_detect_params())[0]
    k0 = kps[kps[:, 3] == 0]
    assert np.array_equal(k0[:, :3], golden["fast_level0_kps"].astype(np.int32))
    corners = ctx.detect_corners([f0], sdvl.default_detect_params(), 1000)[0]
    assert np.array_equal(corners, golden["corners"].astype(np.int32))
    ctx.orb_describe([f0], want=False)
    sel = corners[::31]
    assert np.array_equal(ctx.shi_tomasi([f0])[0][::31], golden["shi_tomasi"])
    d, a = ctx.orb_describe_points(f0, sel)
    assert np.array_equal(d, golden["orb_desc"]) and np.array_equal(a, golden["orb_angle"])
    px, bearing, depth, valid = mg.align_features(200, 20260200)
    feats = (sdvl.AlignFeature * 200)()
    for i in range(200):
        feats[i].px, feats[i].py = px[i]
        feats[i].fx, feats[i].fy, feats[i].fz = bearing[i]
        feats[i].depth, feats[i].valid = depth[i], int(valid[i])
    r = ctx.image_align([(f0, f3, 0, 200, [1, 0, 0, 0, 0, 0, 0])], feats, sdvl.Camera(640, 480, *TUM_CAM), sdvl.default_align_params())[0]
    assert np.abs(np.array(r.T[:]) - golden["align_T"]).max() <= 1e-4 and r.n_meas == int(golden["align_n"])
    c0 = corners[corners[:, 2] == 0][:64]
    border = np.stack([img0[y - 5:y + 5, x - 5:x + 5].reshape(-1) for x, y, _ in c0])
    patch = np.stack([img0[y - 4:y + 4, x - 4:x + 4].reshape(-1) for x, y, _ in c0])
    uv, conv, _ = ctx.align_patches([f0] * len(c0), np.zeros(len(c0), np.int32), border, patch, golden["lk_uv0"])
    assert np.array_equal(conv, golden["lk_conv"]) and np.array_equal(uv, golden["lk_uv"])
    f0.close(); f3.close(); ctx.close()
    trk.configure()
    dev = trk.HostDevice(0)
    batch = trk.TrackerBatch(dev, 1, 640, 480, TUM_CAM)
    from oraclelib import trajectory_pose
    for k in range(6):
        st = batch.step_host([synth.render(trajectory_pose(orc, k), TUM_CAM, 640, 480, seed=20260001, frame_id=k)])[0]
        assert [st.matches, st.attempts, st.inliers, st.n_corners, st.keyframe] == golden["track_counts"][k].tolist()
        assert np.abs(np.array(st.pose[:]) - golden["track_pose"][k]).max() <= 1e-4
    batch.close(); dev.close()
    # input stage, pose stage and the mapper loop
    from oraclelib import TUM_DIST
    ctx = sdvl.Context(0)
    und = ctx.undistort([img0], sdvl.Camera(640, 480, *TUM_CAM), TUM_DIST)[0]
    assert hashlib.sha256(und.tobytes()).hexdigest() == str(golden["undistort_sha256"])
    obs, guess, draws = mg.pose_matches(orc)
    r = ctx.pose_from_matches([(obs, guess, draws)], fx=TUM_CAM[0])[0]
    assert r["n_draws"] == int(golden["pose_draws"]) and np.array_equal(r["inliers"], golden["pose_inliers"])
    assert np.array_equal(r["outliers"], golden["pose_outliers"]) and np.abs(r["pose"] - golden["pose_T"]).max() <= 1e-9
    ctx.close()
    trk.set_mapper(True)
    try:
        dev = trk.HostDevice(0)
        batch = trk.TrackerBatch(dev, 1, 640, 480, TUM_CAM)
    finally:
        trk.set_mapper(False)
    for k in range(14):
        st = batch.step_host([synth.render(trajectory_pose(orc, k), TUM_CAM, 640, 480, seed=20260001, frame_id=k)])[0]
        assert [st.matches, st.attempts, st.inliers, st.n_corners, st.keyframe] == golden["mapper_counts"][k].tolist()
        assert np.abs(np.array(st.pose[:]) - golden["mapper_pose"][k]).max() <= 1e-4
        ms = batch.map_stats(0)
        assert [ms[q] for q in ("candidates", "converged", "initialized", "linked", "connected", "keyframes")] == golden["mapper_stats"][k].tolist()
    batch.close(); dev.close()
