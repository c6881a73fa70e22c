"""Golden vectors of the round-5 inputs (tests/golden/round5_golden.npz, written by tests/golden/make_golden_round5.py): the camera-like
texture's bytes, what the oracle detects and tracks on it at S-A's and S-B's geometry, and the plane-map stub's max_keyframes policy.
The CPU oracle must keep reproducing them (CPU suite); the HIP path must hit the same targets (GPU suite)."""
import hashlib
import importlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_round5 as mg  # noqa: E402
from oraclelib import EUROC_CAM, TUM_CAM  # noqa: E402

TOL_KEYS = ("track_pose", "sb_pose", "cull_pose")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "round5_golden.npz"), allow_pickle=False)


def test_oracle_and_texture_reproduce_golden(orc, synth, golden):
    got = mg.compute(orc, synth)
    for k in golden.files:
        g, w = np.asarray(got[k]), golden[k]
        if k in TOL_KEYS:
            assert np.allclose(g, w, rtol=0, atol=1e-9), k
        else:
            assert np.array_equal(g, w), k
    assert golden["cull_counts"][:, 5].sum() >= 10      # well past the cap of 8 keyframes


@pytest.mark.gpu
def test_hip_path_hits_round5_golden(orc, synth, golden):
    sdvl = importlib.import_module("slam-sdvl_amd")
    trk = importlib.import_module("slam-sdvl_amd.tracker")
    img0 = mg.frame(orc, synth, 0)
    assert hashlib.sha256(img0.tobytes()).hexdigest() == str(golden["sha256_camera_frame0"])
    # the device generator renders the same bytes as the host one
    import bench as B
    ctx = sdvl.Context(0)
    old = B.TEXTURE
    B.TEXTURE = B.TEXTURES["camera"]
    try:
        from oraclelib import trajectory_pose
        view = B.make_view(sdvl, trajectory_pose(orc, 7), 20260001, 7)
    finally:
        B.TEXTURE = old
    buf = ctx.device_malloc(640 * 480)
    ctx.synth_render([view], 640, 480, buf)
    assert hashlib.sha256(ctx.device_download(buf, 640 * 480).tobytes()).hexdigest() == str(golden["sha256_camera_frame7"])
    ctx.device_free(buf)
    f0 = ctx.frame(img0)
    (kps, _), = ctx.fast_cells([f0], sdvl.default_detect_params())[0]
    assert [int((kps[:, 3] == l).sum()) for l in range(3)] == golden["fast_counts"].tolist()
    assert np.array_equal(kps[kps[:, 3] == 1][:, :3], golden["fast_level1_kps"].astype(np.int32))
    assert np.array_equal(ctx.detect_corners([f0], sdvl.default_detect_params(), 1000)[0], golden["corners"].astype(np.int32))
    f0.close(); ctx.close()

    def run(batch, frames, want_counts, want_pose):
        for k, im in enumerate(frames):
            st = batch.step_host([im])[0]
            assert mg.counts(st) == want_counts[k].tolist(), k
            assert np.abs(np.array(st.pose[:]) - want_pose[k]).max() <= 1e-4, k

    trk.configure()
    dev = trk.HostDevice(0)
    batch = trk.TrackerBatch(dev, 1, 640, 480, TUM_CAM)
    run(batch, [mg.frame(orc, synth, k) for k in range(mg.N_TRACK)], golden["track_counts"], golden["track_pose"])
    batch.close()
    over = dict(trk.TUM_OVERRIDES)
    over["SDVL.min_matches"] = 5
    trk.configure(over)
    try:
        for i, seed in enumerate((20260010, 20260011, 20260012, 20260013)):
            batch = trk.TrackerBatch(dev, 1, 752, 480, EUROC_CAM)
            run(batch, [mg.frame(orc, synth, k, EUROC_CAM, 752, 480, seed) for k in range(mg.N_SB)], golden["sb_counts"][i], golden["sb_pose"][i])
            batch.close()
        over = dict(trk.TUM_OVERRIDES)
        over["SDVL.max_keyframes"] = 8
        trk.configure(over)
        batch = trk.TrackerBatch(dev, 1, 640, 480, TUM_CAM)
        run(batch, [mg.frame(orc, synth, k) for k in range(mg.N_CULL)], golden["cull_counts"], golden["cull_pose"])
        batch.close()
    finally:
        trk.configure()
        dev.close()
