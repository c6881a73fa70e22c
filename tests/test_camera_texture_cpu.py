"""The round-5 inputs, checked on the CPU: the camera-like texture (csrc/sdvl_synth.h SDVL_TEXTURE_CAMERA) has the statistics
VERDICT r04 asked for — 2-5 k FAST-10 keypoints per 640x480 frame over the three detection levels, corners on a few % of the
pixels instead of every second one, the selection quota (about 1000 corners) filled, the oracle tracking it with >= 150 matches per
frame — and S-B's chunks (752x480, config_euroc.cfg, seeds 20260010..13) shard as SURVEY §8e says."""
import importlib

import numpy as np

from oraclelib import EUROC_CAM, TUM_CAM, trajectory_pose


def test_camera_texture_has_camera_like_corner_statistics(orc, synth):
    img = synth.render(trajectory_pose(orc, 0), TUM_CAM, 640, 480, texture=1)
    pyr = orc.pyramid(img)
    kps = [len(orc.fast_cells(pyr[l])[0]) for l in range(3)]
    assert 2000 <= sum(kps) <= 5000, kps
    raw0 = len(orc.fast(pyr[0], 10, False))
    assert raw0 < 0.12 * pyr[0].size, raw0                       # the value-noise plane: 0.48
    corners = orc.detect_pyramid(img)
    assert 950 <= len(corners) <= 1100, len(corners)            # quotas 395 / 329 / 274: level 2 falls a few short on this seed
    plane = synth.render(trajectory_pose(orc, 0), TUM_CAM, 640, 480, texture=0)
    assert len(orc.fast(plane, 10, False)) > 3 * raw0            # what the round-1..4 texture does to the same detector


def test_oracle_tracks_the_camera_texture(orc, synth):
    trk = orc.tracker(640, 480, TUM_CAM)
    matches = []
    for k in range(24):
        st = trk.handle_frame(synth.render(trajectory_pose(orc, k), TUM_CAM, 640, 480, frame_id=k, texture=1))
        if k > 0:
            assert st.quality == 0, k
            matches.append(st.matches)
    trk.close()
    assert np.mean(matches) >= 150, np.mean(matches)


def test_s_b_chunks_track_at_euroc_size(orc, synth):
    old = orc.params.min_matches
    orc.params.min_matches = 5                                   # config/config_euroc.cfg:42
    try:
        for seed in (20260010, 20260013):
            trk = orc.tracker(752, 480, EUROC_CAM)
            for k in range(8):
                st = trk.handle_frame(synth.render(trajectory_pose(orc, k), EUROC_CAM, 752, 480, seed=seed, frame_id=k, texture=1))
                if k > 0:
                    assert st.quality == 0 and st.matches >= 100, (seed, k, st.matches)
            trk.close()
    finally:
        orc.params.min_matches = old


def test_chunk_sharding_follows_survey_8e():
    shard = importlib.import_module("slam-sdvl_amd.shard")
    per = 8
    # one rank: every chunk, round-robin over its sequences
    assert [shard.chunk_for_sequence(g, per, 1) for g in range(per)] == [0, 1, 2, 3, 0, 1, 2, 3]
    # two ranks: rank g owns the chunks {c : c mod 2 = g}
    assert {shard.chunk_for_sequence(g, per, 2) for g in range(per)} == {0, 2}
    assert {shard.chunk_for_sequence(g, per, 2) for g in range(per, 2 * per)} == {1, 3}
    # four ranks: one chunk each (BASELINE config 4)
    for r in range(4):
        assert {shard.chunk_for_sequence(g, per, 4) for g in range(r * per, (r + 1) * per)} == {r}
    # eight ranks: chunk r mod 4
    assert [shard.chunk_for_sequence(r * per, per, 8) for r in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
    # every chunk is covered whatever the rank count
    for world in (1, 2, 3, 4, 8):
        assert {shard.chunk_for_sequence(g, per, world) for g in range(world * per)} == {0, 1, 2, 3}, world


def test_kernel_source_stamp_is_gits_blob_hash(tmp_path):
    import subprocess
    stamp = importlib.import_module("slam-sdvl_amd.stamp")
    st = stamp.kernel_source_stamp()
    assert "sdvl_detect.hip" in st["files"] and len(st["combined"]) == 40
    f = stamp.kernel_source_files()[0]
    try:
        want = subprocess.run(["git", "hash-object", f], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        return                                                   # no git here: the format is checked where there is one
    assert stamp.blob_hash(f) == want
