"""Closed-loop parity (BASELINE.json config 3: "full HIP front-end ... pose-delta tolerance check"): B synthetic
sequences tracked on the MI355X through the host layer (sdvl::SDVLBatch) against the CPU oracle tracker run on
the same frames.  Decisions (matches, attempts, inliers, keyframes) must be identical, poses within 1e-4."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

from oraclelib import TUM_CAM, XI, trajectory_pose

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-4


@pytest.fixture(scope="module")
def trk():
    importlib.import_module("slam-sdvl_amd")
    return importlib.import_module("slam-sdvl_amd.tracker")


def run_case(trk, orc, synth, B, n_frames, threads, w=640, h=480, cam=TUM_CAM):
    trk.configure()
    dev = trk.HostDevice(0)
    xis = [XI * (1.0 + 0.15 * i) * (1 if i % 2 == 0 else -1) for i in range(B)]
    seeds = [20260001 + i for i in range(B)]
    batch = trk.TrackerBatch(dev, B, w, h, cam, host_threads=threads)
    oracles = [orc.tracker(w, h, cam) for _ in range(B)]
    worst = 0.0
    for k in range(n_frames):
        imgs = [synth.render(trajectory_pose(orc, k, xis[i]), cam, w, h, seed=seeds[i], frame_id=k) for i in range(B)]
        got = batch.step_host(imgs)
        for i in range(B):
            want = oracles[i].handle_frame(imgs[i])
            g = got[i]
            assert (g.state, g.quality, g.keyframe, g.n_corners) == (want.state, want.quality, want.keyframe, want.n_corners), (k, i)
            assert (g.matches, g.attempts, g.inliers, g.outliers) == (want.matches, want.attempts, want.inliers, want.outliers), (k, i)
            assert g.align_meas == want.align_meas, (k, i)
            d = np.abs(np.array(g.pose[:]) - np.array(want.pose[:])).max()
            worst = max(worst, d)
            assert d <= POSE_TOL, (k, i, d)
            if k > 0:
                assert g.quality == 0 and g.matches >= 100       # the sequence is actually being tracked
    batch.close()
    for o in oracles:
        o.close()
    dev.close()
    return worst


def test_single_sequence_closed_loop(trk, orc, synth):
    assert trk.device_pose()      # RANSAC + pose refinement run in sdvl_pose_from_matches by default
    worst = run_case(trk, orc, synth, B=1, n_frames=14, threads=1)
    assert worst <= POSE_TOL


def test_closed_loop_with_host_pose_stage(trk, orc, synth):
    """the host implementation of the pose stage stays selectable and gives the same decisions"""
    trk.set_device_pose(False)
    try:
        worst = run_case(trk, orc, synth, B=3, n_frames=7, threads=2)
    finally:
        trk.set_device_pose(True)
    assert worst <= POSE_TOL


def test_batch_of_sequences_closed_loop_threaded(trk, orc, synth):
    """several host threads per batch: on the device-resident tables (table rebuilds, results, keyframes spread over the
    threads) and on the host-driven path (requests assembled by the threads)"""
    worst = run_case(trk, orc, synth, B=5, n_frames=12, threads=4)
    assert worst <= POSE_TOL
    trk.set_track_tables(False)
    try:
        worst = run_case(trk, orc, synth, B=5, n_frames=9, threads=4)
    finally:
        trk.set_track_tables(True)
    assert worst <= POSE_TOL


def test_euroc_size_closed_loop(trk, orc, synth):
    from oraclelib import EUROC_CAM
    worst = run_case(trk, orc, synth, B=2, n_frames=8, threads=2, w=752, h=480, cam=EUROC_CAM)
    assert worst <= POSE_TOL


def test_lost_and_relocalize_matches_oracle(trk, orc, synth):
    """three featureless frames -> TRACKING_BAD x3 -> Relocalize over the keyframes (sdvl.cc:73-89,205-238)"""
    trk.configure()
    dev = trk.HostDevice(0)
    batch = trk.TrackerBatch(dev, 1, 640, 480, TUM_CAM)
    ref = orc.tracker(640, 480, TUM_CAM)
    seq = [0, 1, 2, 3, 4, 5, -1, -1, -1, 5, 6, 7]
    relocs = 0
    for k, idx in enumerate(seq):
        if idx < 0:
            img = np.full((480, 640), 127, np.uint8)
        else:
            img = synth.render(trajectory_pose(orc, idx), TUM_CAM, 640, 480, frame_id=idx)
        g = batch.step_host([img])[0]
        w = ref.handle_frame(img)
        assert g.host_path == 0, (k, idx)      # relocalisation included: the step never leaves the device-resident tables
        assert (g.quality, g.matches, g.attempts, g.inliers, g.keyframe, g.relocalized) == \
               (w.quality, w.matches, w.attempts, w.inliers, w.keyframe, w.relocalized), (k, idx)
        if idx >= 0:   # on a featureless frame the alignment is degenerate (its pose is discarded: TRACKING_BAD keeps last_frame_)
            assert np.abs(np.array(g.pose[:]) - np.array(w.pose[:])).max() <= POSE_TOL, (k, idx)
        relocs += g.relocalized
    assert relocs == 1
    batch.close(); ref.close(); dev.close()


def test_mixed_batch_lost_trackers_relocalize_inside_the_tabled_step(trk, orc, synth):
    """a batch of 16 where trackers 3 and 11 are blinded for 5 frames.  Each of them goes TRACKING_BAD x3, then
    Relocalize (sdvl.cc:73-89,205-238) — its alignments against its keyframes and its reloc searches are extra launches of the SAME
    step, the other 14 stay on their device-resident tables; nobody ever enters the host-driven form (FrameStats.host_path stays 0).
    Every tracker equals its own oracle at every frame; the two blinded ones relocalise once each."""
    trk.configure()
    B, n_frames = 16, 20
    blind = {3: range(8, 13), 11: range(9, 14)}      # overlapping, not identical: rounds with one and with two lost trackers
    dev = trk.HostDevice(0)
    xis = [XI * (1.0 + 0.1 * i) * (1 if i % 2 == 0 else -1) for i in range(B)]
    seeds = [20260001 + i for i in range(B)]
    batch = trk.TrackerBatch(dev, B, 640, 480, TUM_CAM)
    oracles = [orc.tracker(640, 480, TUM_CAM) for _ in range(B)]
    relocs = {3: 0, 11: 0}
    shown = [0] * B            # the trajectory index a tracker is shown: a blinded tracker resumes where it was blinded
    for k in range(n_frames):
        imgs = []
        for i in range(B):
            if i in blind and k in blind[i]:
                imgs.append(np.full((480, 640), 127, np.uint8))
            else:
                imgs.append(synth.render(trajectory_pose(orc, shown[i], xis[i]), TUM_CAM, 640, 480, seed=seeds[i], frame_id=shown[i]))
                shown[i] += 1
        got = batch.step_host(imgs)
        for i in range(B):
            w, g = oracles[i].handle_frame(imgs[i]), got[i]
            assert g.host_path == 0, (k, i, g.host_path)
            assert (g.state, g.quality, g.matches, g.inliers, g.keyframe, g.relocalized, g.n_corners) == \
                   (w.state, w.quality, w.matches, w.inliers, w.keyframe, w.relocalized, w.n_corners), (k, i)
            # On a featureless frame the alignment is degenerate: its Gauss-Newton steps are decided by the rounding of a chi2 that does
            # not depend on the pose, the pose wanders (and is discarded: TRACKING_BAD keeps last_frame_), and with it the number of points
            # that project into the image = the attempts (tools/mix_probe.py: 265 against 266 with this tracker blinded ALONE).
            # Everything else of a blinded frame, and every field of every other frame, is compared.
            if not (i in blind and k in blind[i]):
                assert g.attempts == w.attempts, (k, i)
                assert np.abs(np.array(g.pose[:]) - np.array(w.pose[:])).max() <= POSE_TOL, (k, i)
            if i in relocs:
                relocs[i] += g.relocalized
            elif k > 0:
                assert g.quality == 0 and g.matches >= 100
    assert relocs == {3: 1, 11: 1}
    batch.close()
    for o in oracles:
        o.close()
    dev.close()


def test_raw_camera_frames_through_the_lens_undistorted_inside_the_step(trk, orc, synth):
    """frames rendered THROUGH config_tum_f1.cfg's lens are handed to the batch as they come from the camera
    (TrackerBatch.set_distortion: Image::raw); Camera::UndistortImage (camera.cc:100-105, main.cc:133) runs inside the step, fused into the
    frames' upload, from a map that is built ONCE.  The oracle is fed its own cv::undistort of the same bytes: identical decisions."""
    from oraclelib import TUM_DIST
    trk.configure()
    B, n_frames = 3, 9
    dev = trk.HostDevice(0)
    xis = [XI * (1.0 + 0.15 * i) * (1 if i % 2 == 0 else -1) for i in range(B)]
    seeds = [20260001 + i for i in range(B)]
    batch = trk.TrackerBatch(dev, B, 640, 480, TUM_CAM)
    batch.set_distortion(TUM_DIST)
    oracles = [orc.tracker(640, 480, TUM_CAM) for _ in range(B)]
    pkg = importlib.import_module("slam-sdvl_amd")
    lib = pkg.load_library()
    for k in range(n_frames):
        raw = [synth.render(trajectory_pose(orc, k, xis[i]), TUM_CAM, 640, 480, seed=seeds[i], frame_id=k, dist=TUM_DIST) for i in range(B)]
        got = batch.step_host(raw)
        for i in range(B):
            w, g = oracles[i].handle_frame(orc.undistort(raw[i], TUM_CAM, TUM_DIST)), got[i]
            assert (g.state, g.quality, g.keyframe, g.n_corners, g.matches, g.attempts, g.inliers, g.outliers, g.align_meas, g.host_path) == \
                   (w.state, w.quality, w.keyframe, w.n_corners, w.matches, w.attempts, w.inliers, w.outliers, w.align_meas, 0), (k, i)
            assert np.abs(np.array(g.pose[:]) - np.array(w.pose[:])).max() <= POSE_TOL, (k, i)
            if k > 0:
                assert g.quality == 0 and g.matches >= 100       # the undistorted view is the pinhole view again: tracked as usual
    counters = (C.c_int64 * 4)()
    lib.sdvl_ctx_counters.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.sdvl_ctx_counters(dev.ctx_handle(), counters) == 0
    assert counters[2] == 1, counters[2]                          # one map for the camera, not one per call
    batch.close()
    for o in oracles:
        o.close()
    dev.close()


def run_mapper_case(trk, orc, synth, B, n_frames, threads):
    """closed loop with the reference's mapper (map.cc, sequential mode) on both sides: the tracker's per-frame decisions,
    the poses AND the map bookkeeping (candidates alive / converged / initialised / linked, connections, keyframes after
    culling) must agree frame by frame"""
    trk.configure()
    trk.set_mapper(True)
    try:
        dev = trk.HostDevice(0)
        xis = [XI * (1.0 + 0.2 * i) * (1 if i % 2 == 0 else -1) for i in range(B)]
        seeds = [20260001 + i for i in range(B)]
        batch = trk.TrackerBatch(dev, B, 640, 480, TUM_CAM, host_threads=threads)
    finally:
        trk.set_mapper(False)
    oracles = [orc.tracker(640, 480, TUM_CAM) for _ in range(B)]
    for o in oracles:
        o.use_mapper(True)
    worst = 0.0
    stats = None
    for k in range(n_frames):
        imgs = [synth.render(trajectory_pose(orc, k, xis[i]), TUM_CAM, 640, 480, seed=seeds[i], frame_id=k) for i in range(B)]
        got = batch.step_host(imgs)
        for i in range(B):
            want = oracles[i].handle_frame(imgs[i])
            g = got[i]
            assert (g.state, g.quality, g.keyframe, g.n_corners) == (want.state, want.quality, want.keyframe, want.n_corners), (k, i)
            assert (g.matches, g.attempts, g.inliers, g.outliers) == (want.matches, want.attempts, want.inliers, want.outliers), (k, i)
            assert g.align_meas == want.align_meas, (k, i)
            d = np.abs(np.array(g.pose[:]) - np.array(want.pose[:])).max()
            worst = max(worst, d)
            assert d <= POSE_TOL, (k, i, d)
            stats = batch.map_stats(i)
            assert stats == oracles[i].map_stats(), (k, i, stats, oracles[i].map_stats())
            if k > 0:
                assert g.quality == 0 and g.matches >= 100
    batch.close()
    for o in oracles:
        o.close()
    dev.close()
    return worst, stats


def test_closed_loop_with_reference_mapper(trk, orc, synth):
    worst, stats = run_mapper_case(trk, orc, synth, B=1, n_frames=32, threads=1)
    assert worst <= POSE_TOL
    # the map is really being built by triangulation + depth filter, and keyframes get connected
    assert stats["initialized"] >= 100 and stats["converged"] >= 50 and stats["linked"] >= 200 and stats["connected"] >= 10


def test_closed_loop_with_reference_mapper_depth_filter_on_the_host(trk, orc, synth):
    """the same with Point::Update / HasConverged computed by the host layer (SDVL_HOST_DEPTH_FILTER=1): the tracking tables are
    rebuilt from the objects after every mapper update instead of being patched by depth_filter_kernel"""
    trk.set_device_filter(False)
    try:
        worst, stats = run_mapper_case(trk, orc, synth, B=2, n_frames=24, threads=1)
    finally:
        trk.set_device_filter(True)
    assert worst <= POSE_TOL and stats["converged"] >= 20


def test_closed_loop_with_reference_mapper_batched(trk, orc, synth):
    worst, _ = run_mapper_case(trk, orc, synth, B=4, n_frames=16, threads=3)
    assert worst <= POSE_TOL


def test_farm_fibers_give_identical_results(trk, orc, synth):
    """TrackerFarm with cooperative group-steps (a worker thread interleaves two groups, switching at GPU waits) must
    produce exactly the per-frame results of the plain one-step-at-a-time farm"""
    import importlib
    sdvl = importlib.import_module("slam-sdvl_amd")
    import bench as B
    G, Bg, n_steps = 4, 3, 7
    n = G * Bg
    trk.configure()

    def run(fibers, workers):
        farm = trk.TrackerFarm(0, G, Bg, 640, 480, TUM_CAM)
        farm.set_fibers(fibers)
        ctx = B.CtxView(sdvl, farm.ctx_handle(0))
        fb = 640 * 480
        buf = ctx.malloc(n * n_steps * fb)
        for k in range(n_steps):
            views = [B.make_view(sdvl, trajectory_pose(orc, k, XI * (1.0 + 0.1 * i)), 20260001 + i, k) for i in range(n)]
            ctx.render(views, buf + k * n * fb)
        ptrs = (buf + (np.arange(n_steps, dtype=np.uint64)[:, None] * n + np.arange(n, dtype=np.uint64)[None, :]) * fb).astype(np.uint64)
        st = farm.run(ptrs, workers)
        out = [(s.state, s.quality, s.keyframe, s.n_corners, s.matches, s.attempts, s.inliers, s.outliers, s.align_meas, tuple(s.pose[:])) for s in st]
        farm.close()
        return out

    plain = run(1, 4)
    coop = run(2, 2)
    assert plain == coop
    assert all(r[1] == 0 and r[4] >= 100 for r in plain[n:])      # tracking, not idling


def test_transient_ring_images_equal_resident_images(trk, orc, synth):
    """frames that alias a 2-slot ring for the step that tracks them (sdvlh_batch_step_device_transient: the slot is overwritten two
    steps later; only the frames that became keyframes copied their image out, Frame::OwnImages) give the per-frame results of frames
    that alias images resident for good — every decision and every pose bit, over enough steps for keyframes to be created, searched
    against and outlive several rewrites of their slot"""
    import ctypes as C
    import importlib
    sdvl = importlib.import_module("slam-sdvl_amd")
    import bench as B
    n, n_steps = 4, 12
    fb = 640 * 480
    trk.configure()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def run(mode):
        dev = trk.HostDevice(0)
        batch = trk.TrackerBatch(dev, n, 640, 480, TUM_CAM)
        ctx = B.CtxView(sdvl, dev.ctx_handle())
        buf = ctx.malloc(n * n_steps * fb)
        ring = ctx.malloc(2 * n * fb)
        for k in range(n_steps):
            views = [B.make_view(sdvl, trajectory_pose(orc, k, XI * (1.0 + 0.1 * i)), 20260001 + i, k) for i in range(n)]
            ctx.render(views, buf + k * n * fb)
        stream = C.c_void_p(ctx.lib.sdvl_ctx_stream(ctx.h))
        out = []
        for k in range(n_steps):
            if mode == "resident":
                st = batch.step_device([buf + (k * n + i) * fb for i in range(n)])
            else:
                slot = ring + (k % 2) * n * fb
                assert hip.hipMemcpyAsync(slot, buf + k * n * fb, n * fb, 3, stream) == 0   # device to device, on the batch's stream
                st = batch.step_device_transient([slot + i * fb for i in range(n)])
            out += [(s.state, s.quality, s.keyframe, s.n_corners, s.matches, s.attempts, s.inliers, s.outliers, s.align_meas, tuple(s.pose[:])) for s in st]
        batch.close()
        dev.close()
        return out

    resident = run("resident")
    assert run("ring") == resident
    assert sum(r[2] for r in resident[n:]) >= n           # keyframes were created after the bootstrap
    assert all(r[1] == 0 and r[4] >= 100 for r in resident[n:])


@pytest.mark.parametrize("mapper", [False, True])
def test_look_ahead_of_the_next_images_changes_nothing(trk, orc, synth, mapper):
    """SDVLBatch::SetNextImages: the next step's pyramids and corner detection queued behind the current step's chain (what the farm
    does for frames resident in HBM) — the per-frame results equal those of steps that build their frames themselves, bit for bit;
    a look-ahead of OTHER images than the step then brings is dropped, one announced and never used is dropped too"""
    import importlib
    sdvl = importlib.import_module("slam-sdvl_amd")
    import bench as B
    n, n_steps = 4, 12
    fb = 640 * 480
    trk.configure()

    def run(mode):
        dev = trk.HostDevice(0)
        trk.set_mapper(mapper)   # True: the reference's mapper in every step (its searches and the depth filter queue behind the look-ahead)
        try:
            batch = trk.TrackerBatch(dev, n, 640, 480, TUM_CAM)
        finally:
            trk.set_mapper(False)
        ctx = B.CtxView(sdvl, dev.ctx_handle())
        buf = ctx.malloc(n * n_steps * fb)
        for k in range(n_steps):
            views = [B.make_view(sdvl, trajectory_pose(orc, k, XI * (1.0 + 0.1 * i)), 20260001 + i, k) for i in range(n)]
            ctx.render(views, buf + k * n * fb)
        ptrs = lambda k: [buf + (k * n + i) * fb for i in range(n)]
        out = []
        for k in range(n_steps):
            if k + 1 < n_steps:
                if mode == "ahead":
                    batch.set_next_device(ptrs(k + 1))
                elif mode == "wrong":     # every third step names the images of the step after next: not what comes
                    batch.set_next_device(ptrs(min(k + 2, n_steps - 1)) if k % 3 == 0 else ptrs(k + 1))
            elif mode != "plain":
                batch.set_next_device(ptrs(0))   # announced, never used: dropped when the batch goes
            st = batch.step_device(ptrs(k))
            out += [(s.state, s.quality, s.keyframe, s.n_corners, s.matches, s.attempts, s.inliers, s.outliers, s.align_meas, tuple(s.pose[:])) for s in st]
        batch.close()
        dev.close()
        return out

    plain = run("plain")
    assert run("ahead") == plain
    assert run("wrong") == plain
    assert sum(r[2] for r in plain[n:]) >= n           # keyframes were created after the bootstrap
    assert all(r[1] == 0 and r[4] >= 100 for r in plain[n:])


def test_transient_images_with_min_align_level_0_on_a_one_slot_ring(trk, orc, synth):
    """SDVL.min_alignLevel 0 is legal (config.cc:146): the next step's image alignment then reads last_frame's LEVEL 0
    (image_align.cc:212).  With transient images that level aliases a ring slot; here the ring has ONE slot, rewritten before
    every step, so a frame that did not take its image out at the end of its own step (only keyframes did) is
    aligned against the NEXT image.  Results must equal those of frames that alias images resident for good, and the oracle's
    with the same configuration."""
    import ctypes as C
    import importlib
    sdvl = importlib.import_module("slam-sdvl_amd")
    import bench as B
    n, n_steps = 2, 9
    fb = 640 * 480
    over = dict(trk.TUM_OVERRIDES)
    over["SDVL.min_alignLevel"] = 0
    trk.configure(over)
    old = orc.params.min_align_level
    orc.params.min_align_level = 0
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    try:
        def run(mode):
            dev = trk.HostDevice(0)
            batch = trk.TrackerBatch(dev, n, 640, 480, TUM_CAM)
            ctx = B.CtxView(sdvl, dev.ctx_handle())
            buf = ctx.malloc(n * n_steps * fb)
            slot = ctx.malloc(n * fb)
            for k in range(n_steps):
                views = [B.make_view(sdvl, trajectory_pose(orc, k, XI * (1.0 + 0.1 * i)), 20260001 + i, k) for i in range(n)]
                ctx.render(views, buf + k * n * fb)
            stream = C.c_void_p(ctx.lib.sdvl_ctx_stream(ctx.h))
            out, imgs = [], []
            for k in range(n_steps):
                if mode == "resident":
                    st = batch.step_device([buf + (k * n + i) * fb for i in range(n)])
                else:
                    assert hip.hipMemcpyAsync(slot, buf + k * n * fb, n * fb, 3, stream) == 0
                    st = batch.step_device_transient([slot + i * fb for i in range(n)])
                out += [(s.state, s.quality, s.keyframe, s.n_corners, s.matches, s.attempts, s.inliers, s.outliers, s.align_meas, tuple(s.pose[:])) for s in st]
            if mode == "resident":
                imgs = [ctx.download(buf + (k * n) * fb, fb).reshape(480, 640).copy() for k in range(n_steps)]
            batch.close()
            dev.close()
            return out, imgs

        resident, imgs = run("resident")
        ring, _ = run("ring")
        assert ring == resident
        assert all(r[1] == 0 and r[4] >= 100 for r in resident[n:])
        o = orc.tracker(640, 480, TUM_CAM)
        for k in range(n_steps):
            want, g = o.handle_frame(imgs[k]), resident[k * n]
            assert g[:9] == (want.state, want.quality, want.keyframe, want.n_corners, want.matches, want.attempts, want.inliers, want.outliers,
                             want.align_meas), k
            assert np.abs(np.array(g[9]) - np.array(want.pose[:])).max() <= POSE_TOL, k
        o.close()
    finally:
        orc.params.min_align_level = old
        trk.configure()


def test_farm_host_input_ring_gives_identical_results(trk, orc, synth):
    """host-fed farm: frames in pinned host memory.  With the input ring (the images of step s + 1 travel on the group's copy
    stream while step s computes; contiguous runs as one DMA, the rest per image) and without it (every step uploads its own
    frames first) the per-frame results are those of the HBM-resident run"""
    import importlib
    import torch
    sdvl = importlib.import_module("slam-sdvl_amd")
    import bench as B
    G, Bg, n_steps = 3, 4, 6
    n = G * Bg
    fb = 640 * 480
    trk.configure()
    pinned = torch.empty(n * n_steps * fb + 4096, dtype=torch.uint8, pin_memory=True)

    def run(mode):
        farm = trk.TrackerFarm(0, G, Bg, 640, 480, TUM_CAM)
        ctx = B.CtxView(sdvl, farm.ctx_handle(0))
        buf = ctx.malloc(n * n_steps * fb)
        for k in range(n_steps):
            views = [B.make_view(sdvl, trajectory_pose(orc, k, XI * (1.0 + 0.1 * i)), 20260001 + i, k) for i in range(n)]
            ctx.render(views, buf + k * n * fb)
        idx = (np.arange(n_steps, dtype=np.uint64)[:, None] * n + np.arange(n, dtype=np.uint64)[None, :])
        if mode == "resident":
            ptrs = (buf + idx * fb).astype(np.uint64)
        else:
            pinned.numpy()[:n * n_steps * fb] = ctx.download(buf, n * n_steps * fb)
            ptrs = (pinned.data_ptr() + idx * fb).astype(np.uint64)
            farm.set_host_input(True)
            farm.set_input_ring(mode != "no-ring")
        st = farm.run(ptrs, G)
        out = [(s.state, s.quality, s.keyframe, s.n_corners, s.matches, s.attempts, s.inliers, s.outliers, s.align_meas, tuple(s.pose[:])) for s in st]
        farm.close()
        return out

    resident = run("resident")
    assert run("ring") == resident
    assert run("no-ring") == resident
    assert all(r[1] == 0 and r[4] >= 100 for r in resident[n:])


def test_farm_at_bench_size_replicas_agree_and_match_the_oracle(trk, orc, synth):
    """The bench configuration (4096 sequences, 16 groups of 256, one worker per group: BENCH_rNN.json `config.groups` /
    `sequences_per_group` — launch geometry, XCD dealing and block tables depend on the group size) on 8 distinct sequences
    replicated 512 times: where a sequence sits in a group, which group and which stream it runs on must not matter — every
    replica gives the same per-frame record, bit for bit — and the 8 distinct ones equal the CPU oracle (decisions exact, poses
    within 1e-4)."""
    import importlib
    sdvl = importlib.import_module("slam-sdvl_amd")
    import bench as B
    G, Bg, n_steps, distinct = 16, 256, 6, 8
    n = G * Bg
    trk.configure()
    farm = trk.TrackerFarm(0, G, Bg, 640, 480, TUM_CAM)
    ctx = B.CtxView(sdvl, farm.ctx_handle(0))
    fb = 640 * 480
    xis = [XI * (1.0 + 0.2 * i) * (1 if i % 2 == 0 else -1) for i in range(distinct)]
    # a replica's position: sequence i*7 mod 8 — neighbours in a group differ, the same sequence lands in every group
    which = [(i * 7 + i // Bg) % distinct for i in range(n)]
    buf = ctx.malloc(n * n_steps * fb)
    for k in range(n_steps):
        views = [B.make_view(sdvl, trajectory_pose(orc, k, xis[which[i]]), 20260101 + which[i], k) for i in range(n)]
        ctx.render(views, buf + k * n * fb)
    ptrs = (buf + (np.arange(n_steps, dtype=np.uint64)[:, None] * n + np.arange(n, dtype=np.uint64)[None, :]) * fb).astype(np.uint64)
    farm.reserve(Bg * 6)
    st = farm.run(ptrs, G)
    rec = [(s.state, s.quality, s.keyframe, s.n_corners, s.matches, s.attempts, s.inliers, s.outliers, s.align_meas, tuple(s.pose[:])) for s in st]
    first = {}
    for k in range(n_steps):
        for i in range(n):
            key = (k, which[i])
            if key not in first:
                first[key] = rec[k * n + i]
            assert rec[k * n + i] == first[key], (k, i)
    for d in range(distinct):
        o = orc.tracker(640, 480, TUM_CAM)
        i0 = which.index(d)
        for k in range(n_steps):
            img = ctx.download(buf + (k * n + i0) * fb, fb).reshape(480, 640)
            want = o.handle_frame(img)
            g = first[(k, d)]
            assert g[:9] == (want.state, want.quality, want.keyframe, want.n_corners, want.matches, want.attempts, want.inliers, want.outliers,
                             want.align_meas), (k, d)
            assert np.abs(np.array(g[9]) - np.array(want.pose[:])).max() <= POSE_TOL
            if k > 0:
                assert g[1] == 0 and g[4] >= 100
        o.close()
    # a fast path must be SEEN to run: every tracked job of every group searched through the corner bins (for a round and a half the
    # tracked step scanned whole corner lists — same answers, 12 % of the throughput; sdvl_ctx_counters is the witness)
    lib = sdvl.load_library()
    lib.sdvl_ctx_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    binned = scanned = 0
    for g in range(G):
        out = (C.c_int64 * 4)()
        assert lib.sdvl_ctx_counters(C.c_void_p(farm.ctx_handle(g)), out) == 0
        binned += out[0]
        scanned += out[1]
    assert scanned == 0 and binned == n * (n_steps - 1), (binned, scanned)
    farm.close()


def test_closed_loop_without_orb_uses_zmssd_matching(trk, orc, synth):
    """SDVL.use_orb: 0 (the reference's default, config.cc:85): detection margin 1 + PatchSize/2, no descriptors, candidates
    ranked by the integer ZMSSD of the warped 8x8 patch (matcher.cc:447-476) — the whole loop must still agree"""
    over = dict(trk.TUM_OVERRIDES)
    over["SDVL.use_orb"] = 0
    trk.configure(over)
    old = orc.params.use_orb
    orc.params.use_orb = 0
    try:
        dev = trk.HostDevice(0)
        batch = trk.TrackerBatch(dev, 2, 640, 480, TUM_CAM)
        refs = [orc.tracker(640, 480, TUM_CAM) for _ in range(2)]
        for k in range(9):
            imgs = [synth.render(trajectory_pose(orc, k, XI * (1 + 0.3 * i)), TUM_CAM, 640, 480, seed=20260001 + i, frame_id=k) for i in range(2)]
            got = batch.step_host(imgs)
            for i in range(2):
                w, g = refs[i].handle_frame(imgs[i]), got[i]
                assert (g.state, g.quality, g.keyframe, g.n_corners, g.matches, g.attempts, g.inliers, g.outliers, g.align_meas) == \
                       (w.state, w.quality, w.keyframe, w.n_corners, w.matches, w.attempts, w.inliers, w.outliers, w.align_meas), (k, i)
                assert np.abs(np.array(g.pose[:]) - np.array(w.pose[:])).max() <= POSE_TOL, (k, i)
                if k > 0:
                    assert g.quality == 0 and g.matches >= 80
        batch.close()
        for r in refs:
            r.close()
        dev.close()
    finally:
        orc.params.use_orb = old
        trk.configure()


def test_closed_loop_config_c_large_frames_many_features(trk, orc, synth):
    """BASELINE config C: 1280x960, num_features 4000, max_matches 1000.  Exercises the paths the 640x480 loop does not:
    the global-memory image-alignment kernel (> 384 features per job) and pose jobs with up to 1000 matches."""
    cam = np.array([1034.6, 1033.0, 637.2, 510.6])
    over = dict(trk.TUM_OVERRIDES)
    over.update({"SDVL.num_features": 4000, "SDVL.max_matches": 1000})
    trk.configure(over)
    saved = (orc.params.num_features, orc.params.max_matches)
    orc.params.num_features, orc.params.max_matches = 4000, 1000
    try:
        dev = trk.HostDevice(0)
        batch = trk.TrackerBatch(dev, 1, 1280, 960, cam)
        ref = orc.tracker(1280, 960, cam)
        for k in range(5):
            img = synth.render(trajectory_pose(orc, k), cam, 1280, 960, seed=20260100, frame_id=k)
            g, w = batch.step_host([img])[0], ref.handle_frame(img)
            assert (g.state, g.quality, g.keyframe, g.n_corners, g.matches, g.attempts, g.inliers, g.outliers, g.align_meas) == \
                   (w.state, w.quality, w.keyframe, w.n_corners, w.matches, w.attempts, w.inliers, w.outliers, w.align_meas), k
            assert np.abs(np.array(g.pose[:]) - np.array(w.pose[:])).max() <= POSE_TOL, k
            if k > 0:
                assert g.matches > 256 and g.align_meas > 384      # really beyond the limits of the LDS paths
        batch.close(); ref.close(); dev.close()
    finally:
        orc.params.num_features, orc.params.max_matches = saved
        trk.configure()


def test_mapper_mode_lost_and_relocalize(trk, orc, synth):
    """mapper mode through a tracking loss: while relocalisation is pending the mapper stands still (Map::SetRelocalizing,
    sdvl.cc:80-84, map.cc:79-80); afterwards it resumes — decisions and map bookkeeping stay identical to the oracle"""
    trk.configure()
    trk.set_mapper(True)
    try:
        dev = trk.HostDevice(0)
        batch = trk.TrackerBatch(dev, 1, 640, 480, TUM_CAM)
    finally:
        trk.set_mapper(False)
    ref = orc.tracker(640, 480, TUM_CAM)
    ref.use_mapper(True)
    seq = list(range(12)) + [-1, -1, -1, -1] + [11, 12, 13, 14, 15]
    relocs = 0
    for k, idx in enumerate(seq):
        img = np.full((480, 640), 127, np.uint8) if idx < 0 else synth.render(trajectory_pose(orc, idx), TUM_CAM, 640, 480, frame_id=idx)
        g = batch.step_host([img])[0]
        w = ref.handle_frame(img)
        assert g.host_path == 0, (k, idx)      # relocalisation included: the step never leaves the device-resident tables
        assert (g.quality, g.matches, g.attempts, g.inliers, g.keyframe, g.relocalized) == \
               (w.quality, w.matches, w.attempts, w.inliers, w.keyframe, w.relocalized), (k, idx)
        assert batch.map_stats(0) == ref.map_stats(), (k, idx)
        if idx >= 0:
            assert np.abs(np.array(g.pose[:]) - np.array(w.pose[:])).max() <= POSE_TOL, (k, idx)
        relocs += g.relocalized
    assert relocs == 1
    batch.close(); ref.close(); dev.close()


def test_cpp_example_track_sequence_matches_the_oracle(orc, synth, tmp_path):
    """slam-sdvl_amd/host/track_sequence: the reference's main.cc loop written against sdvl_host.h (no Python, no ctypes in
    the path) — on rendered frames and on the same frames read back from a PGM list it reports what the CPU oracle does"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slam-sdvl_amd", "host", "track_sequence")
    assert os.path.exists(exe), "build() makes it (make -C slam-sdvl_amd/host)"
    n = 6
    imgs = [synth.render(trajectory_pose(orc, k), TUM_CAM, 640, 480, frame_id=k) for k in range(n)]
    lst = tmp_path / "frames.txt"
    with open(lst, "w") as fh:
        for k, im in enumerate(imgs):
            p = tmp_path / ("f%03d.pgm" % k)
            with open(p, "wb") as out:
                out.write(b"P5\n# frame %d\n640 480\n255\n" % k)
                out.write(im.tobytes())
            fh.write(str(p) + "\n")
    ref = orc.tracker(640, 480, TUM_CAM)
    wants = [ref.handle_frame(im) for im in imgs]
    ref.close()
    # third run: the tracked step's search without the corner bins (SDVL_TRACK_NO_BINS=1: the views of the current frames lose the
    # bins sdvl_track_align named ahead of the detection, as for frames whose corners were set by hand) — same answers
    # further runs, one per environment switch of the tracking step (README): the host-driven form (round 1's path), the pose stage on
    # the host, the one-shot batch per HandleFrame call — same answers each time
    for args, env in ((["--synthetic", str(n)], {}), (["--list", str(lst)], {}), (["--synthetic", str(n)], {"SDVL_TRACK_NO_BINS": "1"}),
                      (["--synthetic", str(n)], {"SDVL_NO_TRACK_TABLES": "1"}), (["--synthetic", str(n)], {"SDVL_POSE_HOST": "1"}),
                      (["--synthetic", str(n)], {"SDVL_HANDLEFRAME_ONE_SHOT": "1"})):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        rows = [l.split() for l in r.stdout.strip().splitlines()]
        assert len(rows) == n
        for k, (row, w) in enumerate(zip(rows, wants)):
            assert [int(v) for v in row[:6]] == [k, w.state, w.quality, w.matches, w.attempts, w.inliers], (args[0], k)
            assert np.abs(np.array([float(v) for v in row[6:13]]) - np.array(w.pose[:])).max() <= POSE_TOL
        assert "tracked frames/s" in r.stderr


def test_cpp_single_call_api_surface():
    """slam-sdvl_amd/host/api_surface_check: the reference's per-object calls one at a time (Frame ctor and accessors,
    FastDetector::DetectPyramid, ORBDetector::GetDescriptor / Distance, Frame::FilterCorners, ImageAlign::ComputePose,
    Matcher::SearchPoint, FeatureAlign::Reproject / OptimizePose) agree with the batched path and recover the rendered motion"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slam-sdvl_amd", "host", "api_surface_check")
    assert os.path.exists(exe), "build() makes it (make -C slam-sdvl_amd/host)"
    # (SDVL_GPU: the GPU a SDVL(Camera*) built on a thread without a Device opens for itself)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, SDVL_GPU="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) >= 16 and all(l.startswith("ok") for l in lines[:-1]) and lines[-1] == "0 check(s) failed"


def test_cpp_front_end_alone_against_minimal_camera_and_point():
    """slam-sdvl_amd/host/frontend_link_check: frontend.cc linked WITHOUT standalone.cc / mapper.cc / capi.cc, against a second,
    independent implementation of the six members of the reference's Camera / Point it calls (host/minimal_deps.cc; INTEGRATION.md
    route A).  The program tracks one frame through the reference's per-object calls — Frame ctor, FilterCorners, ImageAlign::ComputePose,
    Matcher::SearchPoint, FeatureAlign::Reproject + OptimizePose — and recovers the rendered motion"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slam-sdvl_amd", "host", "frontend_link_check")
    assert os.path.exists(exe), "build() makes it (make -C slam-sdvl_amd/host)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) >= 7 and all(l.startswith("ok") for l in lines[:-1]) and lines[-1] == "0 check(s) failed", r.stdout


def test_cpp_threaded_mode_tracker_and_mapper_on_two_threads():
    """slam-sdvl_amd/host/threaded_mode_check: main.cc's default mode (handler->Start(): the mapper runs on its own thread,
    map.cc:49-71) — two host threads inside the path, each with its own sdvl_ctx / stream, frames shared read-only.  The
    reference is not deterministic there, so: every frame tracked, pose on the rendered trajectory, the mapper thread did
    its work, the map is comparable to sequential mode's (the program prints both and checks)"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slam-sdvl_amd", "host", "threaded_mode_check")
    assert os.path.exists(exe), "build() makes it (make -C slam-sdvl_amd/host)"
    r = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("sequential") and lines[1].startswith("threaded")
    assert len(lines) >= 8 and all(l.startswith("ok") for l in lines[2:])
