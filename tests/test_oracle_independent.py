"""Independent re-derivations (numpy / scipy / libc) that pin the CPU restatement where the reference has no
golden vectors (the reference has no tests; OpenCV/Eigen are absent).  These check the ORACLE."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg
import scipy.ndimage

from oraclelib import TUM_CAM, trajectory_pose, quat_to_R

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
          (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def rand_img(seed, h, w, smooth=0):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w)).astype(np.float64)
    if smooth:
        img = scipy.ndimage.gaussian_filter(img, smooth)
        img = (img - img.min()) / (img.max() - img.min()) * 255
    return img.astype(np.uint8)


# ---------------------------------------------------------------- pyrDown (Appendix A.1)
@pytest.mark.parametrize("shape", [(480, 640), (30, 40), (60, 94), (8, 8), (6, 10)])
def test_pyrdown_matches_separable_integer_gaussian(orc, shape):
    img = rand_img(1, *shape)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    full = scipy.ndimage.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror")   # mirror == REFLECT_101
    full = scipy.ndimage.correlate1d(full, k, axis=0, mode="mirror")
    want = ((full[::2, ::2] + 128) >> 8)[: shape[0] // 2, : shape[1] // 2].astype(np.uint8)
    got = orc.pyr_down(img)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_pyrdown_constant_image(orc):
    assert np.all(orc.pyr_down(np.full((16, 24), 77, np.uint8)) == 77)


# ---------------------------------------------------------------- FAST-9/16 (Appendix A.2)
def fast_bruteforce(img, t):
    """FAST by definition: corner iff >= 9 contiguous circle pixels all brighter than v+t or all darker than v-t;
    score = largest threshold for which it is still a corner; 3x3 strict NMS; row-major output."""
    h, w = img.shape
    im = img.astype(np.int32)
    ring = np.stack([im[3 + dy: h - 3 + dy, 3 + dx: w - 3 + dx] for dx, dy in CIRCLE])   # [16, h-6, w-6]
    v = im[3:h - 3, 3:w - 3]

    def is_corner(thr):
        br = ring > v + thr
        dk = ring < v - thr
        out = np.zeros(v.shape, bool)
        for m in (br, dk):
            mm = np.concatenate([m, m[:8]])
            run = np.ones_like(m)
            for s in range(9):
                run = run & mm[s:s + 16]
            out |= run.any(axis=0)
        return out

    corner = is_corner(t)
    score = np.zeros(v.shape, np.int32)
    ys, xs = np.nonzero(corner)
    d = v[None] - ring                          # [16, ...]
    dd = np.concatenate([d, d[:8]])
    amin = np.stack([dd[s:s + 9].min(axis=0) for s in range(16)]).max(axis=0)
    bmin = np.stack([(-dd[s:s + 9]).min(axis=0) for s in range(16)]).max(axis=0)
    sc = np.maximum(np.maximum(amin, bmin), t) - 1
    score[corner] = sc[corner]
    # cross-check the closed form against the literal definition on a sample
    for y, x in list(zip(ys, xs))[:40]:
        s = score[y, x]
        assert is_corner_at(im, x + 3, y + 3, s) and not is_corner_at(im, x + 3, y + 3, s + 1)
    full = np.zeros((h, w), np.int32)
    full[3:h - 3, 3:w - 3] = score
    out = []
    for y, x in zip(ys + 3, xs + 3):
        s = full[y, x]
        nb = full[y - 1:y + 2, x - 1:x + 2].copy()
        nb[1, 1] = -1
        if (s > nb).all():
            out.append((x, y, s))
    return np.array(out, np.int32).reshape(-1, 3)


def is_corner_at(im, x, y, thr):
    v = im[y, x]
    p = np.array([im[y + dy, x + dx] for dx, dy in CIRCLE])
    for m in (p > v + thr, p < v - thr):
        mm = np.concatenate([m, m[:8]])
        if any(mm[s:s + 9].all() for s in range(16)):
            return True
    return False


@pytest.mark.parametrize("seed,shape,smooth,thr", [(3, (40, 52), 0, 10), (4, (64, 64), 1.0, 10), (5, (33, 47), 1.5, 20),
                                                   (6, (32, 32), 0, 40), (7, (90, 70), 2.0, 5)])
def test_fast_matches_definition(orc, seed, shape, smooth, thr):
    img = rand_img(seed, *shape, smooth=smooth)
    want = fast_bruteforce(img, thr)
    got = orc.fast(img, thr=thr)
    assert len(want) > 0
    assert np.array_equal(got, want)


def test_fast_constant_and_tiny(orc):
    assert len(orc.fast(np.full((32, 32), 100, np.uint8))) == 0
    assert len(orc.fast(rand_img(1, 6, 6))) == 0          # ROI smaller than 7x7: loops do not execute
    # a single bright dot on a dark ground is not a FAST-9 corner for its neighbours, but a dark dot centre is:
    img = np.full((15, 15), 200, np.uint8); img[7, 7] = 10
    got = orc.fast(img, thr=10)
    assert got.tolist() == [[7, 7, 189]]                  # all 16 ring pixels brighter by 190 -> score 190-1


def test_fast_cells_equal_per_roi_fast(orc, synth):
    """SelectPixels runs cv::FAST on each cell ROI independently (fast_detector.cc:80-106)."""
    img = synth.render(trajectory_pose(orc, 0), TUM_CAM, 640, 480)
    kps, offs, ran = orc.fast_cells(img)
    m, cs = 19, 32
    wc = 20
    for (i, j) in [(0, 0), (0, 19), (14, 0), (7, 9), (14, 19), (3, 3)]:
        y0, y1 = max(m, i * cs), min(480 - m, i * cs + cs)
        x0, x1 = max(m, j * cs), min(640 - m, j * cs + cs)
        roi = np.ascontiguousarray(img[y0:y1, x0:x1])
        want = fast_bruteforce(roi, 10) if min(roi.shape) >= 7 else np.zeros((0, 3), np.int32)
        if len(want):
            want = want + np.array([x0, y0, 0], np.int32)
        c = i * wc + j
        got = kps[offs[c]:offs[c + 1]]
        assert ran[c] == 1
        assert np.array_equal(got, want)


def test_detect_pyramid_quota_and_margins(orc, synth):
    img = synth.render(trajectory_pose(orc, 0), TUM_CAM, 640, 480)
    c = orc.detect_pyramid(img)
    # quotas 395/329/274 (fast_detector.cc:161-174); retainBest keeps boundary ties so counts may exceed
    per = [int((c[:, 2] == l).sum()) for l in range(3)]
    assert per[0] >= 395 and per[1] >= 329 and per[2] >= 274 and sum(per) < 1100
    for l in range(3):
        w, h = 640 >> l, 480 >> l
        cl = c[c[:, 2] == l]
        assert cl[:, 0].min() >= 22 and cl[:, 0].max() < w - 22      # margin 19 + FAST border 3
        assert cl[:, 1].min() >= 22 and cl[:, 1].max() < h - 22


# ---------------------------------------------------------------- Shi-Tomasi (extra/utils.cc:61-97)
def test_shi_tomasi_matches_eigenvalue(orc):
    img = rand_img(11, 64, 64, smooth=1.2)
    im = img.astype(np.float64)
    for (x, y) in [(20, 20), (31, 40), (10, 50), (50, 12)]:
        dx = im[y - 4:y + 4, x - 4 + 1:x + 4 + 1] - im[y - 4:y + 4, x - 4 - 1:x + 4 - 1]
        dy = im[y - 4 + 1:y + 4 + 1, x - 4:x + 4] - im[y - 4 - 1:y + 4 - 1, x - 4:x + 4]
        M = np.array([[np.sum(dx * dx), np.sum(dx * dy)], [np.sum(dx * dy), np.sum(dy * dy)]]) / 128.0
        want = np.linalg.eigvalsh(M)[0]
        got = orc.shi_tomasi(img, x, y)
        assert abs(got - want) <= 1e-3 * max(1.0, abs(want))
    assert orc.shi_tomasi(img, 3, 30) == 0.0           # box touches the border


# ---------------------------------------------------------------- ORB (extra/orb_detector.cc)
def test_orb_orientation_and_rotation_consistency(orc):
    img = rand_img(21, 96, 96, smooth=2.0)
    desc, ang = orc.orb_describe(img, [[48, 48]])
    # intensity-centroid by definition over the radius-15 disc used by ORB (umax table)
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    m10 = m01 = 0
    for v in range(-15, 16):
        for u in range(-umax[abs(v)], umax[abs(v)] + 1):
            p = int(img[48 + v, 48 + u]); m10 += u * p; m01 += v * p
    want = np.degrees(np.arctan2(m01, m10)) % 360
    assert abs(((ang[0] - want + 180) % 360) - 180) < 0.02        # fastAtan2 is a <=0.01 deg polynomial
    # descriptor of the 180-degree-rotated image at the mirrored point is identical (steering by the angle)
    rot = np.ascontiguousarray(img[::-1, ::-1])
    d2, a2 = orc.orb_describe(rot, [[95 - 48, 95 - 48]])
    assert abs(((a2[0] - ang[0] - 180 + 180) % 360) - 180) < 0.02
    assert orc.orb_distance(desc[0], d2[0]) <= 6                   # only rounding-tie samples may differ


def test_orb_distance_is_hamming(orc):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, 32).astype(np.uint8); b = rng.integers(0, 256, 32).astype(np.uint8)
    assert orc.orb_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    assert orc.orb_distance(a, a) == 0


# ---------------------------------------------------------------- SE3 / LDLT / rand
def hat6(u):
    w = u[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u[:3]
    return M


def T_to_mat(T):
    M = np.eye(4); M[:3, :3] = quat_to_R(T[:4]); M[:3, 3] = T[4:]
    return M


def test_se3_exp_log_compose_against_expm(orc):
    rng = np.random.default_rng(5)
    for _ in range(20):
        u = rng.normal(size=6) * [0.3, 0.3, 0.3, 0.4, 0.4, 0.4]
        v = rng.normal(size=6) * 0.2
        A, B = orc.se3_exp(u), orc.se3_exp(v)
        assert np.allclose(T_to_mat(A), scipy.linalg.expm(hat6(u)), atol=1e-12)
        assert np.allclose(orc.se3_log(A), u, atol=1e-10)
        assert np.allclose(T_to_mat(orc.se3_mul(A, B)), T_to_mat(A) @ T_to_mat(B), atol=1e-12)
        assert np.allclose(T_to_mat(orc.se3_inv(A)), np.linalg.inv(T_to_mat(A)), atol=1e-12)
    assert np.allclose(orc.se3_exp(np.zeros(6)), [1, 0, 0, 0, 0, 0, 0])


def test_ldlt_solve_against_numpy(orc):
    rng = np.random.default_rng(9)
    for _ in range(20):
        J = rng.normal(size=(40, 6)) * rng.uniform(0.1, 100, size=6)
        A = J.T @ J; b = rng.normal(size=6)
        x = orc.ldlt_solve6(A, b)
        assert np.allclose(A @ x, b, rtol=1e-8, atol=1e-8 * np.abs(b).max())
    assert np.all(orc.ldlt_solve6(np.zeros((6, 6)), np.ones(6)) == 0.0)   # all-zero H -> x = 0, not NaN (App. C)


def test_rand_stream_is_glibc_rand(orc):
    libc = C.CDLL("libc.so.6")
    libc.srand(1)
    want = [libc.rand() for _ in range(2000)]
    assert orc.rand_stream(2000, seed=1).tolist() == want


def test_pose_from_matches_recovers_motion_and_rejects_outliers(orc):
    """SelectInliers + OptimizePose (feature_align.cc:73-82,152-243): independent geometric check — noiseless matches of a
    known camera motion with gross outliers injected: the pose is recovered, exactly the injected set is rejected, and
    the adaptive RANSAC budget stops early (far fewer than max_ransac_its draws)."""
    from oraclelib import TUM_CAM, quat_rot
    rng = np.random.default_rng(11)
    true_pose = orc.se3_exp(np.array([0.05, -0.03, 0.02, 0.01, -0.008, 0.005]))
    n = 120
    P = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-0.9, 0.9, n), rng.uniform(1.5, 3.0, n)], 1)
    pc = P @ quat_rot(true_pose[:4]).T + true_pose[4:]
    a = pc[:, :2] / pc[:, 2:3]
    bad = np.zeros(n, bool)
    bad[rng.choice(n, 15, replace=False)] = True    # few enough that most 5-runs of consecutive matches are clean
    a[bad] += np.sign(rng.normal(size=(int(bad.sum()), 2))) * rng.uniform(20, 60, (int(bad.sum()), 2)) / TUM_CAM[0]
    obs = np.concatenate([a, P, np.zeros((n, 1))], 1)
    r = orc.pose_from_matches(TUM_CAM, obs, orc.se3_exp(np.zeros(6)))
    assert np.abs(r["pose"] - true_pose).max() < 1e-8
    assert sorted(r["outliers"].tolist()) == np.flatnonzero(bad).tolist()
    assert sorted(r["inliers"].tolist()) == np.flatnonzero(~bad).tolist()
    assert 1 <= r["n_draws"] < 100
    empty = orc.pose_from_matches(TUM_CAM, np.zeros((0, 6)), true_pose)
    assert empty["n_draws"] == 0 and len(empty["inliers"]) == 0 and np.array_equal(empty["pose"], true_pose)


def test_undistort_against_float_remap(orc):
    """cv::undistort restatement (camera.cc:100-105) vs an independent float pipeline: the radial-tangential model in
    numpy + scipy's bilinear map_coordinates.  The fixed-point remap (1/32 px coordinates, 15-bit weights) may differ
    from the float result by the coordinate quantisation only: a few grey levels at most on a smooth image, and the
    weight table must be the exact bilinear products."""
    import scipy.ndimage as ndi
    from oraclelib import TUM_CAM, TUM_DIST, EUROC_CAM, EUROC_DIST
    wt = orc.remap_weights().astype(np.int64)
    a = np.arange(32)
    for iy in (0, 1, 7, 31):
        for ix in (0, 5, 16, 31):
            want = np.array([(32 - iy) * (32 - ix), (32 - iy) * ix, iy * (32 - ix), iy * ix]) * 32
            if iy == 0 and ix == 0:
                want = np.array([32767, 0, 0, 1])     # saturate_cast<short>(32768) and the table's sum repair
            assert wt[iy, ix].tolist() == want.tolist(), (iy, ix)
    assert (wt.sum(axis=2) == 32768).all()
    yy, xx = np.mgrid[0:480, 0:640].astype(np.float64)
    smooth = (127 + 80 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + 30 * np.sin((xx + yy) / 41.0)).astype(np.uint8)
    for (w, h, cam, dist) in ((640, 480, TUM_CAM, TUM_DIST), (752, 480, EUROC_CAM, EUROC_DIST)):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        img = (127 + 80 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + 30 * np.sin((xx + yy) / 41.0)).astype(np.uint8)
        got = orc.undistort(img, cam, dist).astype(np.int32)
        x = (xx - cam[2]) / cam[0]; y = (yy - cam[3]) / cam[1]
        r2 = x * x + y * y
        kr = 1 + ((dist[4] * r2 + dist[1]) * r2 + dist[0]) * r2
        u = cam[0] * (x * kr + dist[2] * 2 * x * y + dist[3] * (r2 + 2 * x * x)) + cam[2]
        v = cam[1] * (y * kr + dist[2] * (r2 + 2 * y * y) + dist[3] * 2 * x * y) + cam[3]
        ref = ndi.map_coordinates(img.astype(np.float64), [v, u], order=1, mode="constant", cval=0.0)
        inside = (u >= 1) & (u < w - 2) & (v >= 1) & (v < h - 2)
        diff = np.abs(got - ref)[inside]
        assert diff.max() <= 3.0 and diff.mean() < 0.6, (diff.max(), diff.mean())
        outside = (u < -1) | (u > w) | (v < -1) | (v > h)
        assert (got[outside] == 0).all()                       # BORDER_CONSTANT
    # d0 == 0 -> the reference does not remap at all (camera.cc:46)
    assert np.array_equal(orc.undistort(smooth, TUM_CAM, [0, 0.3, 0, 0, 0]), smooth)


def test_mapper_helpers_against_numpy(orc):
    """GetDepthFromTriangulation (extra/utils.cc:193-205) vs a least-squares solve, PDFNormal (point.cc:203-217) vs the
    closed form, ComputeTau (point.cc:189-201) vs a direct perturbation of the observation angle"""
    import ctypes as C
    from oraclelib import quat_rot
    lib = orc.lib
    lib.sdvl_ref_pdf_normal.restype = C.c_double
    lib.sdvl_ref_pdf_normal.argtypes = [C.c_double] * 3
    lib.sdvl_ref_compute_tau.restype = C.c_double
    lib.sdvl_ref_compute_tau.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    lib.sdvl_ref_triangulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    for _ in range(30):
        T = orc.se3_exp(rng.normal(size=6) * [0.3, 0.3, 0.1, 0.05, 0.05, 0.05])     # cur <- ref
        P_ref = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(1.5, 4.0)])
        v_ref = P_ref / np.linalg.norm(P_ref)
        P_cur = quat_rot(T[:4]) @ P_ref + T[4:]
        v_cur = P_cur / np.linalg.norm(P_cur)
        d = C.c_double()
        assert lib.sdvl_ref_triangulate(T.ctypes.data, v_ref.ctypes.data, v_cur.ctypes.data, C.byref(d)) == 1
        A = np.stack([quat_rot(T[:4]) @ v_ref, v_cur], 1)
        want = np.abs(np.linalg.lstsq(A, -T[4:], rcond=None)[0][0])
        assert abs(d.value - want) < 1e-9 * max(1.0, want)
        assert abs(d.value - np.linalg.norm(P_ref)) < 1e-9          # it IS the distance along the reference bearing
    for mean, sd, x in [(0.5, 0.1, 0.45), (1.0, 2.0, -3.0), (0.2, 1e-3, 0.2)]:
        want = np.exp(-(x - mean) ** 2 / (2 * sd * sd)) / (sd * np.sqrt(2.0 * 3.14159265))
        assert abs(lib.sdvl_ref_pdf_normal(mean, sd, x) - want) <= 1e-12 * want
    assert lib.sdvl_ref_pdf_normal(0.0, 0.0, 0.0) == 0.0
    # tau: depth change when the observation ray in the second view turns by one pixel-error angle
    T = orc.se3_exp(np.array([0.3, 0.0, 0.0, 0, 0, 0]))
    v = np.array([0.0, 0.0, 1.0]); depth = 2.0; ang = 2 * np.arctan(1.0 / (2 * 517.3))
    tau = lib.sdvl_ref_compute_tau(T.ctypes.data, v.ctypes.data, depth, ang)
    t = T[4:]; a = v * depth - t
    alpha = np.arccos(v @ t / np.linalg.norm(t)); beta = np.arccos(a @ -t / (np.linalg.norm(t) * np.linalg.norm(a)))
    want = np.linalg.norm(t) * np.sin(beta + ang) / np.sin(3.14159265 - alpha - beta - ang) - depth
    assert abs(tau - want) < 1e-12 and 0.0 < tau < 0.1


def test_reference_mapper_reconstructs_the_scene_plane(orc, synth):
    """independent end-to-end check of the mapper restatement (map.cc + point.cc depth filter): in the synthetic scene
    every surface point lies on the plane z = 2.  Points the mapper triangulates and filters on its own — no knowledge
    of the plane past the bootstrap keyframe — must land on it, and the trajectory must stay on the ground truth."""
    from oraclelib import TUM_CAM, trajectory_pose
    t = orc.tracker(640, 480, TUM_CAM)
    t.use_mapper(True)
    worst = 0.0
    for k in range(36):
        st = t.handle_frame(synth.render(trajectory_pose(orc, k), TUM_CAM, 640, 480, seed=20260001, frame_id=k))
        worst = max(worst, np.abs(np.array(st.pose[:]) - trajectory_pose(orc, k)).max())
        if k > 0:
            assert st.quality == 0 and st.matches >= 100
    pts = t.mapper_points()
    ms = t.map_stats()
    t.close()
    assert worst < 2e-3
    conv = pts[pts[:, 3] == 1.0]
    assert len(conv) >= 60 and ms["initialized"] >= 150 and ms["keyframes"] >= 4
    err = np.abs(conv[:, 2] - 2.0)
    assert np.median(err) < 0.01 and np.percentile(err, 90) < 0.04, (np.median(err), np.percentile(err, 90))


def test_image_align_recovers_the_rendered_camera_motion(orc, synth):
    """independent check of the ImageAlign restatement (image_align.cc:46-267): frames of the textured plane z = 2 rendered
    from known poses, features with their true depth -> the inverse-compositional Gauss-Newton must land on the pose the
    renderer used, starting from identity"""
    from oraclelib import TUM_CAM, trajectory_pose
    img0 = synth.render(trajectory_pose(orc, 0), TUM_CAM, 640, 480, seed=20260001, frame_id=0)
    rng = np.random.default_rng(20260200)
    n = 200
    px = np.stack([rng.uniform(48, 640 - 48, n), rng.uniform(48, 480 - 48, n)], 1)
    ray = np.stack([(px[:, 0] - TUM_CAM[2]) / TUM_CAM[0], (px[:, 1] - TUM_CAM[3]) / TUM_CAM[1], np.ones(n)], 1)
    bearing = ray / np.linalg.norm(ray, axis=1, keepdims=True)
    depth = 2.0 / bearing[:, 2]
    for k in (1, 3, 6):
        imgk = synth.render(trajectory_pose(orc, k), TUM_CAM, 640, 480, seed=20260001, frame_id=k)
        r = orc.image_align(img0, imgk, TUM_CAM, px, bearing, depth, np.ones(n, np.uint8), [1, 0, 0, 0, 0, 0, 0])
        assert r["n"] > 150
        assert np.abs(r["T"] - trajectory_pose(orc, k)).max() < 2e-3, (k, r["T"], trajectory_pose(orc, k))


def test_align_patch_recovers_a_known_shift(orc, synth):
    """Matcher::AlignPatch restatement (matcher.cc:359-445): template cut at an integer position of an image, search image
    = the same image shifted by a known sub-pixel offset (bilinear resampling) -> the LK iterations converge onto it"""
    import scipy.ndimage as ndi
    from oraclelib import TUM_CAM, trajectory_pose
    base = synth.render(trajectory_pose(orc, 0), TUM_CAM, 640, 480, seed=20260001, frame_id=0).astype(np.float64)
    smooth = ndi.gaussian_filter(base, 1.5)
    img_ref = np.clip(np.rint(smooth), 0, 255).astype(np.uint8)
    rng = np.random.default_rng(4)
    n_ok = 0
    for _ in range(25):
        sx, sy = rng.uniform(-1.2, 1.2, 2)
        moved = ndi.shift(smooth, (sy, sx), order=1, mode="nearest")   # moved(y, x) = smooth(y - sy, x - sx)
        img_cur = np.clip(np.rint(moved), 0, 255).astype(np.uint8)
        cx, cy = int(rng.integers(60, 580)), int(rng.integers(60, 420))
        border = img_ref[cy - 5:cy + 5, cx - 5:cx + 5].copy()           # 10x10 around (cx, cy): rows cy-5 .. cy+4
        patch = border[1:9, 1:9].copy()
        ok, px = orc.align_patch(img_cur, border, patch, [cx, cy])
        if ok:
            n_ok += 1
            # the patch centre convention of the 8x8 template (pixels cx-4 .. cx+3) puts its content at (cx + sx, cy + sy)
            assert abs(px[0] - (cx + sx)) < 0.2 and abs(px[1] - (cy + sy)) < 0.2, (sx, sy, px, cx, cy)
    assert n_ok >= 20
