"""GPU parity tests proper: every kernel behind the C-ABI (include/sdvl_hip.h) against the CPU oracle on the same
seeded inputs.  Bit-exact for integer / byte / index work (pyramid, FAST lists incl. order, Shi-Tomasi, ORB bits,
SearchPoint decisions, AlignPatch offsets); 1e-4 (BASELINE.json north_star) for the image-alignment pose, whose
normal equations are reduced in a different order than the CPU's sequential loop."""
import ctypes as C
import importlib

import numpy as np
import pytest

from oraclelib import EUROC_CAM, TUM_CAM, trajectory_pose

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-4   # BASELINE.json: "pose and patch-offset deltas within 1e-4"


@pytest.fixture(scope="module")
def sdvl():
    return importlib.import_module("slam-sdvl_amd")


@pytest.fixture(scope="module")
def ctx(sdvl):
    c = sdvl.Context(0)
    yield c
    c.close()


def rand_img(seed, h, w):
    return np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.uint8)


def frames_of(synth, orc, cam, w, h, ks, seed=20260001):
    return [synth.render(trajectory_pose(orc, k), cam, w, h, seed=seed, frame_id=k) for k in ks]


# ------------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("shape", [(480, 640), (480, 752), (960, 1280), (66, 130), (48, 80)])
def test_pyramid_bit_exact(ctx, orc, shape):
    h, w = shape
    levels = 5 if min(shape) >= 64 * 4 else 3
    imgs = [rand_img(s, h, w) for s in (1, 2)]
    fr = [ctx.frame(im, levels=levels, pyramid=False) for im in imgs]
    ctx.pyramid_build(fr)
    for f, im in zip(fr, imgs):
        want = orc.pyramid(im, levels)
        for l in range(levels):
            assert np.array_equal(f.level(l), want[l]), "level %d" % l
        f.close()


# ------------------------------------------------------------------------------------------------ K2
def check_fast(ctx, sdvl, orc, imgs, margin=19, thr=10, cell=32, levels=5):
    dp = sdvl.default_detect_params()
    dp.margin, dp.fast_threshold, dp.cell_size = margin, thr, cell
    orc.params.fast_threshold = thr
    orc.params.cell_size = cell
    orc.params.use_orb = 1 if margin == 19 else 0
    try:
        fr = [ctx.frame(im, levels=levels) for im in imgs]
        got, cpl = ctx.fast_cells(fr, dp, cap=60000)
        total = 0
        for f, im, (kps, offs) in zip(fr, imgs, got):
            pyr = orc.pyramid(im, levels)
            base = 0
            for l in range(3):
                wk, woffs, _ = orc.fast_cells(pyr[l], cap=400000)
                g = kps[kps[:, 3] == l]
                assert np.array_equal(g[:, :3], wk), "level %d keypoints (x,y,score) incl. order" % l
                assert np.array_equal(offs[base:base + cpl[l] + 1] - offs[base], woffs), "level %d cell offsets" % l
                base += cpl[l]
                total += len(wk)
            f.close()
        return total
    finally:
        orc.params.fast_threshold = 10
        orc.params.cell_size = 32
        orc.params.use_orb = 1


def sparse_corner_image(seed, h, w):
    """a smooth ramp with a few hundred small bright / dark squares and discs: corners on a few % of the pixels at most, the
    regime of a real camera frame (the synthetic plane and white noise put a corner on every second pixel)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = (60 + 0.1 * xx + 0.08 * yy).astype(np.float64)
    for _ in range(500):
        x, y, r = int(rng.integers(4, w - 12)), int(rng.integers(4, h - 12)), int(rng.integers(2, 7))
        val = float(rng.choice([-45, -30, 30, 45, 70]))
        if rng.random() < 0.5:
            img[y:y + r, x:x + r] += val
        else:
            img[(yy - y) ** 2 + (xx - x) ** 2 <= r * r] += val
    return np.clip(img, 0, 255).astype(np.uint8)


def test_fast_cells_sparse_corners_and_other_cell_sizes(ctx, sdvl, orc):
    """the two paths of fast_cells_kernel — candidate list (few corners) and pixel pairs on packed halves (dense texture) —
    on images of both kinds and on cell sizes other than 32 (odd tested widths, pairs whose second pixel is not tested)"""
    sparse = sparse_corner_image(3, 480, 640)
    dense = rand_img(8, 480, 640)
    n_sparse = check_fast(ctx, sdvl, orc, [sparse])
    assert 200 < n_sparse < 30000, n_sparse                      # a few corners per cell, not hundreds
    mixed = sparse.copy()
    mixed[:, 320:] = dense[:, 320:]                                # cells of both kinds in one frame, and cells that straddle the seam
    assert check_fast(ctx, sdvl, orc, [mixed, sparse], margin=5, thr=20) > 1000
    for cell in (30, 31):      # what the per-frame cell-list capacity admits at 640x480 besides 32 (smaller cells: SDVL_ERR_CAPACITY)
        assert check_fast(ctx, sdvl, orc, [mixed], cell=cell) > 1000, cell
        assert check_fast(ctx, sdvl, orc, [dense], margin=5, thr=30, cell=cell) > 1000, cell


@pytest.mark.parametrize("shape", [(112, 144), (120, 172), (190, 250), (144, 176)])
def test_fast_cells_small_and_unaligned_levels(ctx, sdvl, orc, shape):
    """levels whose rows are not word-aligned (250 px; 172 -> 86 -> 43), narrower than the 36-byte span of a lane's tile load (144 -> 72 -> 36)
    or both: the tile comes in byte by byte there instead of as one 16-byte load per lane — same corners"""
    h, w = shape
    dense = rand_img(21 + w, h, w)
    sparse = sparse_corner_image(5, 480, 640)[:h, :w].copy()
    assert check_fast(ctx, sdvl, orc, [dense, sparse], margin=5, thr=20, levels=3) > 200
    assert check_fast(ctx, sdvl, orc, [dense], thr=35, levels=3) > 50   # the ORB margin (19): ROIs that start three bytes off a word


def test_fast_cells_bit_exact_synthetic(ctx, sdvl, orc, synth):
    n = check_fast(ctx, sdvl, orc, frames_of(synth, orc, TUM_CAM, 640, 480, [0, 7]))
    assert n > 5000


def test_fast_cells_bit_exact_noise_and_thresholds(ctx, sdvl, orc):
    img = rand_img(5, 480, 640)
    assert check_fast(ctx, sdvl, orc, [img], thr=40) > 1000
    assert check_fast(ctx, sdvl, orc, [img], margin=5, thr=25) > 1000      # non-ORB margin (1 + PatchSize/2)
    flat = np.full((480, 640), 90, np.uint8)
    assert check_fast(ctx, sdvl, orc, [flat]) == 0


def test_fast_cells_euroc_size(ctx, sdvl, orc, synth):
    assert check_fast(ctx, sdvl, orc, frames_of(synth, orc, EUROC_CAM, 752, 480, [3])) > 3000


def test_detect_corners_with_many_score_ties_goes_through_the_spill_area(ctx, sdvl, orc):
    """isolated bright pixels on a 10-px lattice: 1680 FAST corners on level 0 and 1518 on level 1, all with the SAME score.
    retainBest keeps every tie at its boundary (fast_detector.cc:140,147-148), so a level's concatenated list (all of them) outgrows
    the LDS share select_pack plans for it (1.5 x (quota + cells) = 1088 / 640 entries) and is worked on in the frame's spill area
    in HBM: same corners, same order as the oracle."""
    img = np.full((480, 640), 40, np.uint8)
    img[24:456:10, 24:616:10] = 200
    want = orc.detect_pyramid(img)
    f = ctx.frame(img)
    got = ctx.detect_corners([f], sdvl.default_detect_params(), 1000)[0]
    assert len(want) == 3198 and np.bincount(want[:, 2]).tolist() == [1680, 1518]   # three times num_features: the ties were kept
    assert np.array_equal(got, want)
    f.close()


def test_corner_capacity_of_a_context(sdvl, orc, synth):
    """sdvl_ctx_set_corner_capacity: frames created afterwards hold that many corners (their resident block shrinks with it); a
    detection that fits behaves as ever, one that does not is reported as SDVL_ERR_CAPACITY by sdvl_frames_corner_counts"""
    lib = sdvl.load_library()
    lib.sdvl_frame_footprint_cap.restype = C.c_int64
    big, small = lib.sdvl_frame_footprint_cap(640, 480, 5, sdvl.MAX_CORNERS), lib.sdvl_frame_footprint_cap(640, 480, 5, 1536)
    assert 480_000 < small < 520_000 and 740_000 < big < 800_000        # 0.50 MB / 0.76 MB per 640x480 frame (1.43 MB in round 2)
    c2 = sdvl.Context(0)
    img, = frames_of(synth, orc, TUM_CAM, 640, 480, [2])
    want = orc.detect_pyramid(img)
    c2.set_corner_capacity(1536)
    f = c2.frame(img)
    assert f.corner_capacity() == 1536
    assert np.array_equal(c2.detect_corners([f], sdvl.default_detect_params(), 1000)[0], want)
    f.close()
    c2.set_corner_capacity(512)
    g = c2.frame(img)
    with pytest.raises(sdvl.SdvlError, match="capacity"):
        c2.detect_corners([g], sdvl.default_detect_params(), 1000)
    g.close()
    c2.close()


def test_feed_slots_and_own_images(ctx, sdvl, orc, synth):
    """sdvl_feed (one copy stream, slots with acquire / release) + sdvl_frame_borrow_image_device + sdvl_frames_own_images: frames
    alias a slot's buffers, one of them takes its image over, the slot is refilled with other images: the owner keeps the old image
    (level 0 and the pyramid it was built from), the alias sees the new one"""
    import torch
    imgs = frames_of(synth, orc, TUM_CAM, 640, 480, [0, 1, 2, 3])
    fb = 640 * 480
    pinned = torch.empty(4 * fb, dtype=torch.uint8, pin_memory=True)
    pinned.numpy()[:] = np.concatenate([im.reshape(-1) for im in imgs])
    slot = ctx.device_malloc(2 * fb)
    feed = sdvl.Feed(0, 1)
    feed.images(0, [pinned.data_ptr(), pinned.data_ptr() + fb], 640, 640, 480, [slot, slot + fb])     # images 0, 1 (one DMA: contiguous)
    import time
    t_end = time.time() + 10.0
    while not feed.arrived(0):                              # the consumer's host-side form of the wait (Farm::StepGroup)
        assert time.time() < t_end, "the slot's transfer never arrived"
        time.sleep(1e-4)
    with pytest.raises(sdvl.SdvlError):
        feed.arrived(1)                                     # no such slot
    feed.acquire(ctx, 0)
    fa, fo = sdvl.Frame(ctx, 640, 480), sdvl.Frame(ctx, 640, 480)
    fa.borrow_image_device(slot)
    fo.borrow_image_device(slot + fb)
    ctx.pyramid_build([fa, fo])
    assert np.array_equal(fa.level(0), imgs[0]) and np.array_equal(fo.level(0), imgs[1])
    ctx.own_images([fo])                                    # fo keeps image 1; fa still aliases the slot
    ctx.own_images([fo])                                    # a second call finds nothing to do
    feed.release(ctx, 0)
    feed.images(0, [pinned.data_ptr() + 2 * fb, pinned.data_ptr() + 3 * fb], 640, 640, 480, [slot, slot + fb])  # images 2, 3 into the same buffers
    feed.acquire(ctx, 0)
    assert np.array_equal(fo.level(0), imgs[1]) and np.array_equal(fo.level(1), orc.pyramid(imgs[1], 5)[1])
    assert np.array_equal(fa.level(0), imgs[2])
    fa.close(); fo.close(); feed.close(); ctx.device_free(slot)


def test_device_retain_best_is_libstdcxx_order(ctx, orc):
    """the device restatement of nth_element + partition leaves the list exactly as libstdc++ does"""
    rng = np.random.default_rng(42)
    cases = []
    for n in [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 33, 40, 64, 100, 169, 176, 500, 1500, 4000]:
        for hi in (4, 30, 255):                      # few distinct responses -> many ties; many -> few ties
            sc = rng.integers(1, hi + 1, n).astype(np.uint32)
            xy = np.arange(n, dtype=np.uint32)
            v = (xy & 0xFFF) | (((xy >> 12) & 0xFFF) << 12) | (sc << 24)
            for k in sorted({0, 1, 2, n // 3, n // 2, n - 1, n, n + 5}):
                cases.append((v, k))
    # adversarial shapes for introselect: sorted, reversed, organ pipe, constant
    n = 3000
    xy = np.arange(n, dtype=np.uint32)
    for sc in (np.arange(n) % 250 + 1, (n - np.arange(n)) % 250 + 1, np.minimum(np.arange(n), n - np.arange(n)) % 250 + 1, np.full(n, 9)):
        v = (xy & 0xFFF) | (((xy >> 12) & 0xFFF) << 12) | (sc.astype(np.uint32) << 24)
        for k in (1, 17, 395, 1500, 2999):
            cases.append((v, k))
    for v, k in cases:
        want = orc.retain_best(v, k)
        assert np.array_equal(ctx.retain_best(v, k), want), (len(v), k, "one lane")
        assert np.array_equal(ctx.retain_best(v, k, cooperative=True), want), (len(v), k, "workgroup")
        if len(v) <= 176:     # a cell's list: 16 lanes of a wave (the form select_cells uses per cell)
            assert np.array_equal(ctx.retain_best(v, k, cooperative=2), want), (len(v), k, "16 lanes")
        if len(v) <= 4096:    # a level's list: one wave (the form select_pack uses per level)
            assert np.array_equal(ctx.retain_best(v, k, cooperative=3), want), (len(v), k, "one wave")


@pytest.mark.parametrize("shape,cam,ks", [((480, 640), TUM_CAM, [0, 3, 11]), ((480, 752), EUROC_CAM, [2])])
def test_detect_corners_on_device_equals_detect_pyramid(ctx, sdvl, orc, synth, shape, cam, ks):
    """FastDetector::DetectPyramid entirely on device: same corners in the same order as the CPU path"""
    h, w = shape
    imgs = frames_of(synth, orc, cam, w, h, ks) + [rand_img(3, h, w)]
    fr = [ctx.frame(im) for im in imgs]
    for nfeat in (1000, 2000, 150):
        got = ctx.detect_corners(fr, sdvl.default_detect_params(), nfeat)
        for im, g in zip(imgs, got):
            want = orc.detect_pyramid(im, nfeatures=nfeat)
            assert np.array_equal(g, want), "nfeatures=%d" % nfeat
    for f in fr:
        f.close()


# ------------------------------------------------------------------------------------------------ K3 / K4
def test_shi_tomasi_and_orb_bit_exact(ctx, sdvl, orc, synth):
    img = frames_of(synth, orc, TUM_CAM, 640, 480, [2])[0]
    corners = orc.detect_pyramid(img)
    f = ctx.frame(img)
    f.set_corners(corners)
    pyr = orc.pyramid(img, 5)
    got = ctx.shi_tomasi([f])[0]
    want = np.array([orc.shi_tomasi(pyr[l], x, y) for x, y, l in corners])
    assert np.array_equal(got, want)
    desc = ctx.orb_describe([f])[0]
    for l in range(3):
        m = corners[:, 2] == l
        wd, wa = orc.orb_describe(pyr[l], corners[m][:, :2])
        assert np.array_equal(desc[m], wd), "ORB bits level %d" % l
        gd, ga = ctx.orb_describe_points(f, corners[m])
        assert np.array_equal(gd, wd) and np.array_equal(ga, wa)
    f.close()


def test_hamming_argmin_matches_search_features_rule(ctx, orc):
    """ORBDetector::Distance + the arg-min of Matcher::SearchFeatures (orb_detector.cc:398-410, matcher.cc:254-289):
    first strict minimum, not found at distance >= threshold, empty and ragged lists."""
    rng = np.random.default_rng(77)
    n = 300
    queries = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    lists = []
    for i in range(n):
        k = [0, 1, 3, 64, 65, 200, 777][i % 7]
        c = rng.integers(0, 256, (k, 32), dtype=np.uint8)
        if k >= 3:
            # near copies of the query (a few flipped bits) so distances fall on both sides of the threshold, with exact ties
            for j in rng.choice(k, 3, replace=False):
                c[j] = queries[i]
                flips = rng.choice(256, int(rng.integers(0, 130)), replace=False)
                for b in flips:
                    c[j, b >> 3] ^= 1 << (b & 7)
            a, b = sorted(rng.choice(k, 2, replace=False))
            c[b] = c[a]                                   # duplicate: the earlier position must win
        lists.append(c)
    for thr in (100, 0, 256, 37):
        gi, gd = ctx.hamming_argmin(queries, lists, thr)
        for i in range(n):
            best, bi = thr + 1, -1
            for j, c in enumerate(lists[i]):
                d = orc.orb_distance(queries[i], c)
                assert d == int(np.unpackbits(queries[i] ^ c).sum())
                if d < best:
                    best, bi = d, j
            if best >= thr:
                bi = -1
            assert (gi[i], gd[i]) == (bi, best), (thr, i)
    gi, gd = ctx.hamming_argmin(np.zeros((0, 32), np.uint8), [], 100)
    assert len(gi) == 0


# ------------------------------------------------------------------------------------------------ K5/K6
def align_inputs(sdvl, orc, img0, cam, n_feat, seed):
    """features of frame 0 on the scene plane z = 2 (camera 0 = world)."""
    h, w = img0.shape
    rng = np.random.default_rng(seed)
    px = np.stack([rng.uniform(48, w - 48, n_feat), rng.uniform(48, h - 48, n_feat)], 1)
    ray = np.stack([(px[:, 0] - cam[2]) / cam[0], (px[:, 1] - cam[3]) / cam[1], np.ones(n_feat)], 1)
    bearing = ray / np.linalg.norm(ray, axis=1, keepdims=True)
    depth = 2.0 / bearing[:, 2]
    valid = np.ones(n_feat, np.uint8)
    valid[::17] = 0
    feats = (sdvl.AlignFeature * n_feat)()
    for i in range(n_feat):
        feats[i].px, feats[i].py = px[i]
        feats[i].fx, feats[i].fy, feats[i].fz = bearing[i]
        feats[i].depth = depth[i]
        feats[i].valid = int(valid[i])
    return px, bearing, depth, valid, feats


@pytest.mark.parametrize("n_feat,k,fast", [(200, 3, False), (1000, 5, False), (200, 40, True), (7, 2, False)])
def test_image_align_pose_within_tolerance(ctx, sdvl, orc, synth, n_feat, k, fast):
    img0, imgk = frames_of(synth, orc, TUM_CAM, 640, 480, [0, k])
    px, bearing, depth, valid, feats = align_inputs(sdvl, orc, img0, TUM_CAM, n_feat, 20260200)
    T0 = np.array([1, 0, 0, 0, 0, 0, 0], np.float64)
    want = orc.image_align(img0, imgk, TUM_CAM, px, bearing, depth, valid, T0, fast=fast)
    f0, fk = ctx.frame(img0), ctx.frame(imgk)
    cam = sdvl.Camera(640, 480, *TUM_CAM)
    res = ctx.image_align([(f0, fk, 0, n_feat, T0)], feats, cam, sdvl.default_align_params(fast))[0]
    got = np.array(res.T[:])
    assert np.abs(got - want["T"]).max() <= POSE_TOL
    assert res.n_meas == want["n"]
    assert abs(res.error - want["error"]) <= 1e-4 * max(1.0, abs(want["error"])) or (res.error >= 1e9 and want["error"] >= 1e9)
    assert np.abs(np.array(res.its[:]) - want["its"]).max() <= 1          # +-1 GN step near convergence is allowed
    if not fast:
        Tgt = trajectory_pose(orc, k)
        assert np.abs(got - Tgt).max() < 5e-3                               # and it actually aligns the frames
    f0.close(); fk.close()


def test_image_align_batch_of_jobs_and_empty(ctx, sdvl, orc, synth):
    imgs = frames_of(synth, orc, TUM_CAM, 640, 480, [0, 2, 4, 6])
    px, bearing, depth, valid, feats = align_inputs(sdvl, orc, imgs[0], TUM_CAM, 300, 7)
    fr = [ctx.frame(im) for im in imgs]
    T0 = np.array([1, 0, 0, 0, 0, 0, 0], np.float64)
    jobs = [(fr[0], fr[1], 0, 300, T0), (fr[0], fr[2], 0, 150, T0), (fr[0], fr[3], 150, 300, T0), (fr[0], fr[1], 10, 10, T0)]
    cam = sdvl.Camera(640, 480, *TUM_CAM)
    res = ctx.image_align(jobs, feats, cam, sdvl.default_align_params())
    for (ref, cur, b, e, T), r, im in zip(jobs[:3], res, imgs[1:]):
        want = orc.image_align(imgs[0], im, TUM_CAM, px[b:e], bearing[b:e], depth[b:e], valid[b:e], T0)
        assert np.abs(np.array(r.T[:]) - want["T"]).max() <= POSE_TOL
        assert r.n_meas == want["n"]
    assert np.array_equal(np.array(res[3].T[:]), T0) and res[3].n_meas == 0      # no features: pose untouched
    for f in fr:
        f.close()


def test_filter_corners_selection_on_the_device(ctx, sdvl, orc, synth):
    """sdvl_filter_corners_begin / _end = Frame::FilterCorners with the per-cell selection of FastDetector::FilterCorners done
    on the device: the kept corner indices (cell order), their truncated scores and ORB descriptors equal the oracle's, with
    some cells locked; on a frame whose corners crowd into few cells (more than the kernel's per-cell bins hold: the scan
    fallback) and on a normal frame"""
    img = frames_of(synth, orc, TUM_CAM, 640, 480, [2])[0]
    corners = orc.detect_pyramid(img)
    # a second corner list of the same image: every level-0 corner of a few cells (FAST without the quota), > 10 per cell
    dense = orc.fast(img, thr=10, nonmax=True)
    dense = dense[(dense[:, 0] >= 64) & (dense[:, 0] < 160) & (dense[:, 1] >= 64) & (dense[:, 1] < 128)][:600]
    dense = np.concatenate([dense[:, :2], np.zeros((len(dense), 1), np.int32)], 1).astype(np.int32)
    assert len(dense) > 100
    pyr = orc.pyramid(img, 5)
    for clist, locked in ((corners, [[100.0, 100.0], [333.3, 250.1], [620.0, 470.0]]), (dense, [[70.0, 70.0]]), (corners, [])):
        f = ctx.frame(img)
        f.set_corners(clist)
        idx, xyl, score, desc = ctx.filter_corners([f], [locked])[0]
        want = orc.filter_corners(img, clist, locked if locked else None)
        assert np.array_equal(idx, want), (len(idx), len(want))
        assert np.array_equal(xyl, clist[want])
        for k in range(0, len(idx), 7):
            x, y, l = xyl[k]
            assert score[k] == int(orc.shi_tomasi(pyr[l], x, y))
            d, _ = orc.orb_describe(pyr[l], [[x, y]])
            inside = 19 <= x < pyr[l].shape[1] - 19 and 19 <= y < pyr[l].shape[0] - 19
            assert np.array_equal(desc[k], d[0] if inside else np.zeros(32, np.uint8))
        f.close()
    cells = {}
    for x, y, _ in dense:
        cells[(x // 32, y // 32)] = cells.get((x // 32, y // 32), 0) + 1
    assert max(cells.values()) > 10      # the dense list really overflows a bin


# ------------------------------------------------------------------------------------------------ K7
def search_requests(sdvl, orc, ctx, img_ref, img_cur, T_ref, T_cur, cam4, n_req, seed, fixed, noise=0.0, describe=True):
    """points seeded on the reference frame's corners (plane z=2 in world = camera 0), searched in the current frame"""
    h, w = img_ref.shape
    rng = np.random.default_rng(seed)
    cref = orc.detect_pyramid(img_ref)
    ccur = orc.detect_pyramid(img_cur)
    pyr_ref = orc.pyramid(img_ref, 5)
    sel = rng.choice(len(cref), size=min(n_req, len(cref)), replace=False)
    f_ref, f_cur = ctx.frame(img_ref), ctx.frame(img_cur)
    f_cur.set_corners(ccur)
    if describe:   # otherwise the search computes the descriptors it compares on the spot (matcher.cc:266-269)
        ctx.orb_describe([f_cur], want=False)
    Tw = orc.se3_inv(T_ref)
    from oraclelib import quat_to_R
    Rw, tw = quat_to_R(Tw[:4]), Tw[4:]
    Rc, tc = quat_to_R(T_cur[:4]), T_cur[4:]
    reqs = (sdvl.SearchReq * len(sel))()
    meta = []
    for i, ci in enumerate(sel):
        x, y, l = cref[ci]
        px = np.array([x * (1 << l), y * (1 << l)], np.float64)
        ray = np.array([(px[0] - cam4[2]) / cam4[0], (px[1] - cam4[3]) / cam4[1], 1.0])
        bearing = ray / np.linalg.norm(ray)
        rw = Rw @ bearing
        s = (2.0 - tw[2]) / rw[2]
        desc, _ = orc.orb_describe(pyr_ref[l], [[x, y]])
        idepth = 1.0 / s * (1.0 + noise * rng.normal())
        istd = 0.05 * idepth if fixed else 0.1 * idepth
        Pw = Rw @ (bearing * s) + tw
        pc = Rc @ Pw + tc
        px0 = np.array([cam4[2] + cam4[0] * pc[0] / pc[2], cam4[3] + cam4[1] * pc[1] / pc[2]]) + rng.normal(size=2) * 0.7
        r = reqs[i]
        r.cur, r.ref = f_cur.h.value, f_ref.h.value
        for k in range(7):
            r.cur_pose[k], r.ref_pose[k] = T_cur[k], T_ref[k]
        r.px[0], r.px[1] = px
        r.bearing[0], r.bearing[1], r.bearing[2] = bearing
        r.idepth, r.idepth_std = idepth, istd
        r.px0[0], r.px0[1] = px0
        r.level, r.fixed = int(l), int(fixed)
        for k in range(32):
            r.desc[k] = int(desc[0, k])
        meta.append(dict(px=px, bearing=bearing, level=int(l), desc=desc[0], idepth=idepth, istd=istd, px0=px0))
    return reqs, meta, ccur, f_ref, f_cur


@pytest.mark.parametrize("describe", [True, False], ids=["descriptors-in-hbm", "descriptors-on-demand"])
@pytest.mark.parametrize("fixed,k_ref,k_cur,noise", [(True, 0, 4, 0.0), (False, 0, 6, 0.02), (True, 10, 3, 0.0), (False, 2, 30, 0.05)])
def test_search_points_matches_oracle(ctx, sdvl, orc, synth, fixed, k_ref, k_cur, noise, describe):
    img_ref, img_cur = frames_of(synth, orc, TUM_CAM, 640, 480, [k_ref, k_cur])
    T_ref, T_cur = trajectory_pose(orc, k_ref), trajectory_pose(orc, k_cur)
    reqs, meta, ccur, f_ref, f_cur = search_requests(sdvl, orc, ctx, img_ref, img_cur, T_ref, T_cur, TUM_CAM, 160, 11, fixed, noise, describe)
    cam = sdvl.Camera(640, 480, *TUM_CAM)
    res = ctx.search_points(reqs, cam, sdvl.default_search_params())
    n_found = 0
    for r, m in zip(res, meta):
        want = orc.search_point(img_ref, img_cur, TUM_CAM, T_ref, T_cur, m["px"], m["bearing"], m["level"], m["desc"],
                                m["idepth"], m["istd"], fixed, ccur, m["px0"])
        assert r.found == want["found"]
        if r.slevel >= 0:
            assert r.slevel == want["slevel"]
        if want["found"]:
            n_found += 1
            assert r.level == want["level"]
            assert np.abs(np.array(r.px[:]) - want["px"]).max() <= POSE_TOL
            assert np.array_equal(np.array(r.px[:]), want["px"])      # same op order, no contraction: bit-identical
    assert n_found >= (40 if fixed else 10)
    if not describe:   # asking for the frame's descriptors afterwards computes all of them, equal to the oracle's
        pyr = orc.pyramid(img_cur, 5)
        got = f_cur.descriptors()
        for l in range(3):
            idx = np.nonzero(ccur[:, 2] == l)[0]
            want_d, _ = orc.orb_describe(pyr[l], ccur[idx, :2])
            assert np.array_equal(got[idx], want_d)
    f_ref.close(); f_cur.close()


def test_search_points_zero_baseline_epipolar_on_a_binned_frame(ctx, sdvl, orc, synth):
    """cur pose == ref pose, epipolar (not fixed) requests: the depth interval projects onto one pixel, the line constants of
    GetCornersInRange are NaN (matcher.cc:139-148) and every range test is false — every corner that passes the level and margin
    tests is a candidate.  On a frame whose corners are binned (sdvl_detect_corners) the search must not narrow that to the cells
    around the pixel: same results as the oracle and as the unbinned frame."""
    img, = frames_of(synth, orc, TUM_CAM, 640, 480, [3])
    T = trajectory_pose(orc, 3)
    reqs, meta, ccur, f_ref, f_cur = search_requests(sdvl, orc, ctx, img, img, T, T, TUM_CAM, 64, 23, False, 0.0, False)
    cam = sdvl.Camera(640, 480, *TUM_CAM)
    plain = ctx.search_points(reqs, cam, sdvl.default_search_params())          # set_corners: no bins, full scan
    f_bin = ctx.frame(img)
    ctx.detect_corners([f_bin], sdvl.default_detect_params(), 1000)              # bins valid; the same corner list (tested elsewhere)
    for r in reqs:
        r.cur = f_bin.h.value
    binned = ctx.search_points(reqs, cam, sdvl.default_search_params())
    n_found = 0
    for a, b, m in zip(plain, binned, meta):
        want = orc.search_point(img, img, TUM_CAM, T, T, m["px"], m["bearing"], m["level"], m["desc"], m["idepth"], m["istd"], False, ccur, m["px0"])
        assert a.found == b.found == want["found"] and a.best_corner == b.best_corner
        assert tuple(a.px) == tuple(b.px)
        if want["found"]:
            n_found += 1
            assert np.array_equal(np.array(b.px[:]), want["px"]) and b.level == want["level"]
    assert n_found >= 20
    f_ref.close(); f_cur.close(); f_bin.close()


@pytest.mark.parametrize("k_cur,noise", [(6, 0.02), (30, 0.05), (60, 0.2)])
def test_depth_filter_behind_the_search_matches_oracle(ctx, sdvl, orc, synth, k_cur, noise):
    """sdvl_search_points_filter: the mapper's depth filter (triangulation, parallax, the minimum-depth tests, Point::Update,
    HasConverged; Unpromote for a miss: map.cc:454-497, point.cc:64-118,164-178) on the device behind the candidate search,
    against the oracle's statement-level restatement fed with the same search results.  Decisions and counters exact; the
    filter state within 1e-11 relative (acos / sin / exp are the device library's, not glibc's)."""
    import math
    img_ref, img_cur = frames_of(synth, orc, TUM_CAM, 640, 480, [0, k_cur])
    T_ref, T_cur = trajectory_pose(orc, 0), trajectory_pose(orc, k_cur)
    reqs, meta, ccur, f_ref, f_cur = search_requests(sdvl, orc, ctx, img_ref, img_cur, T_ref, T_cur, TUM_CAM, 300, 17 + k_cur, False, noise, False)
    n = len(reqs)
    rng = np.random.default_rng(5)
    ref = orc.tracker(640, 480, TUM_CAM)
    ref.use_mapper(True)
    max_failed = 15
    states = (sdvl.DepthState * n)()
    for i, m in enumerate(meta):
        s = states[i]
        s.rho, s.sigma2 = m["idepth"], m["istd"] ** 2
        s.a, s.b = 10.0 + 5.0 * rng.random(), 10.0 + 8.0 * rng.random()
        s.z_range = math.sqrt(36.0)
        s.cos_alpha, s.last_distance = 1.0, 1.0
        s.depth_mean = 2.0 if i % 7 else 40.0          # a scene so deep that the candidate fails the minimum-depth test
        s.fixed = 1 if i % 11 == 3 else 0
        for k in range(3):
            s.position[k] = [0.1 * (i % 5), -0.05 * (i % 3), 2.0][k]
        s.n_failed = int(rng.integers(0, max_failed + 1))
        s.track_row = -1
    fp = sdvl.DepthParams()
    fp.px_error_angle = math.atan(1.0 / (2.0 * TUM_CAM[0])) * 2.0
    fp.min_depth, fp.scale_min_dist, fp.max_failed = 1.0 * 0.25, 0.25, max_failed
    cam = sdvl.Camera(640, 480, *TUM_CAM)
    res, fout = ctx.search_points_filter(reqs, cam, sdvl.default_search_params(), states, fp)
    plain = ctx.search_points(reqs, cam, sdvl.default_search_params())
    seen = {}
    for i in range(n):
        r, o, s = res[i], fout[i], states[i]
        assert (r.found, tuple(r.px)) == (plain[i].found, tuple(plain[i].px))      # the search itself is untouched
        st0 = [s.rho, s.sigma2, s.a, s.b, s.z_range, s.cos_alpha, s.last_distance, s.position[0], s.position[1], s.position[2], s.fixed, s.n_failed]
        want, st1 = ref.depth_filter(T_cur, T_ref, meta[i]["bearing"], r.found, r.px[:], s.depth_mean, st0)
        assert o.outcome == want, (i, o.outcome, want)
        assert o.n_failed == int(st1[11])
        seen[want & 0xFF] = seen.get(want & 0xFF, 0) + 1
        if want & 0x100:
            seen["deleted"] = seen.get("deleted", 0) + 1
        got = np.array([o.rho, o.sigma2, o.a, o.b])
        assert np.all(np.abs(got - st1[:4]) <= 1e-11 * np.abs(st1[:4])), (i, got, st1[:4])
        if (want & 0xFF) in (2, 3):
            assert abs(o.cos_alpha - st1[5]) <= 1e-11 and abs(o.last_distance - st1[6]) <= 1e-11 * st1[6]
            if (want & 0xFF) == 3:      # the position HasConverged froze (p3d_)
                assert np.abs(np.array(o.position[:]) - st1[7:10]).max() <= 1e-11
    # every branch of the loop body was exercised
    assert seen.get(0, 0) > 5 and seen.get(1, 0) > 5 and seen.get(2, 0) + seen.get(3, 0) > 20 and seen.get(3, 0) > 0 and seen.get("deleted", 0) > 0, seen
    f_ref.close(); f_cur.close()


def test_search_points_tree_sums_within_tolerance(ctx, sdvl, orc, synth):
    """sdvl_search_params.lk_tree_sums = 1: AlignPatch's three 64-term sums as a wave butterfly instead of the reference's
    sequential chain.  Tolerance class, not bit class: found flags equal for (almost) every request, offsets within 1e-4."""
    img_ref, img_cur = frames_of(synth, orc, TUM_CAM, 640, 480, [0, 4])
    T_ref, T_cur = trajectory_pose(orc, 0), trajectory_pose(orc, 4)
    reqs, meta, ccur, f_ref, f_cur = search_requests(sdvl, orc, ctx, img_ref, img_cur, T_ref, T_cur, TUM_CAM, 400, 31, True, 0.0, False)
    cam = sdvl.Camera(640, 480, *TUM_CAM)
    exact = ctx.search_points(reqs, cam, sdvl.default_search_params())
    sp = sdvl.default_search_params()
    sp.lk_tree_sums = 1
    tree = ctx.search_points(reqs, cam, sp)
    both, flipped, worst = 0, 0, 0.0
    for a, b in zip(exact, tree):
        assert a.best_corner == b.best_corner and a.slevel == b.slevel      # everything before AlignPatch is untouched
        if a.found != b.found:
            flipped += 1
        elif a.found:
            both += 1
            worst = max(worst, abs(a.px[0] - b.px[0]), abs(a.px[1] - b.px[1]))
    assert both >= 150 and flipped <= 2 and worst <= 1e-4, (both, flipped, worst)
    f_ref.close(); f_cur.close()


def test_search_points_frame_with_more_corners_than_the_lds_stage(ctx, sdvl, orc, synth):
    """1280x960 with num_features 4600: the current frame holds more corners than search_points stages in LDS (4096), the
    rest (coarse-level corners at the end of the list) are read from HBM — found flags, levels and offsets as the oracle's"""
    cam4 = np.array(TUM_CAM) * 2.0
    old = orc.params.num_features
    orc.params.num_features = 4600
    try:
        img_ref, img_cur = frames_of(synth, orc, cam4, 1280, 960, [0, 3])
        T_ref, T_cur = trajectory_pose(orc, 0), trajectory_pose(orc, 3)
        reqs, meta, ccur, f_ref, f_cur = search_requests(sdvl, orc, ctx, img_ref, img_cur, T_ref, T_cur, cam4, 400, 21, True, 0.0, False)
    finally:
        orc.params.num_features = old
    assert 4096 < len(ccur) <= 6144
    cam = sdvl.Camera(1280, 960, *cam4)
    res = ctx.search_points(reqs, cam, sdvl.default_search_params())
    beyond = 0
    for r, m in zip(res, meta):
        want = orc.search_point(img_ref, img_cur, cam4, T_ref, T_cur, m["px"], m["bearing"], m["level"], m["desc"],
                                m["idepth"], m["istd"], True, ccur, m["px0"])
        assert r.found == want["found"]
        if want["found"]:
            assert r.level == want["level"] and np.array_equal(np.array(r.px[:]), want["px"])
        if r.best_corner >= 4096:
            beyond += 1
    assert beyond >= 5          # matches among the corners that live outside the LDS stage
    f_ref.close(); f_cur.close()


def chain_case(ctx, sdvl, orc, synth, n_req=300):
    """requests of one frame pair + four trackers over them for sdvl_search_run_chain"""
    img_ref, img_cur = frames_of(synth, orc, TUM_CAM, 640, 480, [0, 5])
    T_ref, T_cur = trajectory_pose(orc, 0), trajectory_pose(orc, 5)
    reqs, meta, ccur, f_ref, f_cur = search_requests(sdvl, orc, ctx, img_ref, img_cur, T_ref, T_cur, TUM_CAM, n_req, 21, True, 0.0, describe=False)
    n = len(reqs)
    cam = sdvl.Camera(640, 480, *TUM_CAM)
    sp = sdvl.default_search_params()
    rng = np.random.default_rng(99)
    # the points the requests stand for (plane z = 2 in world = camera 0), as the host would hand them over
    from oraclelib import quat_to_R
    Tw = orc.se3_inv(T_ref)
    Rw, tw = quat_to_R(Tw[:4]), Tw[4:]
    pts = np.stack([Rw @ (m["bearing"] / m["idepth"]) + tw for m in meta])
    # two trackers sharing the frame pair: cells of 1-3 candidates, a few candidates without a request, different caps
    trackers = []
    for lo, hi, cap in ((0, n // 2, 40), (n // 2, n, 200)):
        cells, k = [], lo
        while k < hi:
            c = list(range(k, min(hi, k + int(rng.integers(1, 4)))))
            if rng.random() < 0.1:
                c.insert(int(rng.integers(0, len(c) + 1)), -1)
            cells.append(c)
            k = c[-1] + 1 if c[-1] >= 0 else max(x for x in c if x >= 0) + 1
        trackers.append(dict(cells=cells, max_matches=cap, pose=T_cur, draws=rng.integers(0, 2**31 - 1, 100)))
    # a tracker that projected nothing into the image (no candidates) and one whose only candidates have no request
    trackers.append(dict(cells=[], max_matches=200, pose=T_ref, draws=rng.integers(0, 2**31 - 1, 100)))
    trackers.append(dict(cells=[[-1], [-1, -1]], max_matches=200, pose=T_cur, draws=rng.integers(0, 2**31 - 1, 100)))
    return reqs, cam, sp, trackers, pts, f_ref, f_cur


def test_search_chain_equals_search_then_select_then_pose(ctx, sdvl, orc, synth):
    """sdvl_search_run_chain / _chain_end against the separate calls: the same search results, the matches the host replay of
    SelectPoints (first hit per cell, at most max_matches) picks, and bit-identical pose results from them"""
    reqs, cam, sp, trackers, pts, f_ref, f_cur = chain_case(ctx, sdvl, orc, synth)
    res, got = ctx.search_chain(reqs, cam, sp, trackers, pts, fx=TUM_CAM[0])
    want_res = ctx.search_points(reqs, cam, sp)
    for a, b in zip(res, want_res):
        assert (a.found, a.level, a.best_corner, a.stage, a.lk_its, a.slevel, a.px[0], a.px[1]) == \
               (b.found, b.level, b.best_corner, b.stage, b.lk_its, b.slevel, b.px[0], b.px[1])
    jobs = []
    for t in trackers:
        sel = []
        for cell in t["cells"]:
            if len(sel) >= t["max_matches"]:
                break
            for r in cell:
                if r >= 0 and res[r].found:
                    sel.append(r)
                    break
        obs = []
        for r in sel:
            x, y = (res[r].px[0] - TUM_CAM[2]) / TUM_CAM[0], (res[r].px[1] - TUM_CAM[3]) / TUM_CAM[1]
            nrm = np.sqrt(x * x + y * y + 1.0)
            v = np.array([x / nrm, y / nrm, 1.0 / nrm])          # Camera::Unproject
            obs.append([v[0] / v[2], v[1] / v[2], pts[r, 0], pts[r, 1], pts[r, 2], res[r].level])
        jobs.append((np.array(obs).reshape(-1, 6), t["pose"], t["draws"]))
    want = ctx.pose_from_matches(jobs, fx=TUM_CAM[0])
    assert got[0]["n_obs"] == 40 and got[1]["n_obs"] > 60
    for g, t in zip(got[2:], trackers[2:]):     # nothing to select: SelectInliers / OptimizePose leave the pose alone, draw nothing
        assert g["n_obs"] == 0 and g["n_draws"] == 0 and g["refined"] == 0 and len(g["inliers"]) == 0 and len(g["outliers"]) == 0
        assert np.array_equal(g["pose"], np.asarray(t["pose"], np.float64))
    for g, w, j in zip(got, want, jobs):
        assert g["n_obs"] == len(j[0])
        assert g["n_draws"] == w["n_draws"] and g["refined"] == w["refined"]
        assert np.array_equal(g["inliers"], w["inliers"]) and np.array_equal(g["outliers"], w["outliers"])
        assert np.array_equal(g["pose"], w["pose"])
    f_ref.close(); f_cur.close()


def test_search_chain_survives_buffer_growth_inside_the_call(ctx, sdvl, orc, synth):
    """A request batch is filled between sdvl_search_begin and sdvl_search_run_chain; the run grows the context's result and
    staging buffers when they are too small, and growing waits on the stream.  The batch must survive that (it used to live
    in the staging ring, whose bump pointer every wait reset): a fresh context — first a small chain, then one that forces
    every buffer to grow while its batch is open — returns what a warmed-up context returns."""
    reqs, cam, sp, trackers, pts, f_ref, f_cur = chain_case(ctx, sdvl, orc, synth, 300)
    ctx.search_chain(reqs, cam, sp, trackers, pts, fx=TUM_CAM[0])               # warm: nothing grows in the next call
    want_res, want = ctx.search_chain(reqs, cam, sp, trackers, pts, fx=TUM_CAM[0])
    fresh = sdvl.Context(0)
    try:
        small = chain_case(fresh, sdvl, orc, synth, 12)
        fresh.search_chain(small[0], small[1], small[2], small[3], small[4], fx=TUM_CAM[0])
        reqs2, cam2, sp2, trackers2, pts2, g_ref, g_cur = chain_case(fresh, sdvl, orc, synth, 300)
        got_res, got = fresh.search_chain(reqs2, cam2, sp2, trackers2, pts2, fx=TUM_CAM[0])
        for a, b in zip(got_res, want_res):
            assert (a.found, a.level, a.best_corner, a.stage, a.lk_its, a.slevel, a.px[0], a.px[1]) == \
                   (b.found, b.level, b.best_corner, b.stage, b.lk_its, b.slevel, b.px[0], b.px[1])
        for g, w in zip(got, want):
            assert g["n_obs"] == w["n_obs"] and g["n_draws"] == w["n_draws"]
            assert np.array_equal(g["inliers"], w["inliers"]) and np.array_equal(g["outliers"], w["outliers"]) and np.array_equal(g["pose"], w["pose"])
        for f in (small[5], small[6], g_ref, g_cur):
            f.close()
    finally:
        fresh.close()
    f_ref.close(); f_cur.close()


def test_context_of_a_fresh_thread_allocates_on_its_own_gpu(sdvl):
    """HIP's current device is per thread and starts at 0: a context's scratch must land on the context's GPU whichever thread
    grows it (farm workers, fibers, a mapper thread).  With one GPU the check is trivial but still runs the code path."""
    import ctypes as C
    import threading
    import torch
    lib = sdvl.load_library()
    dev = torch.cuda.device_count() - 1
    out = {}

    def work():
        h = C.c_void_p()
        assert lib.sdvl_ctx_create(dev, C.byref(h)) == 0
        d = C.c_int(-1)
        rc = lib.sdvl_ctx_scratch_device(h, C.byref(d))
        out["rc"], out["scratch"] = rc, d.value
        out["ctx_dev"] = lib.sdvl_ctx_device(h)
        lib.sdvl_ctx_destroy(h)

    def grow_elsewhere():
        # the context is created on this thread, its buffers grow on ANOTHER fresh thread (whose current device is 0)
        h = C.c_void_p()
        assert lib.sdvl_ctx_create(dev, C.byref(h)) == 0
        res = {}

        def grow():
            d = C.c_int(-1)
            res["rc"] = lib.sdvl_ctx_scratch_device(h, C.byref(d))
            res["dev"] = d.value
        t2 = threading.Thread(target=grow)
        t2.start(); t2.join()
        out["rc2"], out["scratch2"] = res["rc"], res["dev"]
        lib.sdvl_ctx_destroy(h)

    for fn in (work, grow_elsewhere):
        t = threading.Thread(target=fn)
        t.start(); t.join()
    assert out["rc"] == 0 and out["rc2"] == 0
    assert out["ctx_dev"] == dev and out["scratch"] == dev and out["scratch2"] == dev


def test_frames_upload_from_pinned_and_pageable_memory(ctx, sdvl, orc):
    """sdvl_frames_upload: pinned images are pulled by the gather kernel (contiguous and padded rows), pageable ones go through
    staged copies; either way level 0 is the image and the pyramid built from it is the oracle's"""
    import torch
    rng = np.random.default_rng(9)
    n, w, h = 5, 640, 480
    frames = [sdvl.Frame(ctx, w, h) for _ in range(n)]
    pinned = torch.empty((n, h, w), dtype=torch.uint8, pin_memory=True)
    imgs = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    pinned.numpy()[:] = imgs
    ctx.frames_upload(frames, [pinned.data_ptr() + i * w * h for i in range(n)], w)
    ctx._check(ctx.lib.sdvl_pyramid_build(ctx.h, n, (C.c_void_p * n)(*[f.h for f in frames])))
    for i in range(n):
        assert np.array_equal(frames[i].level(0), imgs[i])
        assert np.array_equal(frames[i].level(2), orc.pyramid(imgs[i], 5)[2])
    # padded rows (pitch 656) in pinned memory, and a mix of pinned and pageable sources in one call
    pitch = 656
    padded = torch.zeros((n, h, pitch), dtype=torch.uint8, pin_memory=True)
    imgs2 = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    padded.numpy()[:, :, :w] = imgs2
    pageable = np.zeros((h, pitch), np.uint8)
    imgs2[3] = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    pageable[:, :w] = imgs2[3]
    addrs = [padded.data_ptr() + i * pitch * h for i in range(n)]
    addrs[3] = pageable.ctypes.data
    ctx.frames_upload(frames, addrs, pitch)
    ctx.synchronize()
    for i in range(n):
        assert np.array_equal(frames[i].level(0), imgs2[i]), i
    # an odd source address (not 16-byte aligned) takes the row path
    odd = torch.zeros(w * h + 64, dtype=torch.uint8, pin_memory=True)
    odd.numpy()[3:3 + w * h] = imgs[1].reshape(-1)
    ctx.frames_upload(frames[:1], [odd.data_ptr() + 3], w)
    assert np.array_equal(frames[0].level(0), imgs[1])
    for f in frames:
        f.close()


def test_align_patches_bit_exact_and_recovers_shift(ctx, sdvl, orc, synth):
    img = frames_of(synth, orc, TUM_CAM, 640, 480, [0])[0]
    corners = orc.detect_pyramid(img)
    c0 = corners[corners[:, 2] == 0][:200]
    rng = np.random.default_rng(20260300)
    shift = rng.uniform(-2, 2, (len(c0), 2))
    f = ctx.frame(img)
    border = np.stack([img[y - 5:y + 5, x - 5:x + 5].reshape(-1) for x, y, _ in c0])      # identity warp, matcher.cc:338-341
    patch = np.stack([img[y - 4:y + 4, x - 4:x + 4].reshape(-1) for x, y, _ in c0])
    uv0 = c0[:, :2] + shift
    uv, conv, its = ctx.align_patches([f] * len(c0), np.zeros(len(c0), np.int32), border, patch, uv0)
    n_conv = 0
    for i in range(len(c0)):
        ok, px = orc.align_patch(img, border[i], patch[i], uv0[i])
        assert bool(conv[i]) == bool(ok)
        assert np.array_equal(uv[i], px)
        if ok:
            n_conv += 1
            assert np.abs(px - c0[i, :2]).max() < 0.2      # LK pulls the shifted start back onto the corner
    assert n_conv > 150
    # window leaving the image -> break, not converged (Appendix B)
    uv, conv, its = ctx.align_patches([f], [0], border[:1], patch[:1], [[2.5, 100.0]])
    assert conv[0] == 0 and its[0] == 0
    f.close()


def test_errors_are_reported_not_swallowed(ctx, sdvl):
    with pytest.raises(sdvl.SdvlError):
        ctx.frame(width=8, height=8)                         # too small
    f = ctx.frame(np.zeros((480, 640), np.uint8))
    with pytest.raises(sdvl.SdvlError):
        f.set_corners([[700, 10, 0]])                        # outside the level image
    with pytest.raises(sdvl.SdvlError):
        f.set_corners(np.zeros((sdvl.MAX_CORNERS + 1, 3), np.int32))   # capacity
    f.close()


# ------------------------------------------------------------------------------------------------ more cases
def test_search_points_zmssd_mode(ctx, sdvl, orc, synth):
    """use_orb = 0 (the config.cc default): margin 1 + PatchSize/2, best corner by integer ZMSSD (matcher.cc:447-476)"""
    img_ref, img_cur = frames_of(synth, orc, TUM_CAM, 640, 480, [0, 4])
    T_ref, T_cur = trajectory_pose(orc, 0), trajectory_pose(orc, 4)
    orc.params.use_orb = 0
    try:
        rng = np.random.default_rng(5)
        cref, ccur = orc.detect_pyramid(img_ref), orc.detect_pyramid(img_cur)
        f_ref, f_cur = ctx.frame(img_ref), ctx.frame(img_cur)
        f_cur.set_corners(ccur)
        sel = rng.choice(len(cref), size=120, replace=False)
        reqs = (sdvl.SearchReq * len(sel))()
        meta = []
        from oraclelib import quat_to_R
        Rc, tc = quat_to_R(T_cur[:4]), T_cur[4:]
        for i, ci in enumerate(sel):
            x, y, l = cref[ci]
            px = np.array([x * (1 << l), y * (1 << l)], np.float64)
            ray = np.array([(px[0] - TUM_CAM[2]) / TUM_CAM[0], (px[1] - TUM_CAM[3]) / TUM_CAM[1], 1.0])
            bearing = ray / np.linalg.norm(ray)
            s = 2.0 / bearing[2]
            pc = Rc @ (bearing * s) + tc
            px0 = np.array([TUM_CAM[2] + TUM_CAM[0] * pc[0] / pc[2], TUM_CAM[3] + TUM_CAM[1] * pc[1] / pc[2]]) + rng.normal(size=2) * 0.5
            r = reqs[i]
            r.cur, r.ref = f_cur.h.value, f_ref.h.value
            for k in range(7):
                r.cur_pose[k], r.ref_pose[k] = T_cur[k], T_ref[k]
            r.px[0], r.px[1] = px
            r.bearing[0], r.bearing[1], r.bearing[2] = bearing
            r.idepth, r.idepth_std = 1.0 / s, 0.05 / s
            r.px0[0], r.px0[1] = px0
            r.level, r.fixed = int(l), 1
            meta.append((px, bearing, int(l), 1.0 / s, 0.05 / s, px0))
        res = ctx.search_points(reqs, sdvl.Camera(640, 480, *TUM_CAM), sdvl.default_search_params(use_orb=False))
        n_found = 0
        for r, (px, bearing, l, idep, istd, px0) in zip(res, meta):
            want = orc.search_point(img_ref, img_cur, TUM_CAM, T_ref, T_cur, px, bearing, l, np.zeros(32, np.uint8), idep, istd, True, ccur, px0)
            assert r.found == want["found"]
            if want["found"]:
                n_found += 1
                assert np.array_equal(np.array(r.px[:]), want["px"]) and r.level == want["level"]
        assert n_found >= 30
        f_ref.close(); f_cur.close()
    finally:
        orc.params.use_orb = 1


def test_detect_corners_config_c_size(ctx, sdvl, orc):
    """1280x960 (BASELINE config 5): 1200 + 300 + 80 cells, 4000 features"""
    img = rand_img(9, 960, 1280)
    img = (img.astype(np.float32) * 0.25 + 96).astype(np.uint8)      # lower contrast: fewer, less tied corners
    f = ctx.frame(img)
    got = ctx.detect_corners([f], sdvl.default_detect_params(), 4000)[0]
    assert len(got) <= 6144
    want = orc.detect_pyramid(img, nfeatures=4000)
    assert np.array_equal(got, want)
    f.close()


def test_empty_batches_and_flat_frames(ctx, sdvl, orc):
    lib = ctx.lib
    assert lib.sdvl_pyramid_build(ctx.h, 0, None) == 0
    assert lib.sdvl_search_points(ctx.h, 0, None, C.byref(sdvl.Camera(640, 480, *TUM_CAM)), C.byref(sdvl.default_search_params()), None) == 0
    flat = np.full((480, 640), 128, np.uint8)
    f = ctx.frame(flat)
    assert len(ctx.detect_corners([f], sdvl.default_detect_params(), 1000)[0]) == 0
    assert ctx.orb_describe([f])[0].shape == (0, 32)
    f.close()


# ------------------------------------------------------------------------------------------------ K8
def make_matches(orc, n, seed, outlier_frac=0.25, noise_px=0.4, fx=525.0):
    """n matches of a camera that moved by a small twist: obs rows = ax, ay, px, py, pz, level"""
    rng = np.random.default_rng(seed)
    true_pose = orc.se3_exp(np.array([0.03, -0.02, 0.01, 0.004, -0.006, 0.003]) * (1 + seed % 3))
    guess = orc.se3_exp(np.zeros(6))
    P = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-0.9, 0.9, n), rng.uniform(1.5, 3.0, n)], 1)
    from oraclelib import quat_rot
    R = quat_rot(true_pose[:4])
    pc = P @ R.T + true_pose[4:]
    a = pc[:, :2] / pc[:, 2:3] + rng.normal(0, noise_px / fx, (n, 2))
    bad = rng.random(n) < outlier_frac
    a[bad] += rng.uniform(-40, 40, (int(bad.sum()), 2)) / fx
    lvl = rng.integers(0, 3, n)
    obs = np.concatenate([a, P, lvl[:, None].astype(np.float64)], 1)
    return obs, guess


@pytest.mark.parametrize("sizes", [(150, 40, 5, 3, 1, 0), (256, 255, 64, 65, 129, 7), (1024, 700, 257)])
def test_pose_from_matches_equals_oracle(ctx, orc, sizes):
    """RANSAC replay + Tukey Gauss-Newton + rescue on the device: same rand() consumption, same inlier / outlier lists
    in the same order; pose within 1e-9 (device sin/cos inside SE3::Exp differ from libm by an ulp)"""
    cam = TUM_CAM
    jobs, wants = [], []
    for j, n in enumerate(sizes):
        obs, guess = make_matches(orc, n, seed=100 + j, outlier_frac=0.1 + 0.1 * (j % 4))
        skip = 17 * j
        draws = orc.rand_stream(skip + 100, seed=3)[skip:]
        jobs.append((obs, guess, draws))
        wants.append(orc.pose_from_matches(cam, obs, guess, rand_seed=3, rand_skip=skip))
    got = ctx.pose_from_matches(jobs, fx=cam[0])
    for j, (g, w) in enumerate(zip(got, wants)):
        assert g["n_draws"] == w["n_draws"], j
        assert np.array_equal(g["inliers"], w["inliers"]), j
        assert np.array_equal(g["outliers"], w["outliers"]), j
        assert np.abs(g["pose"] - w["pose"]).max() <= 1e-9, (j, g["pose"], w["pose"])
    # the big jobs really exercise RANSAC: outliers were rejected and the true motion recovered
    assert len(got[0]["outliers"]) >= 10 and len(got[0]["inliers"]) >= 90


def test_pose_from_matches_many_draws(ctx, orc):
    """max_ransac_its above one workgroup's 128 draws and not a multiple of it (300 = 128 + 128 + 44), with enough gross
    outliers that the adaptive budget keeps drawing"""
    cam = TUM_CAM
    old = orc.params.max_ransac_its
    orc.params.max_ransac_its = 300
    try:
        jobs, wants = [], []
        for j, (n, frac) in enumerate([(180, 0.55), (90, 0.7), (40, 0.3)]):
            obs, guess = make_matches(orc, n, seed=200 + j, outlier_frac=frac)
            draws = orc.rand_stream(300, seed=5)
            jobs.append((obs, guess, draws))
            wants.append(orc.pose_from_matches(cam, obs, guess, rand_seed=5))
        got = ctx.pose_from_matches(jobs, fx=cam[0], max_ransac_its=300)
    finally:
        orc.params.max_ransac_its = old
    for j, (g, w) in enumerate(zip(got, wants)):
        assert g["n_draws"] == w["n_draws"], j
        assert np.array_equal(g["inliers"], w["inliers"]) and np.array_equal(g["outliers"], w["outliers"]), j
        assert np.abs(g["pose"] - w["pose"]).max() <= 1e-9, j
    assert max(w["n_draws"] for w in wants) > 128   # the later workgroups' draws are really consumed


@pytest.mark.parametrize("env", [None, "SDVL_POSE_ALL_SUPPORTERS"])
def test_pose_from_matches_batch_of_40_on_the_farm_forms(ctx, orc, env, monkeypatch):
    """More than 32 jobs in one call: the farm's forms of the pose stage — one LANE per draw, the first 64 draws converged by
    pose_hypotheses, the supporters counted by pose_refine as its replay reaches a draw, draws beyond 64 converged there on demand
    (round 6).  Jobs with few inliers keep drawing past 64; the oracle's n_draws, lists and poses are demanded of every job.
    SDVL_POSE_ALL_SUPPORTERS=1: rounds 3-5's form (every draw against every match in pose_supporters)."""
    if env:
        monkeypatch.setenv(env, "1")
    cam = TUM_CAM
    jobs, wants = [], []
    for j in range(40):
        n = [160, 60, 200, 25][j % 4]
        frac = [0.1, 0.5, 0.62, 0.3][j % 4] if j % 8 < 6 else 0.7
        obs, guess = make_matches(orc, n, seed=300 + j, outlier_frac=frac)
        skip = 11 * j
        draws = orc.rand_stream(skip + 100, seed=7)[skip:]
        jobs.append((obs, guess, draws))
        wants.append(orc.pose_from_matches(cam, obs, guess, rand_seed=7, rand_skip=skip))
    got = ctx.pose_from_matches(jobs, fx=cam[0])
    for j, (g, w) in enumerate(zip(got, wants)):
        assert g["n_draws"] == w["n_draws"], j
        assert np.array_equal(g["inliers"], w["inliers"]) and np.array_equal(g["outliers"], w["outliers"]), j
        assert np.abs(g["pose"] - w["pose"]).max() <= 1e-9, j
    n_draws = [w["n_draws"] for w in wants]
    assert max(n_draws) > 64 and min(n_draws) < 20, n_draws   # both the on-demand draws and the early stop are exercised


def test_pose_from_matches_degenerate_inputs(ctx, orc):
    """all matches identical / all outliers / more matches than the device path takes"""
    cam = TUM_CAM
    obs = np.tile(np.array([[0.1, -0.05, 0.2, -0.1, 2.0, 0.0]]), (12, 1))
    obs2, guess = make_matches(orc, 30, seed=5, outlier_frac=1.0)
    draws = orc.rand_stream(100, seed=1)
    got = ctx.pose_from_matches([(obs, guess, draws), (obs2, guess, draws)], fx=cam[0])
    for g, o in zip(got, (obs, obs2)):
        w = orc.pose_from_matches(cam, o, guess, rand_seed=1)
        assert g["n_draws"] == w["n_draws"]
        assert np.array_equal(g["inliers"], w["inliers"]) and np.array_equal(g["outliers"], w["outliers"])
        assert np.abs(g["pose"] - w["pose"]).max() <= 1e-9
    big, _ = make_matches(orc, 1025, seed=9)
    with pytest.raises(RuntimeError, match="too many matches"):
        ctx.pose_from_matches([(big, guess, draws)], fx=cam[0])


# ------------------------------------------------------------------------------------------------ K0 (input stage)
@pytest.mark.parametrize("shape,cam,dist", [((480, 640), "tum", "tum"), ((480, 752), "euroc", "euroc"), ((66, 130), "tum", "strong")])
def test_undistort_bit_exact(ctx, sdvl, orc, synth, shape, cam, dist):
    """Camera::UndistortImage = cv::undistort (camera.cc:100-105): the remap kernel reproduces the oracle byte for byte —
    stripe-wise principal point, accumulated column coordinates, fixed-point map and weights, constant border"""
    from oraclelib import TUM_DIST, EUROC_DIST
    h, w = shape
    cam4 = TUM_CAM if cam == "tum" else EUROC_CAM
    if shape == (66, 130):
        cam4 = np.array([110.0, 108.0, 63.7, 31.2])
    d = {"tum": TUM_DIST, "euroc": EUROC_DIST, "strong": np.array([-0.45, 0.3, 0.01, -0.008, 0.05])}[dist]
    imgs = [rand_img(77, h, w)]
    if shape == (480, 640):
        imgs.append(frames_of(synth, orc, TUM_CAM, 640, 480, [2])[0])
    c = sdvl.Camera(w, h, *cam4)
    got = ctx.undistort(imgs, c, d)
    for g, im in zip(got, imgs):
        want = orc.undistort(im, cam4, d)
        assert np.array_equal(g, want), int((g != want).sum())
        assert (want != im).mean() > 0.5          # the distortion really moves pixels
    # fused form: raw image -> undistorted level 0 -> pyramid
    levels = 5 if min(shape) >= 256 else 3
    fr = [ctx.frame(width=w, height=h, levels=levels, pyramid=False) for _ in imgs]
    ctx.undistort(imgs, c, d, frames=fr)
    ctx.pyramid_build(fr)
    for f, im in zip(fr, imgs):
        want = orc.pyramid(orc.undistort(im, cam4, d), levels)
        for l in range(levels):
            assert np.array_equal(f.level(l), want[l]), l
        f.close()
    # d0 == 0: the reference clones the image (camera.cc:46)
    same = ctx.undistort(imgs[:1], c, [0.0, 0.4, 0.0, 0.0, 0.0])[0]
    assert np.array_equal(same, imgs[0])


def test_host_layer_camera_undistort(orc):
    """sdvl::Camera::SetDistortions / UndistortImage of the host layer (camera.cc:39-67,100-105)"""
    import ctypes as C
    import importlib
    from oraclelib import TUM_DIST
    importlib.import_module("slam-sdvl_amd")
    trk = importlib.import_module("slam-sdvl_amd.tracker")
    dev = trk.HostDevice(0)
    img = rand_img(5, 480, 640)
    out = np.zeros_like(img)
    cam = np.ascontiguousarray(TUM_CAM, np.float64); dist = np.ascontiguousarray(TUM_DIST, np.float64)
    f = dev.lib.sdvlh_camera_undistort
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    assert f(dev.h, 640, 480, cam.ctypes.data, dist.ctypes.data, img.ctypes.data, 640, out.ctypes.data) == 1
    assert np.array_equal(out, orc.undistort(img, TUM_CAM, TUM_DIST))
    dev.close()


@pytest.mark.parametrize("dense_num", ["0", "64"])
def test_fast_cells_with_every_cell_on_one_path(dense_num):
    """SDVL_FAST_DENSE_NUM (of 64 probed pixels that send a cell down the dense path; default 16): 0 = every cell dense, 64 = every cell
    through the candidate list.  Both yield the default's corners, order included — the switch is read once per process, so each value runs
    the FAST / detection parity tests in a process of its own"""
    import os
    import subprocess
    import sys
    if os.environ.get("SDVL_FAST_DENSE_NUM"):
        pytest.skip("already inside a run with the switch set")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k",
                        "(fast_cells or detect_corners) and not every_cell", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, SDVL_FAST_DENSE_NUM=dense_num))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]
