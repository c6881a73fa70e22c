"""ctypes bindings for the CPU oracle (oracle/libsdvl_oracle.so) and the synthetic sequence generator.

Test infrastructure only: nothing under slam-sdvl_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
CSRC_DIR = os.path.join(ROOT, "slam-sdvl_amd", "csrc")

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "pyramid_levels", "cell_size", "max_fast_levels", "fast_threshold", "num_features", "use_orb", "orb_size",
        "patch_size", "max_align_its", "search_size", "align_patch_size", "max_align_level", "min_align_level",
        "max_img_align_its", "min_feature_score", "max_matches", "min_matches", "max_failed", "max_optim_pose_its",
        "max_ransac_points", "max_ransac_its", "min_keyframe_its")] + [
        ("inlier_error_threshold", C.c_double), ("lost_ratio", C.c_double)]


class FrameStats(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("state", "quality", "matches", "attempts", "inliers", "outliers", "n_corners",
                                       "align_meas", "keyframe", "relocalized")] + [("pose", C.c_double * 7)]


class SynthView(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("u0", C.c_double), ("v0", C.c_double),
                ("R", C.c_double * 9), ("t", C.c_double * 3), ("plane", C.c_double * 4),
                ("seed", C.c_uint32), ("frame_id", C.c_uint32), ("texture", C.c_uint32), ("reserved_", C.c_uint32),
                ("dist", C.c_double * 5)]


def _ensure(path, cmd, cwd):
    if not os.path.exists(path):
        subprocess.check_call(cmd, cwd=cwd)
    return path


def load_oracle(ref_flags=False):
    """the checker (libsdvl_oracle.so: -ffp-contract=off, every parity test compares with it) or, for TIMING ONLY, the same sources
    built here and now with the reference's own flags (CMakeLists.txt:20: -O3 -march=native), libsdvl_oracle_refflags.so"""
    if ref_flags:
        so = os.path.join(ORACLE_DIR, "libsdvl_oracle_refflags.so")
        subprocess.check_call(["make", "-s", "-B", "libsdvl_oracle_refflags.so"], cwd=ORACLE_DIR)   # -march=native: always for THIS host
    else:
        so = _ensure(os.path.join(ORACLE_DIR, "libsdvl_oracle.so"), ["make", "-s"], ORACLE_DIR)
    lib = C.CDLL(so)
    lib.sdvl_ref_shi_tomasi.restype = C.c_double
    return lib


def load_synth():
    so = os.path.join(CSRC_DIR, "libsdvl_synth.so")
    _ensure(so, ["g++", "-O2", "-ffp-contract=off", "-march=x86-64-v3", "-fPIC", "-shared", "-o", so,
                 os.path.join(CSRC_DIR, "synth_host.cc")], CSRC_DIR)
    return C.CDLL(so)


def ptr(a, t):
    return a.ctypes.data_as(t)


TUM_DIST = np.array([0.2624, -0.9531, -0.0054, 0.0026, 1.1633])   # config/config_tum_f1.cfg:15-19
EUROC_DIST = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0])  # config/config_euroc.cfg:15-19
TUM_CAM = np.array([517.3, 516.5, 318.6, 255.3])          # config/config_tum_f1.cfg:11-14
TUM2_CAM = np.array([520.9, 521.0, 325.1, 249.7])         # config/config_tum_f2.cfg:11-14
TUM2_DIST = np.array([0.2312, -0.7849, -0.0033, -0.0001, 0.9172])  # config/config_tum_f2.cfg:15-19
EUROC_CAM = np.array([458.654, 457.296, 367.215, 248.375])  # config/config_euroc.cfg
XI = np.array([0.004, 0.002, 0.001, 0.0008, -0.0012, 0.0005])  # SURVEY §8d trajectory twist per frame


class Oracle:
    def __init__(self, ref_flags=False):
        self.lib = load_oracle(ref_flags)
        self.params = Params()
        self.lib.sdvl_ref_default_params(C.byref(self.params))

    # --- math
    def se3_exp(self, u):
        u = np.ascontiguousarray(u, np.float64); T = np.zeros(7)
        self.lib.sdvl_ref_se3_exp(ptr(u, f64p), ptr(T, f64p)); return T

    def se3_log(self, T):
        T = np.ascontiguousarray(T, np.float64); u = np.zeros(6)
        self.lib.sdvl_ref_se3_log(ptr(T, f64p), ptr(u, f64p)); return u

    def se3_mul(self, A, B):
        A = np.ascontiguousarray(A, np.float64); B = np.ascontiguousarray(B, np.float64); Cc = np.zeros(7)
        self.lib.sdvl_ref_se3_mul(ptr(A, f64p), ptr(B, f64p), ptr(Cc, f64p)); return Cc

    def se3_inv(self, A):
        A = np.ascontiguousarray(A, np.float64); B = np.zeros(7)
        self.lib.sdvl_ref_se3_inv(ptr(A, f64p), ptr(B, f64p)); return B

    def ldlt_solve6(self, A, b):
        A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64); x = np.zeros(6)
        self.lib.sdvl_ref_ldlt_solve6(ptr(A, f64p), ptr(b, f64p), ptr(x, f64p)); return x

    def rand_stream(self, n, seed=1):
        out = np.zeros(n, np.int32)
        self.lib.sdvl_ref_rand_stream(C.c_uint(seed), n, ptr(out, i32p)); return out

    # --- detection
    def pyr_down(self, img):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        out = np.zeros((h // 2, w // 2), np.uint8)
        self.lib.sdvl_ref_pyr_down(ptr(img, u8p), w, h, w, ptr(out, u8p), w // 2); return out

    def pyramid(self, img, levels=5):
        pyr = [np.ascontiguousarray(img, np.uint8)]
        for _ in range(1, levels):
            pyr.append(self.pyr_down(pyr[-1]))
        return pyr

    def fast(self, img, thr=10, nonmax=True, cap=200000):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        out = np.zeros((cap, 3), np.int32)
        n = self.lib.sdvl_ref_fast(ptr(img, u8p), w, h, w, thr, int(nonmax), cap, ptr(out, i32p))
        assert n <= cap
        return out[:n].copy()

    def fast_cells(self, img, cap=200000):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        cs = self.params.cell_size
        ncells = -(-w // cs) * -(-h // cs)
        out = np.zeros((cap, 3), np.int32); offs = np.zeros(ncells + 1, np.int32); ran = np.zeros(ncells, np.uint8)
        n = self.lib.sdvl_ref_fast_cells(ptr(img, u8p), w, h, w, C.byref(self.params), cap, ptr(out, i32p),
                                         ptr(offs, i32p), ptr(ran, u8p))
        assert n <= cap
        return out[:n].copy(), offs, ran

    def detect_pyramid(self, img, nfeatures=None, cap=20000):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        out = np.zeros((cap, 3), np.int32)
        nf = self.params.num_features if nfeatures is None else nfeatures
        n = self.lib.sdvl_ref_detect_pyramid(ptr(img, u8p), w, h, w, C.byref(self.params), nf, cap, ptr(out, i32p))
        assert n <= cap
        return out[:n].copy()

    def retain_best(self, packed, n_points):
        packed = np.ascontiguousarray(packed, np.uint32).copy()
        n = self.lib.sdvl_ref_retain_best(ptr(packed, C.POINTER(C.c_uint32)), len(packed), int(n_points))
        return packed[:n]

    def shi_tomasi(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        return self.lib.sdvl_ref_shi_tomasi(ptr(img, u8p), w, h, w, int(x), int(y))

    def filter_corners(self, img, corners, locked=None):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        corners = np.ascontiguousarray(corners, np.int32)
        locked = np.zeros((0, 2)) if locked is None else np.ascontiguousarray(locked, np.float64)
        out = np.zeros(4096, np.int32)
        n = self.lib.sdvl_ref_filter_corners(ptr(img, u8p), w, h, w, C.byref(self.params), len(corners),
                                             ptr(corners, i32p), len(locked), ptr(locked, f64p), 4096, ptr(out, i32p))
        return out[:n].copy()

    def orb_describe(self, img, xy):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        xy = np.ascontiguousarray(xy, np.int32); n = len(xy)
        desc = np.zeros((n, 32), np.uint8); ang = np.zeros(n, np.float32)
        self.lib.sdvl_ref_orb_describe(ptr(img, u8p), w, h, w, n, ptr(xy, i32p), ptr(desc, u8p), ptr(ang, f32p))
        return desc, ang

    def orb_distance(self, a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        return self.lib.sdvl_ref_orb_distance(ptr(a, u8p), ptr(b, u8p))

    def pose_from_matches(self, cam, obs, pose, w=640, h=480, rand_seed=1, rand_skip=0):
        """SelectInliers + OptimizePose on obs[n][6] = ax, ay, px, py, pz, level"""
        cam = np.ascontiguousarray(cam, np.float64); obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 6)
        n = len(obs); pose = np.array(pose, np.float64)
        nd = C.c_int(); ni = C.c_int(); no = C.c_int()
        ii = np.zeros(max(n, 1), np.int32); oi = np.zeros(max(n, 1), np.int32)
        self.lib.sdvl_ref_pose_from_matches(C.byref(self.params), w, h, ptr(cam, f64p), n, ptr(obs, f64p), C.c_uint(rand_seed),
                                            int(rand_skip), ptr(pose, f64p), C.byref(nd), C.byref(ni), ptr(ii, i32p),
                                            C.byref(no), ptr(oi, i32p))
        return dict(pose=pose, n_draws=nd.value, inliers=ii[:ni.value].copy(), outliers=oi[:no.value].copy())

    def undistort(self, img, cam, dist):
        """Camera::UndistortImage = cv::undistort; cam = fx fy u0 v0, dist = d0..d4"""
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        cam = np.ascontiguousarray(cam, np.float64); dist = np.ascontiguousarray(dist, np.float64)
        out = np.zeros((h, w), np.uint8)
        self.lib.sdvl_ref_undistort(ptr(img, u8p), w, h, w, ptr(cam, f64p), ptr(dist, f64p), ptr(out, u8p))
        return out

    def remap_weights(self):
        out = np.zeros(4096, np.int16)
        self.lib.sdvl_ref_remap_weights(out.ctypes.data_as(C.POINTER(C.c_int16)))
        return out.reshape(32, 32, 4)

    # --- alignment
    def image_align(self, img1, img2, cam, px, bearing, depth, valid, T, fast=False):
        img1 = np.ascontiguousarray(img1, np.uint8); img2 = np.ascontiguousarray(img2, np.uint8); h, w = img1.shape
        cam = np.ascontiguousarray(cam, np.float64); px = np.ascontiguousarray(px, np.float64)
        bearing = np.ascontiguousarray(bearing, np.float64); depth = np.ascontiguousarray(depth, np.float64)
        valid = np.ascontiguousarray(valid, np.uint8); T = np.array(T, np.float64)
        err = C.c_double(); chi2 = C.c_double(); its = np.zeros(8, np.int32)
        n = self.lib.sdvl_ref_image_align(ptr(img1, u8p), ptr(img2, u8p), w, h, C.byref(self.params), ptr(cam, f64p),
                                          len(px), ptr(px, f64p), ptr(bearing, f64p), ptr(depth, f64p), ptr(valid, u8p),
                                          ptr(T, f64p), int(fast), C.byref(err), C.byref(chi2), ptr(its, i32p))
        return dict(T=T, n=n, error=err.value, chi2=chi2.value, its=its)

    def search_point(self, ref_img, cur_img, cam, ref_pose, cur_pose, feat_px, feat_bearing, feat_level, feat_desc,
                     idepth, idepth_std, fixed, corners, px0):
        ref_img = np.ascontiguousarray(ref_img, np.uint8); cur_img = np.ascontiguousarray(cur_img, np.uint8)
        h, w = ref_img.shape
        cam = np.ascontiguousarray(cam, np.float64)
        ref_pose = np.ascontiguousarray(ref_pose, np.float64); cur_pose = np.ascontiguousarray(cur_pose, np.float64)
        feat_px = np.ascontiguousarray(feat_px, np.float64); feat_bearing = np.ascontiguousarray(feat_bearing, np.float64)
        feat_desc = np.ascontiguousarray(feat_desc, np.uint8); corners = np.ascontiguousarray(corners, np.int32)
        px = np.array(px0, np.float64); lvl = C.c_int(-1); border = np.zeros(100, np.uint8); slevel = C.c_int(-1)
        found = self.lib.sdvl_ref_search_point(
            ptr(ref_img, u8p), ptr(cur_img, u8p), w, h, C.byref(self.params), ptr(cam, f64p), ptr(ref_pose, f64p),
            ptr(cur_pose, f64p), ptr(feat_px, f64p), ptr(feat_bearing, f64p), int(feat_level), ptr(feat_desc, u8p),
            C.c_double(idepth), C.c_double(idepth_std), int(fixed), len(corners), ptr(corners, i32p), ptr(px, f64p),
            C.byref(lvl), ptr(border, u8p), C.byref(slevel))
        return dict(found=found, px=px, level=lvl.value, border=border, slevel=slevel.value)

    def align_patch(self, img, border, patch, px0, max_its=10):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        border = np.ascontiguousarray(border, np.uint8); patch = np.ascontiguousarray(patch, np.uint8)
        px = np.array(px0, np.float64)
        ok = self.lib.sdvl_ref_align_patch(ptr(img, u8p), w, h, w, ptr(border, u8p), ptr(patch, u8p), ptr(px, f64p), max_its)
        return ok, px

    # --- tracker
    def tracker(self, w, h, cam, plane=(0, 0, 1, 2.0), first_pose=(1, 0, 0, 0, 0, 0, 0)):
        return OracleTracker(self, w, h, cam, plane, first_pose)


class OracleTracker:
    def __init__(self, orc, w, h, cam, plane, first_pose):
        self.lib = orc.lib
        self.w, self.h = w, h
        cam = np.ascontiguousarray(cam, np.float64); plane = np.ascontiguousarray(plane, np.float64)
        fp = np.ascontiguousarray(first_pose, np.float64)
        self.lib.sdvl_ref_tracker_create.restype = C.c_void_p
        self.h_ = self.lib.sdvl_ref_tracker_create(C.byref(orc.params), w, h, ptr(cam, f64p), ptr(plane, f64p), ptr(fp, f64p))
        self.set_max_keyframes(1000)   # SDVL.max_keyframes of the reference's cfg files, what tracker.configure() sets on the product side

    def handle_frame(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        st = FrameStats()
        self.lib.sdvl_ref_tracker_handle_frame(C.c_void_p(self.h_), ptr(img, u8p), self.w, C.byref(st))
        return st

    def set_max_keyframes(self, n):
        """SDVL.max_keyframes (the plane-map stub honours it like Map::LimitKeyframes, map.cc:190-205)"""
        self.lib.sdvl_ref_tracker_set_max_keyframes(C.c_void_p(self.h_), int(n))

    def use_mapper(self, on=True, max_search_keyframes=5, max_keyframes=100, map_scale=1.0, scale_min_dist=0.25):
        """the reference's mapper in sequential mode (map.cc) instead of the plane map stub"""
        self.lib.sdvl_ref_tracker_use_mapper(C.c_void_p(self.h_), int(on), int(max_search_keyframes), int(max_keyframes),
                                             C.c_double(map_scale), C.c_double(scale_min_dist))

    def depth_filter(self, cur_pose, ref_pose, bearing, found, px, depth_mean, state12):
        """the body of Map::UpdateCandidates' loop behind SearchPoint (map.cc:454-497) for one candidate -> (outcome, new state)"""
        st = np.array(state12, np.float64)
        cp, rp, bv, pp = (np.ascontiguousarray(a, np.float64) for a in (cur_pose, ref_pose, bearing, px))
        self.lib.sdvl_ref_depth_filter.restype = C.c_int
        self.lib.sdvl_ref_depth_filter.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_double, C.c_void_p]
        rc = self.lib.sdvl_ref_depth_filter(C.c_void_p(self.h_), cp.ctypes.data, rp.ctypes.data, bv.ctypes.data, int(found), pp.ctypes.data,
                                            float(depth_mean), st.ctypes.data)
        return rc, st

    def map_stats(self):
        out = np.zeros(6, np.int32)
        self.lib.sdvl_ref_tracker_map_stats(C.c_void_p(self.h_), ptr(out, i32p))
        return dict(zip(("candidates", "converged", "initialized", "linked", "connected", "keyframes"), out.tolist()))

    def mapper_points(self, cap=20000):
        """x y z converged of the live points created by the mapper (bootstrap points excluded)"""
        out = np.zeros((cap, 4), np.float64)
        n = self.lib.sdvl_ref_tracker_mapper_points(C.c_void_p(self.h_), cap, ptr(out, f64p))
        return out[:n].copy()

    def close(self):
        if self.h_:
            self.lib.sdvl_ref_tracker_destroy(C.c_void_p(self.h_)); self.h_ = None

    def __del__(self):
        self.close()


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class Synth:
    """Host renderer of the synthetic plane scene (slam-sdvl_amd/csrc/sdvl_synth.h)."""

    def __init__(self):
        self.lib = load_synth()

    def render(self, T_cw, cam, w, h, seed=20260001, frame_id=0, plane=(0, 0, 1, 2.0), texture=0, dist=None):
        """dist = (k1, k2, p1, p2, k3): the view through that lens (what Camera::UndistortImage undoes); None: pinhole"""
        v = SynthView()
        for i in range(5):
            v.dist[i] = float(dist[i]) if dist is not None else 0.0
        v.fx, v.fy, v.u0, v.v0 = [float(c) for c in cam]
        R = quat_to_R(T_cw[:4])
        for i in range(9):
            v.R[i] = float(R.flat[i])
        for i in range(3):
            v.t[i] = float(T_cw[4 + i])
        for i in range(4):
            v.plane[i] = float(plane[i])
        v.seed = seed; v.frame_id = frame_id; v.texture = texture
        out = np.zeros((h, w), np.uint8)
        self.lib.sdvl_synth_render_host(C.byref(v), w, h, ptr(out, u8p), w)
        return out


def quat_rot(q):
    """rotation matrix of a unit quaternion (w, x, y, z)"""
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def trajectory_pose(orc, k, xi=XI):
    """T_k = Exp(k * xi) as a world->camera pose (7 doubles)."""
    return orc.se3_exp(np.asarray(xi) * k)
