"""slam-sdvl_amd — MI355X-native SDVL tracking front-end (import with importlib.import_module("slam-sdvl_amd")).

Thin ctypes view of the C-ABI in include/sdvl_hip.h (libsdvl_hip.so, hand-written gfx950 kernels).  There is NO
CPU fallback: importing works anywhere the library loads (so that symbol checks run without a GPU), but every
compute entry point needs an MI355X and raises SdvlError otherwise.  Nothing here touches oracle/.
"""
import ctypes as C
import os

import numpy as np

# PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  Two HIP runtimes in one process
# fight over the GPU (whichever initialises second sees no device), so when torch is installed it is imported
# FIRST: libsdvl_hip.so then binds to the already-loaded runtime by soname and shares devices and streams with it.
try:
    import torch  # noqa: F401  (plumbing only: runtime sharing, torch.distributed in bench.py)
except ImportError:  # pure C-ABI use without torch: the system ROCm runtime is loaded instead
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsdvl_hip.so")
HOST_LIB_PATH = os.path.join(_HERE, "host", "libsdvl_host.so")

MAX_LEVELS = 8
MAX_CORNERS = 6144

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)


class SdvlError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("width", "height", "fx", "fy", "u0", "v0")]


class DetectParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("cell_size", "max_fast_levels", "fast_threshold", "margin")]


class Keypoint(C.Structure):
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("score", C.c_uint8), ("level", C.c_uint8), ("cell", C.c_uint16)]


class AlignFeature(C.Structure):
    _fields_ = [("px", C.c_double), ("py", C.c_double), ("fx", C.c_double), ("fy", C.c_double), ("fz", C.c_double),
                ("depth", C.c_double), ("valid", C.c_int32), ("pad_", C.c_int32)]


class AlignJob(C.Structure):
    _fields_ = [("ref", C.c_void_p), ("cur", C.c_void_p), ("feat_begin", C.c_int32), ("feat_end", C.c_int32),
                ("T", C.c_double * 7)]


class AlignParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("max_level", "min_level", "max_its", "patch_size", "fast")]


class AlignResult(C.Structure):
    _fields_ = [("T", C.c_double * 7), ("error", C.c_double), ("chi2", C.c_double), ("n_meas", C.c_int32),
                ("its", C.c_int32 * MAX_LEVELS), ("stop", C.c_int32), ("iters_run", C.c_int32), ("pad_", C.c_int32)]


class SearchParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("patch_size", "max_align_its", "search_size", "max_fast_levels", "margin", "use_orb", "lk_tree_sums", "pad_")]


class SearchReq(C.Structure):
    _fields_ = [("cur", C.c_void_p), ("ref", C.c_void_p), ("cur_pose", C.c_double * 7), ("ref_pose", C.c_double * 7),
                ("px", C.c_double * 2), ("bearing", C.c_double * 3), ("idepth", C.c_double), ("idepth_std", C.c_double),
                ("px0", C.c_double * 2), ("level", C.c_int32), ("fixed", C.c_int32), ("desc", C.c_uint8 * 32)]


class SearchReqPacked(C.Structure):
    """sdvl_search_req_packed: a request of a sdvl_search_begin batch (frames named by slot)"""
    _fields_ = [("cur", C.c_int32), ("ref", C.c_int32), ("level", C.c_int32), ("fixed", C.c_int32), ("px", C.c_double * 2),
                ("bearing", C.c_double * 3), ("idepth", C.c_double), ("idepth_std", C.c_double), ("px0", C.c_double * 2),
                ("desc", C.c_uint8 * 32)]


class ChainFrame(C.Structure):
    _fields_ = [("cand_begin", C.c_int32), ("cand_end", C.c_int32), ("max_matches", C.c_int32), ("rand_begin", C.c_int32),
                ("pose", C.c_double * 7)]


class SearchRes(C.Structure):
    _fields_ = [("px", C.c_double * 2), ("found", C.c_int32), ("level", C.c_int32), ("best_corner", C.c_int32),
                ("stage", C.c_int32), ("lk_its", C.c_int32), ("slevel", C.c_int32)]


class DepthState(C.Structure):
    """sdvl_depth_state: the depth-filter state of the Point behind a candidate request (point.h:136-150)"""
    _fields_ = [("rho", C.c_double), ("sigma2", C.c_double), ("a", C.c_double), ("b", C.c_double), ("z_range", C.c_double),
                ("cos_alpha", C.c_double), ("last_distance", C.c_double), ("depth_mean", C.c_double), ("position", C.c_double * 3),
                ("fixed", C.c_int32), ("n_failed", C.c_int32), ("track_row", C.c_int32), ("pad_", C.c_int32)]


class DepthParams(C.Structure):
    _fields_ = [("px_error_angle", C.c_double), ("min_depth", C.c_double), ("scale_min_dist", C.c_double),
                ("max_failed", C.c_int32), ("pad_", C.c_int32)]


class DepthOut(C.Structure):
    _fields_ = [("outcome", C.c_int32), ("n_failed", C.c_int32), ("rho", C.c_double), ("sigma2", C.c_double), ("a", C.c_double),
                ("b", C.c_double), ("cos_alpha", C.c_double), ("last_distance", C.c_double), ("position", C.c_double * 3)]


class SynthView(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("u0", C.c_double), ("v0", C.c_double),
                ("R", C.c_double * 9), ("t", C.c_double * 3), ("plane", C.c_double * 4),
                ("seed", C.c_uint32), ("frame_id", C.c_uint32), ("texture", C.c_uint32), ("reserved_", C.c_uint32),
                ("dist", C.c_double * 5)]


# every symbol include/sdvl_hip.h declares
ABI_SYMBOLS = [
    "sdvl_ctx_create", "sdvl_ctx_destroy", "sdvl_last_error", "sdvl_ctx_synchronize", "sdvl_ctx_stream",
    "sdvl_ctx_bind_thread", "sdvl_ctx_device", "sdvl_pointer_device", "sdvl_ctx_scratch_device",
    "sdvl_ctx_timing_enable", "sdvl_ctx_timing_only", "sdvl_ctx_timing_get", "sdvl_ctx_timing_reset",
    "sdvl_frame_create", "sdvl_frame_create_many", "sdvl_frame_destroy", "sdvl_frame_upload", "sdvl_frames_upload", "sdvl_ctx_prefetch_images", "sdvl_ctx_prefetch_fence", "sdvl_frame_set_image_device", "sdvl_frame_borrow_image_device",
    "sdvl_pyramid_build", "sdvl_frame_download_level", "sdvl_fast_num_cells", "sdvl_fast_cells",
    "sdvl_detect_corners", "sdvl_frames_corner_counts", "sdvl_frame_download_corners", "sdvl_retain_best",
    "sdvl_frame_set_corners", "sdvl_frames_set_corners", "sdvl_frame_num_corners", "sdvl_shi_tomasi", "sdvl_orb_describe",
    "sdvl_frame_download_descriptors", "sdvl_filter_inputs", "sdvl_filter_inputs_begin", "sdvl_filter_inputs_end", "sdvl_filter_corners_begin", "sdvl_filter_corners_end", "sdvl_orb_describe_points", "sdvl_hamming_argmin", "sdvl_image_align", "sdvl_image_align_begin", "sdvl_image_align_end", "sdvl_align_store_create", "sdvl_align_store_destroy", "sdvl_align_store_write", "sdvl_image_align_begin_stored", "sdvl_search_points", "sdvl_search_begin", "sdvl_search_slot", "sdvl_search_run", "sdvl_align_patches", "sdvl_pose_from_matches", "sdvl_search_points_filter", "sdvl_search_run_filter", "sdvl_search_run_chain", "sdvl_search_chain_end", "sdvl_frame_footprint", "sdvl_undistort", "sdvl_frames_upload_undistorted", "sdvl_ctx_set_wait_hook", "sdvl_ctx_wait_done", "sdvl_ctx_wait_block", "sdvl_ctx_health", "sdvl_ctx_counters",
    "sdvl_track_create", "sdvl_track_destroy", "sdvl_frame_register", "sdvl_frames_register", "sdvl_track_upload", "sdvl_track_append", "sdvl_track_align", "sdvl_track_search",
    "sdvl_track_collect", "sdvl_track_features", "sdvl_track_stats",
    "sdvl_synth_render", "sdvl_device_malloc", "sdvl_device_free", "sdvl_device_download",
    "sdvl_frames_own_images", "sdvl_feed_create", "sdvl_feed_destroy", "sdvl_feed_last_error", "sdvl_feed_slot_arrived", "sdvl_feed_images", "sdvl_ctx_feed_acquire", "sdvl_ctx_feed_release", "sdvl_frame_footprint_cap", "sdvl_ctx_set_corner_capacity", "sdvl_frame_corner_capacity", "sdvl_detect_scratch_bytes",
    "sdvl_ctx_set_wait_spin", "sdvl_ctx_fork_mark", "sdvl_ctx_fork_begin", "sdvl_ctx_fork_end", "sdvl_host_alloc_pinned", "sdvl_host_free_pinned", "sdvl_host_register", "sdvl_host_unregister",
]

_lib = None


def load_library():
    """dlopen libsdvl_hip.so; raises SdvlError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdvlError("libsdvl_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                            % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.sdvl_last_error.restype = C.c_char_p
        lib.sdvl_last_error.argtypes = [C.c_void_p]
        lib.sdvl_ctx_stream.restype = C.c_void_p
        lib.sdvl_ctx_stream.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Frame:
    """Device-resident Frame (pyramid + corners + descriptors) behind an sdvl_frame handle."""

    def __init__(self, ctx, width, height, levels=5):
        self.ctx = ctx
        self.width, self.height, self.levels = width, height, levels
        h = C.c_void_p()
        ctx._check(ctx.lib.sdvl_frame_create(ctx.h, width, height, levels, C.byref(h)))
        self.h = h

    def upload(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.height, self.width)
        self._keep = img
        self.ctx._check(self.ctx.lib.sdvl_frame_upload(self.ctx.h, self.h, _ptr(img, u8p), self.width))
        return self

    def set_image_device(self, dev_ptr, stride=None):
        self.ctx._check(self.ctx.lib.sdvl_frame_set_image_device(self.ctx.h, self.h, C.c_void_p(dev_ptr), stride or self.width))
        return self

    def borrow_image_device(self, dev_ptr):
        """level 0 aliases the caller's HBM image (dense rows) until the next upload / own_images"""
        self.ctx._check(self.ctx.lib.sdvl_frame_borrow_image_device(self.ctx.h, self.h, C.c_void_p(dev_ptr)))
        return self

    def corner_capacity(self):
        return self.ctx.lib.sdvl_frame_corner_capacity(self.h)

    def level(self, l):
        w, h = self.width >> l, self.height >> l
        out = np.zeros((h, w), np.uint8)
        self.ctx._check(self.ctx.lib.sdvl_frame_download_level(self.ctx.h, self.h, l, _ptr(out, u8p), w))
        return out

    def set_corners(self, xyl):
        xyl = np.ascontiguousarray(xyl, np.int32).reshape(-1, 3)
        self.ctx._check(self.ctx.lib.sdvl_frame_set_corners(self.ctx.h, self.h, len(xyl), _ptr(xyl, i32p)))
        return self

    def descriptors(self, cap=MAX_CORNERS):
        """host mirror of the frame's ORB descriptors, [n_corners][32] (computed now if they have not been)"""
        out = np.zeros((cap, 32), np.uint8)
        n = (C.c_int32 * 1)()
        hs = (C.c_void_p * 1)(self.h)
        self.ctx._check(self.ctx.lib.sdvl_frames_corner_counts(self.ctx.h, 1, hs, n))
        self.ctx._check(self.ctx.lib.sdvl_frame_download_descriptors(self.ctx.h, self.h, cap, _ptr(out, u8p)))
        return out[:n[0]]

    def close(self):
        if self.h:
            self.ctx.lib.sdvl_frame_destroy(self.ctx.h, self.h)
            self.h = None



class Distortion(C.Structure):
    _fields_ = [("d", C.c_double * 5)]


class PoseObs(C.Structure):
    _fields_ = [("ax", C.c_double), ("ay", C.c_double), ("px", C.c_double), ("py", C.c_double), ("pz", C.c_double), ("inv_cov", C.c_double)]


class PoseJob(C.Structure):
    _fields_ = [("obs_begin", C.c_int32), ("obs_end", C.c_int32), ("rand_begin", C.c_int32), ("nits_begin", C.c_int32), ("pose", C.c_double * 7)]


class PoseParams(C.Structure):
    _fields_ = [("max_ransac_points", C.c_int32), ("max_ransac_its", C.c_int32), ("max_optim_pose_its", C.c_int32), ("pad_", C.c_int32),
                ("inlier_threshold", C.c_double), ("fx", C.c_double)]


class PoseResult(C.Structure):
    _fields_ = [("pose", C.c_double * 7), ("n_draws", C.c_int32), ("n_inliers", C.c_int32), ("n_outliers", C.c_int32), ("refined", C.c_int32)]


def ransac_budget_table(size, max_points, max_its):
    """iteration budget after an improvement to s supporters, s = 0..size (feature_align.cc:199-207)"""
    import math
    npoints = min(max_points, size)
    out = []
    for s in range(size + 1):
        nits = max_its
        if size > 0:
            tmp = 1.0 - (1.0 - (float(s) / float(size)))
            for _ in range(1, npoints):
                tmp *= tmp
            if not tmp < 1e-5:
                den = math.log(1.0 - tmp) if tmp < 1.0 else -math.inf
                nits = min(max_its, int(math.log(1.0 - 0.99) / den))
        out.append(nits)
    return out

class Context:
    """One sdvl_ctx = one HIP stream on one MI355X."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.sdvl_ctx_create(device, C.byref(h))
        if rc != 0:
            raise SdvlError("sdvl_ctx_create(device=%d) failed with %d: no MI355X visible; there is no CPU fallback" % (device, rc))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise SdvlError("sdvl error %d: %s" % (rc, self.lib.sdvl_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.lib.sdvl_ctx_destroy(self.h)
            self.h = None

    def stream(self):
        return self.lib.sdvl_ctx_stream(self.h)

    def synchronize(self):
        self._check(self.lib.sdvl_ctx_synchronize(self.h))

    # ---- timing
    def timing_enable(self, on=True):
        self._check(self.lib.sdvl_ctx_timing_enable(self.h, int(on)))

    def timing_only(self, name=None):
        """time only the launches of kernel `name`; None = every launch"""
        self.lib.sdvl_ctx_timing_only.argtypes = [C.c_void_p, C.c_char_p]
        self._check(self.lib.sdvl_ctx_timing_only(self.h, name.encode() if name else None))

    def timing_reset(self):
        self._check(self.lib.sdvl_ctx_timing_reset(self.h))

    def timing_get(self):
        names = ((C.c_char * 32) * 32)()
        ms = (C.c_double * 32)()
        launches = (C.c_int64 * 32)()
        n = C.c_int()
        self._check(self.lib.sdvl_ctx_timing_get(self.h, 32, names, ms, launches, C.byref(n)))
        return {names[i].value.decode(): (ms[i], launches[i]) for i in range(n.value)}

    # ---- frames
    def frame(self, img=None, width=None, height=None, levels=5, pyramid=True):
        if img is not None:
            height, width = img.shape
        f = Frame(self, width, height, levels)
        if img is not None:
            f.upload(img)
            if pyramid:
                self.pyramid_build([f])
        return f

    def pyramid_build(self, frames):
        arr = (C.c_void_p * len(frames))(*[f.h for f in frames])
        self._check(self.lib.sdvl_pyramid_build(self.h, len(frames), arr))

    def set_corner_capacity(self, max_corners):
        """corners_ capacity of the frames created from now on (<= MAX_CORNERS)"""
        self._check(self.lib.sdvl_ctx_set_corner_capacity(self.h, int(max_corners)))

    def own_images(self, frames):
        """frames whose level 0 aliases a caller image copy it into their own storage (one launch)"""
        arr = (C.c_void_p * len(frames))(*[f.h for f in frames])
        self._check(self.lib.sdvl_frames_own_images(self.h, len(frames), arr))

    def fast_cells(self, frames, dp, cap=16384):
        """-> list of (keypoints[(x,y,score,level,cell)], cell_offsets) per frame"""
        n = len(frames)
        cpl = (C.c_int * 4)()
        tot = C.c_int()
        self._check(self.lib.sdvl_fast_num_cells(frames[0].width, frames[0].height, C.byref(dp), cpl, C.byref(tot)))
        kps = (Keypoint * (n * cap))()
        offs = np.zeros((n, tot.value + 1), np.int32)
        arr = (C.c_void_p * n)(*[f.h for f in frames])
        self._check(self.lib.sdvl_fast_cells(self.h, n, arr, C.byref(dp), cap, kps, _ptr(offs, i32p)))
        raw = np.frombuffer(kps, dtype=np.dtype([("x", "<u2"), ("y", "<u2"), ("score", "u1"), ("level", "u1"), ("cell", "<u2")]))
        out = []
        for i in range(n):
            k = raw[i * cap: i * cap + offs[i, -1]]
            out.append((np.stack([k["x"], k["y"], k["score"], k["level"], k["cell"]], 1).astype(np.int32), offs[i].copy()))
        return out, [cpl[i] for i in range(dp.max_fast_levels)]

    def detect_corners(self, frames, dp, nfeatures=1000):
        """DetectPyramid fully on device -> list of corners [(x, y, level)] per frame"""
        n = len(frames)
        arr = (C.c_void_p * n)(*[f.h for f in frames])
        self._check(self.lib.sdvl_detect_corners(self.h, n, arr, C.byref(dp), nfeatures))
        counts = np.zeros(n, np.int32)
        self._check(self.lib.sdvl_frames_corner_counts(self.h, n, arr, _ptr(counts, i32p)))
        out = []
        for f, c in zip(frames, counts):
            xyl = np.zeros((max(int(c), 1), 3), np.int32)
            k = C.c_int()
            self._check(self.lib.sdvl_frame_download_corners(self.h, f.h, len(xyl), _ptr(xyl, i32p), C.byref(k)))
            out.append(xyl[:k.value].copy())
        return out

    def retain_best(self, packed, n_points, cooperative=False):
        packed = np.ascontiguousarray(packed, np.uint32).copy()
        k = C.c_int()
        self._check(self.lib.sdvl_retain_best(self.h, _ptr(packed, C.POINTER(C.c_uint32)), len(packed), int(n_points), int(cooperative),
                                              C.byref(k)))
        return packed[:k.value]

    def shi_tomasi(self, frames, cap=MAX_CORNERS):
        n = len(frames)
        out = np.zeros((n, cap), np.float64)
        arr = (C.c_void_p * n)(*[f.h for f in frames])
        self._check(self.lib.sdvl_shi_tomasi(self.h, n, arr, cap, _ptr(out, f64p)))
        return [out[i, :self.lib.sdvl_frame_num_corners(frames[i].h)].copy() for i in range(n)]

    def orb_describe(self, frames, want=True, cap=MAX_CORNERS):
        n = len(frames)
        arr = (C.c_void_p * n)(*[f.h for f in frames])
        if not want:
            self._check(self.lib.sdvl_orb_describe(self.h, n, arr, cap, None))
            return None
        out = np.zeros((n, cap, 32), np.uint8)
        self._check(self.lib.sdvl_orb_describe(self.h, n, arr, cap, _ptr(out, u8p)))
        return [out[i, :self.lib.sdvl_frame_num_corners(frames[i].h)].copy() for i in range(n)]

    def filter_corners(self, frames, locked_px, cell_size=32, margin=19, min_feature_score=50):
        """sdvl_filter_corners_begin / _end: Frame::FilterCorners for the frames, selection on the device.  locked_px[i] = (k, 2)
        positions whose cells are locked (FastDetector::LockCell).  Returns per frame (indices, xyl, scores, descriptors)."""
        n = len(frames)
        w, h = frames[0].width, frames[0].height
        gw, gh = (w + cell_size - 1) // cell_size, (h + cell_size - 1) // cell_size
        words = (gw * gh + 31) // 32
        mask = np.zeros((n, words), np.uint32)
        for i, pts in enumerate(locked_px):
            for x, y in np.asarray(pts, np.float64).reshape(-1, 2):
                c = int(y / cell_size) * gw + int(x / cell_size)
                mask[i, c >> 5] |= np.uint32(1 << (c & 31))
        arr = (C.c_void_p * n)(*[f.h for f in frames])
        self._check(self.lib.sdvl_filter_corners_begin(self.h, n, arr, mask.ctypes.data_as(C.POINTER(C.c_uint32)), words, cell_size, margin,
                                                       min_feature_score, 1))
        cap = gw * gh
        counts = np.zeros(n, np.int32)
        rec = np.zeros((n, cap, 14), np.int32)      # 56-byte records: index, x, y, level, score, pad, desc[32]
        self._check(self.lib.sdvl_filter_corners_end(self.h, n, cap, _ptr(counts, i32p), rec.ctypes.data_as(C.c_void_p)))
        out = []
        for i in range(n):
            r = rec[i, :counts[i]]
            out.append((r[:, 0].copy(), r[:, 1:4].copy(), r[:, 4].copy(), r[:, 6:].copy().view(np.uint8).reshape(-1, 32)))
        return out

    def orb_describe_points(self, frame, xyl):
        xyl = np.ascontiguousarray(xyl, np.int32).reshape(-1, 3)
        desc = np.zeros((len(xyl), 32), np.uint8)
        ang = np.zeros(len(xyl), np.float32)
        self._check(self.lib.sdvl_orb_describe_points(self.h, frame.h, len(xyl), _ptr(xyl, i32p), _ptr(desc, u8p), _ptr(ang, f32p)))
        return desc, ang

    def hamming_argmin(self, queries, cand_lists, threshold=100):
        """ORBDetector::Distance over per-query candidate lists + SearchFeatures' arg-min (orb_detector.cc:398-410,
        matcher.cc:254-289).  queries [n][32] u8; cand_lists: n arrays [k_i][32] u8.  -> (best_index[n], best_dist[n])"""
        q = np.ascontiguousarray(queries, np.uint8).reshape(-1, 32)
        n = len(q)
        offs = np.zeros(n + 1, np.int32)
        offs[1:] = np.cumsum([len(c) for c in cand_lists])
        cands = (np.ascontiguousarray(np.concatenate([np.asarray(c, np.uint8).reshape(-1, 32) for c in cand_lists]))
                 if n else np.zeros((0, 32), np.uint8))
        if len(cands) == 0:
            cands = np.zeros((1, 32), np.uint8)
        bi = np.zeros(n, np.int32)
        bd = np.zeros(n, np.int32)
        self.lib.sdvl_hamming_argmin.argtypes = [C.c_void_p, C.c_int, u8p, i32p, u8p, C.c_int, i32p, i32p]
        self._check(self.lib.sdvl_hamming_argmin(self.h, n, _ptr(q, u8p), _ptr(offs, i32p), _ptr(cands, u8p), int(threshold),
                                                 _ptr(bi, i32p), _ptr(bd, i32p)))
        return bi, bd

    def image_align(self, jobs, feats, cam, ap):
        """jobs: list of (ref Frame, cur Frame, feat_begin, feat_end, T7); feats: AlignFeature array"""
        n = len(jobs)
        ja = (AlignJob * n)()
        for i, (ref, cur, b, e, T) in enumerate(jobs):
            ja[i].ref = ref.h.value
            ja[i].cur = cur.h.value
            ja[i].feat_begin, ja[i].feat_end = b, e
            for k in range(7):
                ja[i].T[k] = float(T[k])
        res = (AlignResult * n)()
        self._check(self.lib.sdvl_image_align(self.h, n, ja, len(feats), feats, C.byref(cam), C.byref(ap), res))
        return res

    def search_points(self, reqs, cam, sp):
        n = len(reqs)
        res = (SearchRes * n)()
        self._check(self.lib.sdvl_search_points(self.h, n, reqs, C.byref(cam), C.byref(sp), res))
        return res

    def frames_upload(self, frames, addresses, stride):
        """sdvl_frames_upload: images at host `addresses` (ints; pinned memory is read by one gather kernel, pageable memory is
        copied image by image) become level 0 of `frames`; the memory must stay alive until the stream has passed"""
        n = len(frames)
        fr = (C.c_void_p * n)(*[f.h for f in frames])
        im = (C.c_void_p * n)(*[int(a) for a in addresses])
        self._check(self.lib.sdvl_frames_upload(self.h, n, fr, im, int(stride)))

    def search_points_filter(self, reqs, cam, sp, states, fparams):
        """sdvl_search_points_filter without tracking tables: (search results, depth-filter outcomes)"""
        n = len(reqs)
        res = (SearchRes * n)()
        fout = (DepthOut * n)()
        self._check(self.lib.sdvl_search_points_filter(self.h, n, reqs, C.byref(cam), C.byref(sp), states, C.byref(fparams), None, res, fout))
        return res, fout

    def search_chain(self, reqs, cam, sp, trackers, req_points, fx, max_ransac_points=5, max_ransac_its=100, max_optim_pose_its=10,
                     inlier_error_threshold=2.0):
        """sdvl_search_begin / _slot / _run_chain / _chain_end.  reqs: SearchReq array (frames by handle); trackers: list of
        dict(cells=[[request index or -1, ...], ...] in SelectPoints order, max_matches, pose7, draws=raw rand() values);
        req_points[n][3].  Returns (search results, per tracker dict(pose, n_draws, refined, n_obs, inliers, outliers))."""
        n = len(reqs)
        lib = self.lib
        lib.sdvl_search_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(SearchReqPacked))]
        lib.sdvl_search_slot.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        packed = C.POINTER(SearchReqPacked)()
        self._check(lib.sdvl_search_begin(self.h, n, C.byref(packed)))
        for i, r in enumerate(reqs):
            d = packed[i]
            d.cur = lib.sdvl_search_slot(self.h, r.cur, r.cur_pose)
            d.ref = lib.sdvl_search_slot(self.h, r.ref, r.ref_pose)
            assert d.cur >= 0 and d.ref >= 0
            d.level, d.fixed = r.level, r.fixed
            d.px[0], d.px[1] = r.px[0], r.px[1]
            for k in range(3):
                d.bearing[k] = r.bearing[k]
            d.idepth, d.idepth_std = r.idepth, r.idepth_std
            d.px0[0], d.px0[1] = r.px0[0], r.px0[1]
            for k in range(32):
                d.desc[k] = r.desc[k]
        nf = len(trackers)
        cf = (ChainFrame * nf)()
        cand_req, cand_first, rand_raw = [], [], []
        for f, t in enumerate(trackers):
            cf[f].cand_begin = len(cand_req)
            for cell in t["cells"]:
                first = len(cand_req)
                for r in cell:
                    cand_req.append(int(r))
                    cand_first.append(first)
            cf[f].cand_end = len(cand_req)
            cf[f].max_matches = int(t["max_matches"])
            cf[f].rand_begin = len(rand_raw)
            rand_raw += [int(d) for d in t["draws"][:max_ransac_its]]
            for c in range(7):
                cf[f].pose[c] = float(t["pose"][c])
        cr = np.asarray(cand_req, np.int32); cfi = np.asarray(cand_first, np.int32); rr = np.asarray(rand_raw, np.int32)
        pts = np.ascontiguousarray(req_points, np.float64).reshape(n, 3)
        prm = PoseParams(max_ransac_points, max_ransac_its, max_optim_pose_its, 0, inlier_error_threshold / fx, fx)
        res = (SearchRes * n)()
        lib.sdvl_search_run_chain.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self._check(lib.sdvl_search_run_chain(self.h, n, C.byref(cam), C.byref(sp), res, nf, cf, len(cr), _ptr(cr, i32p), _ptr(cfi, i32p),
                                              _ptr(pts, f64p), len(rr), _ptr(rr, i32p), C.byref(prm)))
        pres = (PoseResult * nf)()
        n_obs = np.zeros(nf, np.int32)
        total = sum(int(t["max_matches"]) for t in trackers)
        lists = np.zeros(max(total, 1), np.int32)
        lib.sdvl_search_chain_end.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self._check(lib.sdvl_search_chain_end(self.h, nf, pres, _ptr(n_obs, i32p), _ptr(lists, i32p)))
        out, b = [], 0
        for f, t in enumerate(trackers):
            out.append(dict(pose=np.array(list(pres[f].pose)), n_draws=pres[f].n_draws, refined=pres[f].refined, n_obs=int(n_obs[f]),
                            inliers=lists[b:b + pres[f].n_inliers].copy(),
                            outliers=lists[b + pres[f].n_inliers:b + pres[f].n_inliers + pres[f].n_outliers].copy()))
            b += int(t["max_matches"])
        return res, out

    def align_patches(self, frames, levels, border, patch, uv, max_its=10):
        n = len(frames)
        arr = (C.c_void_p * n)(*[f.h for f in frames])
        levels = np.ascontiguousarray(levels, np.int32)
        border = np.ascontiguousarray(border, np.uint8).reshape(n, 100)
        patch = np.ascontiguousarray(patch, np.uint8).reshape(n, 64)
        uv = np.array(uv, np.float64).reshape(n, 2)
        conv = np.zeros(n, np.uint8)
        its = np.zeros(n, np.int32)
        self._check(self.lib.sdvl_align_patches(self.h, n, arr, _ptr(levels, i32p), _ptr(border, u8p), _ptr(patch, u8p), max_its,
                                                _ptr(uv, f64p), _ptr(conv, u8p), _ptr(its, i32p)))
        return uv, conv, its

    def pose_from_matches(self, jobs, fx, max_ransac_points=5, max_ransac_its=100, max_optim_pose_its=10, inlier_error_threshold=2.0):
        """jobs: list of (obs[n][6] = ax, ay, px, py, pz, level; pose7; rand_draws[max_ransac_its] raw rand() values).
        Returns per job dict(pose, n_draws, inliers, outliers, refined)."""
        n = len(jobs)
        pj = (PoseJob * n)()
        obs_all, rand_all, nits_all = [], [], []
        for k, (obs, pose, draws) in enumerate(jobs):
            obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 6)
            size = len(obs)
            pj[k].obs_begin = len(obs_all); pj[k].obs_end = len(obs_all) + size
            pj[k].rand_begin = len(rand_all); pj[k].nits_begin = len(nits_all)
            for c in range(7):
                pj[k].pose[c] = float(pose[c])
            for o in obs:
                obs_all.append(PoseObs(o[0], o[1], o[2], o[3], o[4], 1.0 / (1 << int(o[5]))))
            assert len(draws) >= max_ransac_its
            rand_all += [int(d) % size if size else 0 for d in draws[:max_ransac_its]]
            nits_all += ransac_budget_table(size, max_ransac_points, max_ransac_its)
        po = (PoseObs * max(len(obs_all), 1))(*obs_all)
        ra = np.asarray(rand_all, np.int32); ni = np.asarray(nits_all, np.int32)
        prm = PoseParams(max_ransac_points, max_ransac_its, max_optim_pose_its, 0, inlier_error_threshold / fx, fx)
        res = (PoseResult * n)()
        lists = np.zeros(max(len(obs_all), 1), np.int32)
        self.lib.sdvl_pose_from_matches.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._check(self.lib.sdvl_pose_from_matches(self.h, n, pj, len(obs_all), po, len(ra), _ptr(ra, i32p), len(ni), _ptr(ni, i32p),
                                                    C.byref(prm), res, _ptr(lists, i32p)))
        out = []
        for k in range(n):
            b = pj[k].obs_begin
            out.append(dict(pose=np.array(list(res[k].pose)), n_draws=res[k].n_draws, refined=res[k].refined,
                            inliers=lists[b:b + res[k].n_inliers].copy(),
                            outliers=lists[b + res[k].n_inliers:b + res[k].n_inliers + res[k].n_outliers].copy()))
        return out

    def undistort(self, imgs, cam, dist, frames=None):
        """Camera::UndistortImage for a batch of host images (numpy u8 [h, w]).  With `frames` the result becomes their
        level 0 (fused upload); otherwise it is returned as numpy arrays."""
        n = len(imgs)
        imgs = [np.ascontiguousarray(im, np.uint8) for im in imgs]
        h, w = imgs[0].shape
        src = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        d = Distortion((C.c_double * 5)(*[float(x) for x in dist]))
        if frames is not None:
            arr = (C.c_void_p * n)(*[f.h for f in frames])
            self._check(self.lib.sdvl_frames_upload_undistorted(self.h, n, arr, src, w, 0, C.byref(cam), C.byref(d)))
            return None
        buf = self.device_malloc(n * w * h)
        try:
            dst = (C.c_void_p * n)(*[buf + i * w * h for i in range(n)])
            self.lib.sdvl_undistort.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_int]
            self._check(self.lib.sdvl_undistort(self.h, n, src, w, 0, w, h, C.byref(cam), C.byref(d), dst, w))
            out = self.device_download(buf, n * w * h).reshape(n, h, w)
        finally:
            self.device_free(buf)
        return [out[i].copy() for i in range(n)]

    # ---- synthetic frames in HBM
    def device_malloc(self, nbytes):
        p = C.c_void_p()
        self._check(self.lib.sdvl_device_malloc(self.h, C.c_int64(nbytes), C.byref(p)))
        return p.value

    def device_free(self, p):
        self._check(self.lib.sdvl_device_free(self.h, C.c_void_p(p)))

    def device_download(self, p, nbytes):
        out = np.zeros(nbytes, np.uint8)
        self._check(self.lib.sdvl_device_download(self.h, C.c_void_p(p), C.c_int64(nbytes), _ptr(out, u8p)))
        return out

    def synth_render(self, views, width, height, dev_out, frame_bytes=None):
        n = len(views)
        arr = (SynthView * n)(*views)
        self._check(self.lib.sdvl_synth_render(self.h, n, arr, width, height, C.c_void_p(dev_out), C.c_int64(frame_bytes or width * height)))


def default_detect_params(use_orb=True, orb_size=31, patch_size=8):
    """Config defaults (config.cc:55-85) with the TUM cfg overrides (config/config_tum_f1.cfg:34-42)."""
    return DetectParams(cell_size=32, max_fast_levels=3, fast_threshold=10,
                        margin=(4 + orb_size // 2) if use_orb else (1 + patch_size // 2))


def default_align_params(fast=False):
    return AlignParams(max_level=4, min_level=2, max_its=30, patch_size=4, fast=int(fast))


def default_search_params(use_orb=True, orb_size=31, patch_size=8):
    return SearchParams(patch_size=8, max_align_its=10, search_size=6, max_fast_levels=3,
                        margin=(4 + orb_size // 2) if use_orb else (1 + patch_size // 2), use_orb=int(use_orb))


class Feed:
    """sdvl_feed: one copy stream filled by one thread; consumers (contexts) acquire / release its slots"""

    def __init__(self, device=0, n_slots=2):
        self.lib = load_library()
        self.lib.sdvl_feed_last_error.restype = C.c_char_p
        self.lib.sdvl_feed_last_error.argtypes = [C.c_void_p]
        self.lib.sdvl_feed_slot_arrived.argtypes = [C.c_void_p, C.c_int]
        h = C.c_void_p()
        if self.lib.sdvl_feed_create(device, n_slots, C.byref(h)) != 0:
            raise SdvlError("sdvl_feed_create failed")
        self.h = h

    def images(self, slot, host_ptrs, stride, width, height, dev_dst):
        n = len(host_ptrs)
        src = (C.c_void_p * n)(*[int(p) for p in host_ptrs])
        dst = (C.c_void_p * n)(*[int(p) for p in dev_dst])
        if self.lib.sdvl_feed_images(self.h, slot, n, src, stride, width, height, dst) != 0:
            raise SdvlError("sdvl_feed_images: %s" % self.lib.sdvl_feed_last_error(self.h).decode())

    def arrived(self, slot):
        """True once the slot's last transfer is in HBM (never blocks)"""
        r = self.lib.sdvl_feed_slot_arrived(self.h, slot)
        if r < 0:
            raise SdvlError("sdvl_feed_slot_arrived: %s" % self.lib.sdvl_feed_last_error(self.h).decode())
        return r == 1

    def acquire(self, ctx, slot):
        ctx._check(self.lib.sdvl_ctx_feed_acquire(ctx.h, self.h, slot))

    def release(self, ctx, slot):
        ctx._check(self.lib.sdvl_ctx_feed_release(ctx.h, self.h, slot))

    def close(self):
        if self.h:
            self.lib.sdvl_feed_destroy(self.h)
            self.h = None
