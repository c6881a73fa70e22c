"""ctypes view of the host layer (host/libsdvl_host.so): B independent SDVL trackers on one MI355X stepping
together (sdvl::SDVLBatch).  No CPU fallback: construction fails without a GPU."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "host", "libsdvl_host.so")


class FrameStats(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("state", "quality", "matches", "attempts", "inliers", "outliers", "n_corners",
                                       "align_meas", "keyframe", "relocalized")] + [("pose", C.c_double * 7)]


_lib = None


def load_host_library():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError("libsdvl_host.so is not built (%s): run __graft_entry__.build()" % HOST_LIB_PATH)
        lib = C.CDLL(HOST_LIB_PATH)
        lib.sdvlh_last_error.restype = C.c_char_p
        lib.sdvlh_device_create.restype = C.c_void_p
        lib.sdvlh_device_ctx.restype = C.c_void_p
        lib.sdvlh_device_ctx.argtypes = [C.c_void_p]
        lib.sdvlh_device_destroy.argtypes = [C.c_void_p]
        lib.sdvlh_batch_create.restype = C.c_void_p
        lib.sdvlh_batch_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.sdvlh_batch_destroy.argtypes = [C.c_void_p]
        lib.sdvlh_batch_step_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.sdvlh_batch_step_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.sdvlh_config_set.argtypes = [C.c_char_p, C.c_double]
        _lib = lib
    return _lib


# config/config_tum_f1.cfg:34-42 over the defaults of config.cc:55-85
TUM_OVERRIDES = {"SDVL.cell_size": 32, "SDVL.min_avg_shift": 5, "SDVL.max_matches": 200, "SDVL.max_keyframes": 1000,
                 "SDVL.use_orb": 1, "SDVL.fast_threshold": 10, "SDVL.lost_ratio": 0.7, "SDVL.num_features": 1000}


def configure(overrides=None):
    lib = load_host_library()
    lib.sdvlh_config_reset()
    for k, v in (TUM_OVERRIDES if overrides is None else overrides).items():
        if lib.sdvlh_config_set(k.encode(), float(v)) != 0:
            raise KeyError(k)


class HostDevice:
    def __init__(self, gpu=0):
        self.lib = load_host_library()
        self.h = self.lib.sdvlh_device_create(gpu)
        if not self.h:
            raise RuntimeError("no MI355X: %s" % self.lib.sdvlh_last_error().decode())

    def ctx_handle(self):
        return self.lib.sdvlh_device_ctx(self.h)

    def close(self):
        if self.h:
            self.lib.sdvlh_device_destroy(self.h)
            self.h = None


class TrackerBatch:
    def __init__(self, dev, B, w, h, cam4, plane4=(0, 0, 1, 2.0), first_poses=None, host_threads=1):
        self.lib = dev.lib
        self.dev = dev
        self.B, self.w, self.h_img = B, w, h
        cam4 = np.ascontiguousarray(cam4, np.float64)
        plane4 = np.ascontiguousarray(plane4, np.float64)
        if first_poses is None:
            first_poses = np.tile(np.array([1, 0, 0, 0, 0, 0, 0], np.float64), (B, 1))
        first_poses = np.ascontiguousarray(first_poses, np.float64).reshape(B, 7)
        self.h = self.lib.sdvlh_batch_create(dev.h, B, w, h, cam4.ctypes.data, plane4.ctypes.data, first_poses.ctypes.data, host_threads)
        if not self.h:
            raise RuntimeError(self.lib.sdvlh_last_error().decode())
        self._stats = (FrameStats * B)()

    def step_host(self, imgs):
        imgs = [np.ascontiguousarray(im, np.uint8) for im in imgs]
        ptrs = (C.c_void_p * self.B)(*[im.ctypes.data for im in imgs])
        if self.lib.sdvlh_batch_step_host(self.h, ptrs, self.w, self._stats) != 0:
            raise RuntimeError(self.lib.sdvlh_last_error().decode())
        return self._stats

    def step_device(self, dev_ptrs):
        ptrs = (C.c_void_p * self.B)(*[int(p) for p in dev_ptrs])
        if self.lib.sdvlh_batch_step_device(self.h, ptrs, self.w, self._stats) != 0:
            raise RuntimeError(self.lib.sdvlh_last_error().decode())
        return self._stats

    def close(self):
        if self.h:
            self.lib.sdvlh_batch_destroy(self.h)
            self.h = None
