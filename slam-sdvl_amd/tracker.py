"""ctypes view of the host layer (host/libsdvl_host.so): B independent SDVL trackers on one MI355X stepping
together (sdvl::SDVLBatch).  No CPU fallback: construction fails without a GPU."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "host", "libsdvl_host.so")


class FrameStats(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("state", "quality", "matches", "attempts", "inliers", "outliers", "n_corners",
                                       "align_meas", "keyframe", "relocalized")] + [("pose", C.c_double * 7)] + [
        (n, C.c_int) for n in ("align_features", "align_iters", "search_requests", "lk_iters", "host_path")]


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
        lib.sdvlh_batch_step_device_transient.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.sdvlh_batch_set_next_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
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


def set_device_pose(on):
    """pose stage (RANSAC + refinement, feature_align.cc:73-82,152-243) on the device (default) or the host implementation"""
    load_host_library().sdvlh_set_device_pose(int(bool(on)))


def set_track_tables(on):
    """device-resident tracking tables (default for persistent batches) on / off: off = the host assembles every request"""
    load_host_library().sdvlh_set_track_tables(int(bool(on)))


def set_device_filter(on):
    """the mapper's depth filter (Point::Update / HasConverged behind the candidate search) on the device (default) / on the host"""
    load_host_library().sdvlh_set_device_filter(int(bool(on)))


def bind_to_gpu_numa_node(gpu=0):
    """Restrict this process (and the threads it creates later) to the CPUs of the NUMA node the GPU hangs off: the host
    objects of the trackers, the pinned staging buffers and the doorbell writes then stay on that socket.  Returns the
    node, or None when the topology cannot be read (nothing is changed then).  Call it before creating devices."""
    import os
    try:
        import torch
        p = torch.cuda.get_device_properties(gpu)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def set_mapper(on):
    """map mode of the batches / farms created afterwards: the reference's mapper in sequential mode (map.cc) instead of
    the plane map stub; the first keyframe is bootstrapped from the scene plane in both modes"""
    load_host_library().sdvlh_set_mapper(int(bool(on)))


def device_pose():
    return bool(load_host_library().sdvlh_device_pose())


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

    def set_next_device(self, dev_ptrs):
        """look-ahead: the device images the NEXT step_device call will be given (None: none); their pyramids and corner detection
        are queued behind the coming step's search / pose chain (SDVLBatch::SetNextImages).  The buffers must keep their CONTENT until that
        call: the look-ahead is recognised by device address alone.  It belongs to the one coming step and is dropped if that step cannot use it"""
        ptrs = (C.c_void_p * self.B)(*[int(p) for p in dev_ptrs]) if dev_ptrs is not None else None
        if self.lib.sdvlh_batch_set_next_device(self.h, ptrs, self.w) != 0:
            raise RuntimeError(self.lib.sdvlh_last_error().decode())

    def set_distortion(self, dist5):
        """the batch's camera gets a lens (Camera.d1..d5 of the reference's cfg files) and the images of every later step are RAW camera
        frames: Camera::UndistortImage (main.cc:133) runs inside the step, fused into the frames' upload"""
        d = np.ascontiguousarray(dist5, np.float64)
        self.lib.sdvlh_batch_set_distortion.argtypes = [C.c_void_p, C.c_void_p]
        if self.lib.sdvlh_batch_set_distortion(self.h, d.ctypes.data) != 0:
            raise RuntimeError("sdvlh_batch_set_distortion")

    def step_device_transient(self, dev_ptrs):
        """frames in HBM that stay valid for THIS step only (a slot of an input ring): aliased while tracked, frames that become
        keyframes take a copy at the end of the step"""
        ptrs = (C.c_void_p * self.B)(*[int(p) for p in dev_ptrs])
        if self.lib.sdvlh_batch_step_device_transient(self.h, ptrs, self.w, self._stats) != 0:
            raise RuntimeError(self.lib.sdvlh_last_error().decode())
        return self._stats

    STAGES = ["upload_pyr", "fast", "select", "corners_orb", "prelude", "image_align", "prepare", "search", "finish", "pose", "mapping",
              "epilogue", "mapper", "total", "map_candidates", "map_connections", "map_init", "map_finish",
              "map_begin", "map_emit", "map_search", "map_apply", "relocalize"]

    def map_stats(self, i):
        """mapper mode only: candidates, converged, initialized, linked, connected, keyframes of tracker i"""
        out = (C.c_int * 6)()
        self.lib.sdvlh_batch_map_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        if self.lib.sdvlh_batch_map_stats(self.h, int(i), out) != 0:
            raise RuntimeError("tracker %d has no mapper" % i)
        return dict(zip(("candidates", "converged", "initialized", "linked", "connected", "keyframes"), list(out)))

    def stage_times(self, reset=False):
        """accumulated wall seconds per host stage of SDVLBatch::HandleFrames -> (dict, steps)"""
        out = (C.c_double * len(self.STAGES))()
        self.lib.sdvlh_batch_stage_times.restype = C.c_long
        self.lib.sdvlh_batch_stage_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        n = self.lib.sdvlh_batch_stage_times(self.h, out, len(self.STAGES), int(reset))
        return {k: out[i] for i, k in enumerate(self.STAGES)}, n

    def close(self):
        if self.h:
            self.lib.sdvlh_batch_destroy(self.h)
            self.h = None


class TrackerFarm:
    """G groups x Bg sequences on one GPU; each group = host thread + sdvl_ctx (stream) + SDVLBatch, free-running."""

    def __init__(self, gpu, G, Bg, w, h, cam4, plane4=(0, 0, 1, 2.0), first_poses=None, host_threads_per_group=1):
        self.lib = load_host_library()
        self.G, self.Bg, self.w, self.h_img = G, Bg, w, h
        n = G * Bg
        cam4 = np.ascontiguousarray(cam4, np.float64)
        plane4 = np.ascontiguousarray(plane4, np.float64)
        if first_poses is None:
            first_poses = np.tile(np.array([1, 0, 0, 0, 0, 0, 0], np.float64), (n, 1))
        first_poses = np.ascontiguousarray(first_poses, np.float64).reshape(n, 7)
        self.lib.sdvlh_farm_create.restype = C.c_void_p
        self.lib.sdvlh_farm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.sdvlh_farm_destroy.argtypes = [C.c_void_p]
        self.lib.sdvlh_farm_ctx.restype = C.c_void_p
        self.lib.sdvlh_farm_ctx.argtypes = [C.c_void_p, C.c_int]
        self.lib.sdvlh_farm_batch.restype = C.c_void_p
        self.lib.sdvlh_farm_batch.argtypes = [C.c_void_p, C.c_int]
        self.lib.sdvlh_farm_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        self.lib.sdvlh_farm_reserve.argtypes = [C.c_void_p, C.c_int]
        self.h = self.lib.sdvlh_farm_create(gpu, G, Bg, w, h, cam4.ctypes.data, plane4.ctypes.data, first_poses.ctypes.data,
                                            host_threads_per_group)
        if not self.h:
            raise RuntimeError("no MI355X: %s" % self.lib.sdvlh_last_error().decode())

    def ctx_handle(self, g=0):
        return self.lib.sdvlh_farm_ctx(self.h, g)

    def reserve(self, frames_per_group):
        """pool HBM frames up front (keyframes keep theirs): no hipMalloc on the tracking path for that many frames"""
        if self.lib.sdvlh_farm_reserve(self.h, int(frames_per_group)) != 0:
            raise RuntimeError(self.lib.sdvlh_last_error().decode())

    def set_distortion(self, dist5):
        """every group's camera gets the lens; the frames of run() are RAW camera frames from then on (TrackerBatch.set_distortion)"""
        d = np.ascontiguousarray(dist5, np.float64)
        self.lib.sdvlh_batch_set_distortion.argtypes = [C.c_void_p, C.c_void_p]
        for g in range(self.G):
            if self.lib.sdvlh_batch_set_distortion(self.lib.sdvlh_farm_batch(self.h, g), d.ctypes.data) != 0:
                raise RuntimeError("sdvlh_batch_set_distortion")

    def set_fibers(self, n):
        """n > 1: every worker thread interleaves n group-steps, switching at GPU waits (use G = n * workers groups)"""
        self.lib.sdvlh_farm_set_fibers.argtypes = [C.c_void_p, C.c_int]
        self.lib.sdvlh_farm_set_fibers(self.h, int(n))

    def set_host_input(self, on):
        """the pointers of run() are HOST pointers (pinned memory): every step uploads its frames inside the step"""
        self.lib.sdvlh_farm_set_host_input.argtypes = [C.c_void_p, C.c_int]
        self.lib.sdvlh_farm_set_host_input(self.h, int(bool(on)))

    def alloc_stats(self, n_steps):
        """output records for run(); touched here so that the workers do not take the first-touch page faults"""
        out = (FrameStats * (n_steps * self.G * self.Bg))()
        C.memset(out, 0, C.sizeof(out))
        return out

    def run(self, dev_frames, workers=0, out=None):
        """dev_frames: int array [n_steps, G*Bg] of device pointers -> FrameStats array [n_steps*G*Bg];
        workers = host threads (0 = one per group); workers < G schedules group-steps dynamically"""
        dev_frames = np.ascontiguousarray(dev_frames, np.uint64)
        n_steps = dev_frames.shape[0]
        assert dev_frames.shape[1] == self.G * self.Bg
        if out is None:
            out = self.alloc_stats(n_steps)
        assert len(out) >= n_steps * self.G * self.Bg
        if self.lib.sdvlh_farm_run(self.h, n_steps, dev_frames.ctypes.data, self.w, out, int(workers)) != 0:
            raise RuntimeError(self.lib.sdvlh_last_error().decode())
        return out

    def feed_stats(self):
        """last host-fed run -> (seconds the feeder spent inside the transfer calls, seconds it waited for a free ring slot, transfers,
        seconds the workers waited for their transfer to be issued and to arrive (summed over groups), group-steps that began before their
        images had arrived, seconds the feeder waited for its own earlier transfers)"""
        out = (C.c_double * 6)()
        self.lib.sdvlh_farm_feed_stats.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.sdvlh_farm_feed_stats(self.h, out)
        return out[0], out[1], int(out[2]), out[3], int(out[4]), out[5]

    def set_input_ring(self, on):
        """host-fed runs: the images of step s + 1 travel on the group's copy stream while step s computes (default on)"""
        self.lib.sdvlh_farm_set_input_ring.argtypes = [C.c_void_p, C.c_int]
        self.lib.sdvlh_farm_set_input_ring(self.h, int(bool(on)))

    def stage_times(self, reset=False):
        tot, steps = None, 0
        self.lib.sdvlh_batch_stage_times.restype = C.c_long
        self.lib.sdvlh_batch_stage_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        for g in range(self.G):
            out = (C.c_double * len(TrackerBatch.STAGES))()
            n = self.lib.sdvlh_batch_stage_times(self.lib.sdvlh_farm_batch(self.h, g), out, len(TrackerBatch.STAGES), int(reset))
            steps += n
            tot = [a + b for a, b in zip(tot, out)] if tot else list(out)
        return {k: tot[i] for i, k in enumerate(TrackerBatch.STAGES)}, steps

    def close(self):
        if self.h:
            self.lib.sdvlh_farm_destroy(self.h)
            self.h = None
