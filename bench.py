#!/usr/bin/env python3
"""bench.py — tracked frames/sec of the MI355X-native SDVL front-end (BASELINE.json metric).

A "step" = one pass of the hot path (SDVL::HandleFrame in STATE_RUNNING: pyramid, FAST/ORB, sparse image alignment,
grid reprojection with SearchPoint/AlignPatch, RANSAC + pose refinement) over one batch of B independent synthetic
640x480 sequences per GPU (workload S-A of SURVEY §8d, TUM intrinsics / TUM cfg parameters, 5-level pyramid,
max 200 matches).  All input frames are rendered into HBM before the timed region starts.

  python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU - started by torch.distributed.run, or by this
                                                          file itself when it is launched directly; fewer GPUs than N = error)

Sequences shard across ranks with no data-path collective ("weak" scaling: B sequences per GPU); RCCL is used only
for the barrier and the throughput reduction.  Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP events
on the kernel's own stream over the timed region) and `cpu_baseline` (the CPU oracle on one host core and on all usable host cores,
bounded sample; N = 1 only).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
PCIE_LONE_STREAM_GBS = 57.0   # H2D of large pinned pieces on one stream, measured on these boxes (profiles/r03/pcie_probe.txt)
W_IMG, H_IMG = 640, 480
TUM_CAM = np.array([517.3, 516.5, 318.6, 255.3])
FEATS_LABEL = "~200 feats"
ORACLE_PARAMS = {}      # fields of the oracle's parameter block that differ from its defaults (workload S-C)
# S-A is the metric configuration; S-C (1280x960, num_features 4000, max_matches 1000, 64 sequences per GPU)
# is BASELINE.json's roofline case, run with `--workload S-C`
WORKLOADS = {
    # seqs: tried in this order, the first whose resident input frames + keyframe pool fit the GPU is used (with the device-
    # resident tracking tables the kernels are the limit, and 256 sequences per launch amortise their latency better than 128)
    "S-A": dict(w=640, h=480, cam=[517.3, 516.5, 318.6, 255.3], seqs=[4096, 3072, 2048], over={}, label="~200 feats", cpu_frames=300),
    # SURVEY §8(d) S-B / BASELINE config 4: EuRoC MH_01's geometry (config/config_euroc.cfg:9-14,42: 752x480, its intrinsics,
    # min_matches 5), four independent 300-frame chunks with seeds 20260010..13.  On one GPU the sequences follow the chunks round-robin
    # (sequence g: chunk g mod 4); with `--gpus 4` every rank tracks ONE chunk (shard.chunk_for_sequence)
    "S-B": dict(w=752, h=480, cam=[458.654, 457.296, 367.215, 248.375], seqs=[4096, 3072, 2048], over={"SDVL.min_matches": 5}, label="~200 feats",
                cpu_frames=300, chunks=4, seed0=20260010),
    # (seeds 20260100 + rank*64 + i, SURVEY §8d: with 64 sequences per GPU that is 20260100 + the global sequence index)
    "S-C": dict(w=1280, h=960, cam=[1034.6, 1033.0, 637.2, 510.6], seqs=[64], label="~1000 feats", cpu_frames=60, seed0=20260100,
                over={"SDVL.num_features": 4000, "SDVL.max_matches": 1000}),
}
XI = np.array([0.004, 0.002, 0.001, 0.0008, -0.0012, 0.0005])
TEXTURES = {"plane": 0, "camera": 1}   # csrc/sdvl_synth.h: SDVL_TEXTURE_PLANE_NOISE (rounds 1-4: a FAST corner on every second pixel) / SDVL_TEXTURE_CAMERA
TEXTURE = 0
# the lenses of the reference's configuration files (Camera.d1..d5): with --distortion every leg's frames are what a camera with that lens
# records and Camera::UndistortImage (camera.cc:100-105, main.cc:133) runs on every frame, fused into the upload of the tracked step
DISTORTIONS = {"none": None,
               "tum_f1": [0.2624, -0.9531, -0.0054, 0.0026, 1.1633],       # config/config_tum_f1.cfg:15-19
               "tum_f2": [0.2312, -0.7849, -0.0033, -0.0001, 0.9172],      # config/config_tum_f2.cfg:15-19
               "euroc": [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0]}   # config/config_euroc.cfg:15-19
DISTORTION = None
SEQ_SEED = None     # global sequence index -> texture seed (main() sets it from the workload)


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def se3_exp(u):
    """SE3::Exp (extra/se3.cc:72-94) in numpy, used only to lay out the synthetic trajectories."""
    ups, om = u[:3], u[3:]
    th = np.linalg.norm(om)
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        q = np.array([1.0, *(0.5 * om)])
        V = np.eye(3)
    else:
        q = np.array([np.cos(th / 2), *(np.sin(th / 2) / th * om)])
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om
    return np.concatenate([q, V @ ups])


def make_view(pkg, T, seed, frame_id):
    v = pkg.SynthView()
    v.fx, v.fy, v.u0, v.v0 = [float(c) for c in TUM_CAM]
    R = quat_to_R(T[:4])
    for i in range(9):
        v.R[i] = float(R.flat[i])
    for i in range(3):
        v.t[i] = float(T[4 + i])
    v.plane[0], v.plane[1], v.plane[2], v.plane[3] = 0.0, 0.0, 1.0, 2.0
    v.seed, v.frame_id, v.texture = seed, frame_id, TEXTURE
    for i in range(5):
        v.dist[i] = float(DISTORTION[i]) if DISTORTION is not None else 0.0
    return v



class CtxView:
    """minimal view of the sdvl_ctx owned by the host layer's Device (timing + synthetic rendering)"""

    def __init__(self, pkg, handle):
        self.lib = pkg.load_library()
        self.h = C.c_void_p(handle)
        self.pkg = pkg

    def check(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.sdvl_last_error(self.h).decode())

    def malloc(self, n):
        p = C.c_void_p()
        self.check(self.lib.sdvl_device_malloc(self.h, C.c_int64(n), C.byref(p)))
        return p.value

    def render(self, views, dev_out):
        arr = (self.pkg.SynthView * len(views))(*views)
        self.check(self.lib.sdvl_synth_render(self.h, len(views), arr, W_IMG, H_IMG, C.c_void_p(dev_out), C.c_int64(W_IMG * H_IMG)))

    def download(self, p, n):
        out = np.zeros(n, np.uint8)
        self.check(self.lib.sdvl_device_download(self.h, C.c_void_p(p), C.c_int64(n), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def timing(self, on, only=None):
        self.lib.sdvl_ctx_timing_only.argtypes = [C.c_void_p, C.c_char_p]
        self.check(self.lib.sdvl_ctx_timing_only(self.h, only.encode() if only else None))
        self.check(self.lib.sdvl_ctx_timing_enable(self.h, int(on)))
        self.check(self.lib.sdvl_ctx_timing_reset(self.h))

    def timing_get(self):
        names = ((C.c_char * 32) * 32)()
        ms = (C.c_double * 32)()
        launches = (C.c_int64 * 32)()
        n = C.c_int()
        self.check(self.lib.sdvl_ctx_timing_get(self.h, 32, names, ms, launches, C.byref(n)))
        return {names[i].value.decode(): (ms[i], launches[i]) for i in range(n.value)}


def algorithmic_bytes_per_frame(kernel, n_c, n_f, n_s, i_ia, i_fa, n_m=190, n_kp=10000):
    """SURVEY §8(d) per-frame algorithmic bytes, split by kernel (640x480, 5 levels, 3 FAST levels)."""
    P = [(W_IMG >> l) * (H_IMG >> l) for l in range(5)]
    return {
        # per-cell retainBest: every keypoint list read once, the surviving heads (~1.5 n_c) and one length per cell written back
        "select_cells": 4 * n_kp + 4 * 1.5 * n_c + 4 * 400,
        # per-level retainBest + corners_ + bins: lengths and surviving heads read, corner records + bin entries written
        "select_pack": 4 * 400 + 4 * 1.5 * n_c + (16 + 8) * n_c,
        "pyr_down": sum(P[:4]) + sum(P[1:]),                         # pyramid read + write (4 launches per frame batch)
        "fast_cells": sum(P[:3]) + 16 * n_c,                         # FAST read + keypoint write
        "orb_describe": n_c * (961 + 32),                            # 31x31 window + descriptor
        # round 4: two launches — the reference windows of the three levels (+ the table rows the features come from) are read by
        # image_align_pre, the current windows per Gauss-Newton evaluation by image_align (the items between them stay in L2)
        "image_align_pre": 3 * n_f * 49 + n_f * (48 + 24),
        "image_align": i_ia * n_f * 25,
        # corner list + warp window + patches + LK windows + the 31x31 ORB window of the matched corner (descriptors are
        # computed by the search itself, for the corners it compares: at least one per match)
        "search_points": 12 * n_c + n_s * (121 + 164 + i_fa * 81) + n_m * 961,
        "search_prepare": n_s * (120 + 80),                            # request records read, scalar-phase records written
        "pose_hypotheses": 5 * 48 * 100 + 100 * 64,                  # the five matches of every RANSAC draw + one result per draw
        "pose_supporters": 48 * n_m + 100 * 64,                      # match records read once (L2 serves the draws' re-reads) + the draws' results
        "pose_refine": 48 * n_m + 100 * 64 + 4 * n_m + 80,           # matches + draw results read, index lists + pose written
        # device-resident tracking tables (sdvl_track.hip): feature + point rows read, alignment records / requests / new rows written
        "track_align_prep": n_f * (48 + 24 + 56),
        "track_project": n_f * (48 + 144) + n_s * (136 + 24 + 8 + 120),   # (+ the scalar-phase record of every request: search_prepare's work)
        "track_commit": n_s * (40 + 8) + n_m * (48 + 24) + n_f * 16,
        # keyframes only: Shi-Tomasi scores + corner records read, one record + one 31x31 ORB window per kept corner (~1 in 4)
        "filter_select": n_c * (16 + 8) + 0.25 * n_c * 56,
        "filter_describe": 0.25 * n_c * (961 + 32),
        "shi_tomasi": n_c * (100 + 8),
        "filter_gather": n_c * (16 + 8 + 32),
        "image_align_big": 3 * n_f * 49 + i_ia * n_f * 25,          # the same kernel with its caches in HBM (> 384 features per job)
        "select_matches": 40 * n_s + 48 * n_m,
        "depth_filter": 184 * n_s,                                   # mapper: filter state in (104 B) and out (80 B) per candidate request
        "align_patches": n_s * (164 + i_fa * 81),
        "frames_upload": 2 * P[0], "undistort": 2 * P[0],            # image read + level 0 written
        "compact_cells": 8 * n_kp, "registry_write": 128, "track_upload": n_f * (144 + 48),
    }.get(kernel)


VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2   # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles at 2.4 GHz


PROFILE_TAG = "sa_plane"    # which committed profile belongs to this run: profiles/rNN/<workload>_<texture>/ (main() sets it)


def _profile_files(fname):
    """committed profile files for this run's workload and texture, oldest first: profiles/rNN/<tag>/<fname> (round 5 on), and for S-A
    on the plane texture also rounds 1-4's profiles/rNN/<fname>"""
    import glob
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*", PROFILE_TAG, fname)))
    if PROFILE_TAG == "sa_plane":
        files = sorted(glob.glob(os.path.join(root, "profiles", "r*", fname))) + files
    return files, root


def _pmc_rows(fname):
    """rows of the newest committed <fname> for this workload and texture (written by tools/summarize_profiles.py from separate --pmc passes)"""
    import csv
    files, root = _profile_files(fname)
    if not files:
        return [], None
    with open(files[-1]) as fh:
        return list(csv.DictReader(fh)), os.path.relpath(files[-1], root)


def profile_staleness():
    """(stale, detail): were the committed PMC passes of this workload and texture (newest profiles/rNN/<tag>/source_stamp.json) taken on
    the kernel sources of THIS tree?  stale = None when no stamp is committed for it (rounds 1-4's profiles carry none)."""
    files, root = _profile_files("source_stamp.json")
    if not files:
        return None, "no source_stamp.json committed for %s" % PROFILE_TAG
    try:
        st = json.load(open(files[-1]))
        stamp = importlib.import_module("slam-sdvl_amd.stamp").kernel_source_stamp()
    except (OSError, ValueError) as e:
        return None, "unreadable stamp: %s" % e
    changed = sorted(k for k in set(st.get("files", {})) | set(stamp["files"]) if st.get("files", {}).get(k) != stamp["files"].get(k))
    return bool(changed), {"stamp": os.path.relpath(files[-1], root), "profiled_workload": st.get("workload"), "profiled_texture": st.get("texture"),
                           "changed_files": changed}


def measured_issue_rates():
    """wave instructions per second that streams of independent instructions reach on this chip at 8 waves per SIMD
    (tools/valu_peak_probe.cpp, committed as profiles/rNN/valu_peak_probe.txt): packed-f16, permute and FP64 instructions issue at
    about half the rate of FP32 / integer ones — fast_cells is mostly the former.  {} when no probe output is committed."""
    import glob
    import re
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*", "valu_peak_probe.txt")))
    out = {}
    if files:
        for line in open(files[-1]):
            m = re.match(r"(\S.*?)\s+8 wave\(s\) per SIMD:.*?([0-9.]+e[+-]?[0-9]+) wave instructions/s", line)
            if m:
                out[m.group(1).strip()] = float(m.group(2))
    return out


# timer names whose kernels carry another name in the rocprofv3 output (the forms the tracked step launches come first)
PMC_ALIASES = {
    "image_align": ["image_align_track_wave_pre_kernel", "image_align_wave_pre_kernel", "image_align_track_kernel", "image_align_wave_kernel",
                    "image_align_lds_kernel"],
    "image_align_pre": ["image_align_track_pre_kernel", "image_align_pre_kernel"],
    "filter_select": ["filter_select_binned_kernel", "filter_select_kernel"],
}


def _pmc_row(rows, kernel):
    """the row of timer name `kernel` (rocprofv3 names look like 'fast_cells_wave_kernel' or 'void image_align_lds_kernel<false>')"""
    for prefix in PMC_ALIASES.get(kernel, []):
        for row in rows:
            if row["kernel"].replace("void ", "").startswith(prefix):
                return row
    for row in rows:
        name = row["kernel"].replace("void ", "")
        if name.startswith(kernel + "_") or name.startswith(kernel + "<"):
            return row
    return None


def _frames_per_dispatch(row):
    import re
    m = re.search(r"-> (\d+) frames per dispatch", row.get("note", ""))
    return float(m.group(1)) if m else None


def pmc_traffic_bytes(kernel, frames_per_launch):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/rNN/pmc_hbm_traffic.csv, newest round), scaled to
    this run's frames per launch.  The counters need their own rocprofv3 passes (tools/profile_bench.sh), so they cannot be
    sampled inside the timed region.  FETCH_SIZE on gfx950 tallies a 128-B request as 64 B (MI355X_MICROARCH.md, HBM): the guide
    calibrates the x2 for 16-B-per-lane streams; summarize_profiles.py marks per row whether the kernel's loads are of that kind
    (column fetch_x2) and the figure here applies it.  Returns (bytes or None, where the figure came from)."""
    rows, src_file = _pmc_rows("pmc_hbm_traffic.csv")
    row = _pmc_row(rows, kernel)
    if row is None:
        return None, None
    per = _frames_per_dispatch(row)
    try:
        fx = 2.0 if row.get("fetch_x2", "0") in ("1", "2", "2.0") else 1.0
        total = (float(row["FETCH_SIZE_avg_KB"]) * fx + float(row["WRITE_SIZE_avg_KB"])) * 1024.0
    except ValueError:
        return None, None
    if not per or total != total:
        return None, None
    src = "%s row '%s': (FETCH_SIZE_avg_KB %s x %g + WRITE_SIZE_avg_KB %s) * 1024 per %d-frame dispatch, scaled to %.0f frames" % (
        src_file, row["kernel"], row["FETCH_SIZE_avg_KB"], fx, row["WRITE_SIZE_avg_KB"], int(per), frames_per_launch)
    return int(total * frames_per_launch / per), src


def pmc_valu(timers_ms_per_step, frames_per_step):
    """wave-level VALU instructions from the committed SQ pass (profiles/rNN/pmc_sq_lds.csv, SQ_INSTS_VALU per dispatch):
    per kernel {insts per dispatch, frames per dispatch, dispatches} and the whole path's instructions per tracked frame
    (every kernel's instructions x its dispatches / the frames the run processed).  None when no SQ pass is committed."""
    rows, src_file = _pmc_rows("pmc_sq_lds.csv")
    if not rows:
        return None
    out, ref_frames = {}, None
    fast = _pmc_row(rows, "fast_cells")
    if fast is None:
        return None
    per = _frames_per_dispatch(fast)
    if not per:
        return None
    ref_frames = per * float(fast["dispatches"])          # fast_cells runs once per frame batch: frames the profiled run processed
    path = 0.0
    for name in timers_ms_per_step:
        row = _pmc_row(rows, name)
        if row is None:
            continue
        try:
            insts = float(row["SQ_INSTS_VALU_avg"])
            disp = float(row["dispatches"])
        except (KeyError, ValueError):
            continue
        out[name] = {"insts_per_dispatch": insts, "insts_per_frame": insts * disp / ref_frames}
        path += insts * disp / ref_frames
    return {"kernels": out, "path_insts_per_frame": path, "source": src_file, "frames_per_dispatch": per}


def effective_cpus():
    """CPUs this process may actually use: affinity mask and cgroup quota (the GPU boxes run under a CPU quota)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(frames, mapper=False, threads=1, ref_flags=False):
    """the CPU oracle (oracle/, kind "port") over a bounded sample of sequence 0: `threads` independent trackers, one host
    thread each (the reference tracker is single-threaded, so N cores = N independent sequences).  The ctypes
    call releases the GIL, every tracker owns its state.  Returns (tracked frames/s over all threads, tracked, wall s)."""
    import threading
    import oraclelib as ol
    orc = ol.Oracle(ref_flags)      # ref_flags: the timing-only build with the reference's compiler flags (oracle/Makefile)
    for k, v in ORACLE_PARAMS.items():
        setattr(orc.params, k, v)
    out = [None] * threads

    def run(i):
        trk = orc.tracker(W_IMG, H_IMG, TUM_CAM)
        if mapper:
            trk.use_mapper(True)
        tracked, t_total = 0, 0.0
        for k, im in enumerate(frames):
            t0 = time.perf_counter()
            st = trk.handle_frame(im)
            dt = time.perf_counter() - t0
            if k > 0:                   # frame 0 is the bootstrap keyframe (STATE_FIRST_FRAME), not a tracked frame
                t_total += dt
                tracked += int(st.quality != 2)
        trk.close()
        out[i] = (tracked, t_total)

    if threads == 1:
        run(0)
    else:
        ts = [threading.Thread(target=run, args=(i,)) for i in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    tracked = sum(o[0] for o in out)
    wall = max(o[1] for o in out)       # the slowest thread's summed HandleFrame time = the job's wall time
    return tracked / wall, tracked, wall


def hw_queues_leg(args, n_queues):
    """The resident leg once more in a CHILD process whose HIP runtime multiplexes the farm's streams onto `n_queues` hardware queues
    (GPU_MAX_HW_QUEUES; the runtime's default is 4, the farm has 16 groups = 16 streams).  With 16 queues up to sixteen kernels share the chip:
    the farm is 3-5 % faster on most boxes of the pool (profiles/r06/ab_round6.txt) and every dispatch lasts longer, so the per-kernel durations the
    `roofline` block divides by stop describing a kernel — which is why `value` and `roofline` stay on the runtime's default and this figure is
    reported beside them (value_hw_queues_16).  Started before this process touches the GPU, like the latency legs; skipped under a profiler."""
    import subprocess
    preload = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB"))
    if "rocprof" in preload.lower():
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--texture", args.texture, "--distortion", args.distortion,
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--seqs", str(args.user_seqs), "--groups", str(args.groups), "--threads", str(args.threads),
           "--workers", str(args.workers), "--fibers", str(args.fibers), "--cpu-frames", "0", "--host-steps", "0", "--sustained-frames", "0",
           "--latency-frames", "0", "--lost-mix-steps", "0", "--hw-queues-leg", "0"]
    try:
        r = subprocess.run(cmd, env=dict(os.environ, GPU_MAX_HW_QUEUES=str(n_queues)), capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            sys.stderr.write("bench.py: hardware-queues leg failed (rc %d): %s\n" % (r.returncode, r.stderr[-400:]))
            return None
        d = json.loads(line[-1])
    except (subprocess.SubprocessError, ValueError) as e:
        sys.stderr.write("bench.py: hardware-queues leg failed: %s\n" % e)
        return None
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "gpu_max_hw_queues": n_queues,
            "note": "the same resident leg (same sequences, steps and farm shape) in a child process with GPU_MAX_HW_QUEUES=%d: the farm's 16 streams on "
                    "16 hardware queues instead of the runtime's 4; not `value`: with sixteen kernels sharing the chip a dispatch's duration no longer "
                    "describes its kernel, so the roofline block is measured on the runtime's default" % n_queues}


def latency_legs(wl, texture, n_frames, mapper=False, dist=None):
    """The reference's own shape of use (main.cc:126-159): ONE camera through SDVL::HandleFrame, frame after frame — and 16 cameras,
    one host thread + one HIP stream each — measured by host/track_sequence, the C++ loop against the reference's API, as child
    processes BEFORE this process touches the GPU (they have the chip to themselves).  The window is main.cc:136-138: the
    HandleFrame call alone; the image is on the device by then (Camera::UndistortImage, main.cc:133, is outside, as in the
    reference).  Returns the `latency` block or None."""
    import subprocess
    # Under a profiler the preloaded tool library has initialised the GPU in THIS process already: starting children from it is the
    # forbidden exec-after-GPU-init, and their lone-camera kernels would land in the farm's per-kernel averages
    preload = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB"))
    if "rocprof" in preload.lower():
        sys.stderr.write("bench.py: a rocprofiler library is preloaded; the latency legs (child processes) are skipped\n")
        return None
    exe = os.path.join(ROOT, "slam-sdvl_amd", "host", "track_sequence")
    if not os.path.exists(exe):
        sys.stderr.write("bench.py: %s is not built; latency legs skipped\n" % exe)
        return None
    base = [exe, "--synthetic", str(n_frames), "--texture", texture, "--size", str(wl["w"]), str(wl["h"]), "--cam"] + [repr(float(c)) for c in wl["cam"]] + [
        "--seed", str(wl.get("seed0", 20260001)), "--prerender", "--quiet", "--json"]
    for k, v in wl["over"].items():
        base += ["--set", k, repr(float(v))]
    if mapper:
        base.append("--mapper")
    if dist is not None:   # frames rendered through the lens; track_sequence calls camera.UndistortImage before every HandleFrame (outside the window, main.cc:133)
        base += ["--dist"] + [repr(float(d)) for d in dist]
    out = {}
    for name, n, extra in (("b1", 1, []), ("b1_lookahead", 1, ["--lookahead"]), ("b16", 16, []), ("b16_batched", 16, ["--batch"])):
        try:
            r = subprocess.run(base + ["--trackers", str(n)] + extra, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                sys.stderr.write("bench.py: latency leg %s failed (rc %d): %s\n" % (name, r.returncode, r.stderr[-400:]))
                continue
            out[name] = json.loads(line[-1])
        except (subprocess.SubprocessError, ValueError) as e:
            sys.stderr.write("bench.py: latency leg %s failed: %s\n" % (name, e))
    if not out:
        return None
    blk = {"api": "SDVL::HandleFrame, one call per frame and camera (host/track_sequence.cc = the loop of main.cc:126-159); window = the call alone "
                  "(main.cc:136-138), the frame is in HBM by then (Camera::UndistortImage, main.cc:133, outside the window as in the reference)",
           "frames_per_sequence": n_frames}
    if "b1" in out:
        blk.update({"b1_frames_per_s": out["b1"]["frames_per_s"], "b1_ms_per_frame_p50": out["b1"]["ms_per_frame_p50"],
                    "b1_ms_per_frame_p95": out["b1"]["ms_per_frame_p95"], "b1_tracked": out["b1"]["tracked"]})
    if "b1_lookahead" in out:
        blk.update({"b1_lookahead_frames_per_s": out["b1_lookahead"]["frames_per_s"], "b1_lookahead_ms_per_frame_p50": out["b1_lookahead"]["ms_per_frame_p50"],
                    "b1_lookahead_note": "the same camera when the caller reads a sequence (main.cc's Video.type 1) and names the next frame one call ahead "
                                         "(SDVL::SetNextImage, an addition to the reference's API): its pyramid and corners are built behind the current chain"})
    if "b16" in out:
        blk.update({"b16_frames_per_s": out["b16"]["frames_per_s"], "b16_frames_per_s_per_camera": out["b16"]["frames_per_s_per_camera"],
                    "b16_ms_per_frame_p50": out["b16"]["ms_per_frame_p50"], "b16_ms_per_frame_p95": out["b16"]["ms_per_frame_p95"],
                    "b16_tracked": out["b16"]["tracked"],
                    "b16_note": "16 cameras = 16 host threads, each its own Device (HIP stream) and SDVL::HandleFrame loop"})
    if "b16_batched" in out:
        blk.update({"b16_batched_frames_per_s": out["b16_batched"]["frames_per_s"], "b16_batched_ms_per_step_p50": out["b16_batched"]["ms_per_frame_p50"],
                    "b16_batched_ms_per_step_p95": out["b16_batched"]["ms_per_frame_p95"],
                    "b16_batched_note": "the same 16 cameras through ONE sdvl::SDVLBatch::HandleFrames call per frame (one thread, one stream, one launch "
                                        "per kernel for all 16): what a caller with several cameras should use"})
    return blk


def host_memory_available():
    """bytes this process can still take: MemAvailable, capped by the cgroup's limit minus its current use (None: unknown)"""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
                break
    except (OSError, ValueError):
        pass
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        if mx != "max":
            left = int(mx) - int(open("/sys/fs/cgroup/memory.current").read().strip())
            avail = left if avail is None else min(avail, left)
    except (OSError, ValueError):
        pass
    return avail


def visible_gpus():
    """GPUs this process could use, counted WITHOUT initialising HIP/HSA in it (a process that has touched the GPU must never
    start rank children): the KFD topology in sysfs lists every node, GPUs are the ones with SIMDs; ROCR_/HIP_/CUDA_VISIBLE_DEVICES
    narrow the set.  Returns None when sysfs has no KFD topology (then a throw-away child asks the runtime)."""
    import glob
    import subprocess
    n = None
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if nodes:
        n = 0
        for f in nodes:
            try:
                props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
                n += 1 if int(props.get("simd_count", "0")) > 0 else 0
            except (OSError, ValueError):
                pass
    if n is None or n == 0:
        try:   # the runtime's answer, from a child that exits again: this process stays clean
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
            n = int(out.stdout.strip().splitlines()[-1])
        except (subprocess.SubprocessError, ValueError, IndexError):
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, dry=False):
    """`python bench.py --gpus N` without a distributed launcher: start the N ranks as child processes, one per GPU.  This process
    never touches the GPU — not even to count devices (visible_gpus reads sysfs) — because a process that has initialised HIP must
    not start workers.  Every rank gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like under torch.distributed.run and an equal share
    of the CPUs (SDVL_BENCH_CPU_SHARE; inside the rank the share is bound to the NUMA node of its GPU).  Rank 0 prints the JSON
    line.  The rendezvous port is found by bind-and-close, so another process can take it before rank 0 listens: when the ranks die
    of exactly that (EADDRINUSE in rank 0's stderr) they are started again on a fresh port, up to three times.
    Returns the exit code: non-zero if the box has fewer GPUs than ranks or if any rank fails."""
    import socket
    import subprocess
    import tempfile
    if not dry:
        have = visible_gpus()
        if have is not None and have < n:
            sys.stderr.write("bench.py: --gpus %d but %d GPU(s) are visible on this box - refusing to measure fewer GPUs than asked for\n" % (n, have))
            return 2
        # have is None: nothing could be counted here; every rank checks its own GPU and fails loudly ("rank r wants GPU r ...")
    share = max(1, effective_cpus() // n)
    rc = 0
    for attempt in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        err0 = tempfile.TemporaryFile()
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), SDVL_BENCH_CPU_SHARE=str(share), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stderr=err0 if r == 0 else None))
        rc = 0
        try:
            pending = set(range(n))
            while pending:
                for r in sorted(pending):
                    code = procs[r].poll()
                    if code is None:
                        continue
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 1
                        sys.stderr.write("bench.py: rank %d exited with %d - stopping the other ranks\n" % (r, code))
                        for q in pending:
                            procs[q].terminate()     # exactly the children started above, by handle
                time.sleep(0.05)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        err0.seek(0)
        text = err0.read().decode(errors="replace")
        err0.close()
        sys.stderr.write(text)
        if rc != 0 and ("EADDRINUSE" in text or "ddress already in use" in text) and attempt < 2:
            sys.stderr.write("bench.py: the rendezvous port %d was taken before rank 0 could listen - starting the ranks again\n" % port)
            continue
        break
    return rc


def seq_seed_fn(shard, wl, per_gpu, world):
    """global sequence index -> texture seed.  S-B: the seed names the chunk (20260010..13; shard.chunk_for_sequence: rank r tracks the
    chunks {c : c mod G = r}, chunk r mod 4 with more ranks than chunks), a rank's sequences differ by their twists; S-C: 20260100 + the
    global index (= 20260100 + rank*64 + i at SURVEY's 64 sequences per GPU); S-A: 20260001 + the global index."""
    if "chunks" in wl:
        return lambda g: wl["seed0"] + shard.chunk_for_sequence(g, per_gpu, world, wl["chunks"])
    if "seed0" in wl:
        return lambda g: wl["seed0"] + g
    return shard.sequence_seed


def host_budget_per_rank(world):
    """host memory ONE rank may plan with (pinned pools, keyframe objects): what the node has free, divided by the ranks that share it"""
    free = host_memory_available()
    return None if free is None else free / max(1, world)


def dry_rank(args, rank, world):
    """SDVL_BENCH_DRY=1: the N > 1 plumbing of this file without a GPU - sharding, gloo rendezvous, barrier, the one
    reduction (shard.reduce_throughput) and rank 0's JSON line - so that the launcher is testable on CPU."""
    shard = importlib.import_module("slam-sdvl_amd.shard")
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, K = args.seqs, args.steps
    my = shard.sequences_for_rank(rank, world, B)
    # what this rank would render: the texture seeds of its sequences (workload-dependent) and the host memory it may plan with
    seed_of = seq_seed_fn(shard, WORKLOADS[args.workload], B, world)
    mine = {"rank": rank, "first_sequence": my[0], "seeds": sorted(set(seed_of(g) for g in my)), "seed_of_first": seed_of(my[0]), "seed_of_last": seed_of(my[-1]),
            "host_budget_bytes": host_budget_per_rank(world)}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.02 * (1 + rank))                      # ranks finish at different times: the job's time is the slowest rank's
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tracked_all, elapsed_max = shard.reduce_throughput(len(my) * K, elapsed, dist, "cpu")
    # the host-fed leg runs on every rank: the same reductions as the real run (sum of tracked frames, max of the ranks' times, min / max
    # of the ranks' own link rates), with a sleep in place of the farm
    Kh = args.host_steps if args.host_steps >= 0 else (16 if world == 1 else 8)
    host_fed = link = None
    if Kh > 0:
        frame_bytes = 640 * 480
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (1 + rank))
        own_h = time.perf_counter() - t0                  # this rank's own time: its link rate; the job's time ends behind the barrier
        if world > 1:
            dist.barrier()
        elapsed_h = time.perf_counter() - t0
        th_all, eh_max = shard.reduce_throughput(len(my) * Kh, elapsed_h, dist, "cpu")
        lo, hi = shard.reduce_min_max(len(my) * Kh * frame_bytes / own_h / 1e9, dist, "cpu")
        agg = world * len(my) * Kh * frame_bytes / eh_max / 1e9
        host_fed = {"value": round(th_all / eh_max, 2), "unit": "frames/s", "steps": Kh, "pcie_h2d_gb_per_s": round(agg, 6),
                    "pcie_h2d_gb_per_s_per_gpu": {"min": round(lo, 6), "max": round(hi, 6)}, "tracked": th_all}
        link = {"bound": "pcie_h2d", "achieved": round(agg / world, 6), "peak": PCIE_LONE_STREAM_GBS, "unit": "GB/s per GPU",
                "frac": round(agg / world / PCIE_LONE_STREAM_GBS, 8), "per_gpu_min_max": host_fed["pcie_h2d_gb_per_s_per_gpu"]}
    if rank == 0:
        print(json.dumps({"metric": "tracked frames/sec (dry run, no GPU)", "value": round(tracked_all / elapsed_max, 2), "unit": "frames/s",
                          "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(elapsed_max / K * 1e3, 3), "dry": True,
                          "tracked": tracked_all, "cpu_share": int(os.environ.get("SDVL_BENCH_CPU_SHARE", "0")),
                          "sequences": [my[0], my[-1]], "scaling": "weak", "workload": args.workload, "per_rank": per_rank,
                          "host_memory_available_bytes": host_memory_available(),
                          "value_host_fed": host_fed["value"] if host_fed else None, "host_fed": host_fed, "roofline": {"link": link}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="S-A",
                    help="S-A: the metric configuration (640x480); S-B: EuRoC geometry 752x480, four chunks; S-C: 1280x960, 4000 features")
    ap.add_argument("--texture", choices=sorted(TEXTURES), default="plane",
                    help="plane: value noise, a FAST corner on every second tested pixel (rounds 1-4); camera: piecewise-smooth shading + soft-edged "
                         "shapes, 2-5 k FAST keypoints per 640x480 frame (a few % of the pixels, like a camera frame)")
    ap.add_argument("--distortion", choices=sorted(DISTORTIONS), default="none",
                    help="the camera's lens (the cfg file's Camera.d1..d5): the synthetic frames are rendered THROUGH that lens and every leg undistorts "
                         "them first, as main.cc:133 does (Camera::UndistortImage: a remap from a per-camera map cached on the device)")
    ap.add_argument("--latency-frames", type=int, default=-1,
                    help="frames of the LATENCY legs (one sequence alone and 16 sequences through the C++ SDVL::HandleFrame loop, host/track_sequence): "
                         "0 = skip; default 300 for S-A / S-B on one GPU")
    ap.add_argument("--seqs", type=int, default=int(os.environ.get("SDVL_BENCH_SEQS", "0")), help="independent sequences per GPU (0 = the workload's)")
    ap.add_argument("--groups", type=int, default=0, help="groups per GPU, each = host thread + HIP stream (0 = auto)")
    ap.add_argument("--threads", type=int, default=0, help="extra host threads inside a group for per-sequence stages (0 = 1)")
    ap.add_argument("--workers", type=int, default=0, help="host threads that execute group-steps (0 = auto)")
    ap.add_argument("--fibers", type=int, default=1,
                    help="group-steps a worker thread interleaves, switching at GPU waits (1 = one at a time)")
    ap.add_argument("--mapper", action="store_true",
                    help="run the reference's mapper (map.cc, sequential mode) inside every step instead of the plane map stub")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="frames of the CPU baseline sample (0 = skip, -1 = the workload's)")
    ap.add_argument("--sustained-frames", type=int, default=-1,
                    help="frames per sequence of the third, SUSTAINED leg: fresh trackers run long sequences (two lengths of SURVEY §8d's S-A: 600 frames) "
                         "host-fed from a pool of distinct sequences, with SDVL.max_keyframes = --sustained-max-keyframes so that the map — and with "
                         "it HBM and host memory — stops growing; reported as value_sustained with the memory in use at the middle and at the end; "
                         "0 = skip; default 600 for S-A on one GPU")
    ap.add_argument("--sustained-max-keyframes", type=int, default=48,
                    help="SDVL.max_keyframes of the sustained leg (the reference's cfg files say 1000, its default is 100: config.cc:63); 0 = the workload's")
    ap.add_argument("--lost-mix-steps", type=int, default=-1,
                    help="steps of the LOST-MIX leg (VERDICT r05 #1): a fresh farm on a pool of distinct sequences resident in HBM runs 40 undisturbed steps "
                         "(its own baseline), then this many steps in which 5 %% of the trackers are blinded (a featureless frame) for 5 frames every 50 "
                         "frames: they go TRACKING_BAD x3, relocalise over their keyframes inside the tabled step and rejoin; reported as value_lost_mix "
                         "and lost_mix; 0 = skip; default 100 for S-A / S-B on one GPU")
    ap.add_argument("--hw-queues-leg", type=int, default=-1,
                    help="hardware queues of the extra HW-QUEUES leg (the resident leg again in a child process with GPU_MAX_HW_QUEUES set to this; "
                         "reported as value_hw_queues_16): 0 = skip; default 16 for S-A / S-B on one GPU when the caller has not set GPU_MAX_HW_QUEUES")
    ap.add_argument("--host-steps", type=int, default=-1,
                    help="steps of the second, HOST-FED leg (frames in pinned host memory, uploaded inside the step, as SDVL::HandleFrame(const cv::Mat&) "
                         "receives them): reported as value_host_fed next to the HBM-resident value; 0 = skip; "
                         "default 16 on one GPU, 8 per rank on several (10 GB of pinned host memory per rank, NUMA-local to its GPU)")
    args = ap.parse_args()
    global W_IMG, H_IMG, TUM_CAM, FEATS_LABEL, ORACLE_PARAMS, TEXTURE, SEQ_SEED, DISTORTION
    DISTORTION = DISTORTIONS[args.distortion]
    wl = WORKLOADS[args.workload]
    W_IMG, H_IMG, TUM_CAM, FEATS_LABEL = wl["w"], wl["h"], np.array(wl["cam"]), wl["label"]
    ORACLE_PARAMS = {k.split(".")[1]: v for k, v in wl["over"].items()}
    TEXTURE = TEXTURES[args.texture]
    global PROFILE_TAG
    PROFILE_TAG = "%s_%s" % (args.workload.lower().replace("-", ""), args.texture) + ("_" + args.distortion if args.distortion != "none" else "")
    args.user_seqs = args.seqs   # what the caller asked for (0 = the workload's choices): handed on to the hardware-queues leg's child
    seq_choices = [args.seqs] if args.seqs > 0 else list(wl["seqs"])
    args.seqs = seq_choices[0]
    if args.cpu_frames < 0:
        args.cpu_frames = wl["cpu_frames"]

    dry = bool(os.environ.get("SDVL_BENCH_DRY"))   # test hook (tests/test_bench_launcher.py): ranks, barrier and reduction without a GPU
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # launched directly with --gpus N: this process only starts the N ranks (it never touches the GPU itself)
            raise SystemExit(launch_ranks(args.gpus, dry))
        world, rank, local_rank = 1, 0, 0
    else:
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d: one rank per GPU (torch.distributed.run --nproc-per-node N bench.py "
                             "--gpus N, or plain `python bench.py --gpus N`, which starts the ranks itself)" % (world, args.gpus))
    distributed = world > 1
    if args.host_steps < 0:
        # the host-fed leg runs on EVERY rank (round 4): the drop-in figure is the one whose ranks share something — the host's DRAM and
        # PCIe roots.  8 steps per rank on several GPUs (10 GB of pinned memory each, allocated after the NUMA binding), 16 on one
        args.host_steps = 16 if world == 1 else 8
    if args.sustained_frames < 0:
        args.sustained_frames = 600 if (world == 1 and args.workload == "S-A" and not args.mapper and not dry) else 0
    if args.lost_mix_steps < 0:
        args.lost_mix_steps = 100 if (world == 1 and args.workload in ("S-A", "S-B") and not args.mapper and not dry) else 0
    dist = None
    if dry:
        return dry_rank(args, rank, world)
    if args.latency_frames < 0:
        args.latency_frames = 300 if (world == 1 and args.workload in ("S-A", "S-B")) else 0
    latency = None
    if world == 1 and args.latency_frames > 2:
        latency = latency_legs(wl, args.texture, args.latency_frames, args.mapper, DISTORTION)   # child processes, before this one touches the GPU
    if args.hw_queues_leg < 0:
        args.hw_queues_leg = 16 if (world == 1 and args.workload in ("S-A", "S-B") and not args.mapper and "GPU_MAX_HW_QUEUES" not in os.environ) else 0
    hw_queues = hw_queues_leg(args, args.hw_queues_leg) if (world == 1 and args.hw_queues_leg > 0) else None
    # (GPU_MAX_HW_QUEUES: the farm's 16 groups are 16 HIP streams and the runtime multiplexes a process's streams onto 4 hardware queues by
    #  default.  16 queues gave +5 % over three alternating pairs on one box (profiles/r05/ab_round5.txt) and nothing on the next
    #  (ab_queues_by_steps.txt: within the run-to-run spread at 10 / 20 / 40 / 80 steps), while every dispatch gets 2-10 x longer because
    #  sixteen kernels share the chip - so the runtime's default stays; a caller's own setting is reported in config.gpu_max_hw_queues.)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d wants GPU %d but only %d are visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("slam-sdvl_amd")
    trk = importlib.import_module("slam-sdvl_amd.tracker")
    shard = importlib.import_module("slam-sdvl_amd.shard")
    numa_node = None if os.environ.get("SDVL_NO_NUMA_BIND") else trk.bind_to_gpu_numa_node(local_rank)
    trk.configure(dict(trk.TUM_OVERRIDES, **wl["over"]))
    ncpu = effective_cpus()
    cpus_rank = int(os.environ.get("SDVL_BENCH_CPU_SHARE", "0")) or max(1, ncpu // max(1, world))
    K, Wm = args.steps, args.warmup
    fibers = args.fibers
    n_frames = 1 + Wm + K                     # bootstrap keyframe + warmup + timed
    frame_bytes = W_IMG * H_IMG
    lib0 = pkg.load_library()
    lib0.sdvl_frame_footprint_cap.restype = C.c_int64
    lib0.sdvl_detect_scratch_bytes.restype = C.c_int64
    n_feat_cfg = int(wl["over"].get("SDVL.num_features", 1000))
    corner_cap = min(6144, max(1024, (2 * n_feat_cfg + 63) // 64 * 64))      # Device::CornerCap (host/sdvl_host.cc)
    footprint = lib0.sdvl_frame_footprint_cap(W_IMG, H_IMG, 5, corner_cap)   # what a frame keeps for as long as it lives
    det_params = pkg.default_detect_params()
    scratch_per_frame = lib0.sdvl_detect_scratch_bytes(W_IMG, H_IMG, C.byref(det_params))   # per frame of a group's batch, per context
    free_b, total_b = torch.cuda.mem_get_info()
    B = G = Bg = reserve_frames = need = None
    for cand in seq_choices:
        # one group = one HIP stream + one host-side step at a time; workers = host threads.  With a full CPU share (16 per
        # GPU on the MI355X boxes) one group per worker is best; with fewer CPUs every worker interleaves two groups (fibers),
        # which hides their GPU waits and is worth ~10 % there.
        B = cand
        fibers = args.fibers
        if args.groups:
            G = args.groups
        else:
            workers_auto = max(1, min(cpus_rank, 16) // max(1, args.threads))   # a group's helper threads come out of the same share
            # (with the reference's mapper in the step the host is the limiter and a step has four more GPU waits — candidate,
            #  connection and seed searches, FilterCorners: two groups per worker keep the core busy across them, +5 %)
            if fibers <= 1 and (cpus_rank < 12 or args.mapper) and not args.workers:
                fibers = 2
            G = max(1, min(B // 8 if B >= 8 else 1, workers_auto * max(1, fibers)))
            if args.workload == "S-C":   # 4x the pixels and features per frame: 64 frames per launch fill the chip, more streams only
                G = max(1, min(G, max(4, B // 64)))   # stretch the latency-bound kernels (512 sequences: 8 groups 33-40k, 16 groups 31k, 4 groups 28k frames/s;
                                                      # SURVEY's 64 sequences: 4 groups of 16 23.6k, 8 of 8 17k, 2 of 32 15k, 1 of 64 9k: the chain's latency per launch hardly depends on its size)
        while B % G:
            G -= 1
        Bg = B // G
        steps_all = Wm + K + max(0, args.host_steps)
        reserve_frames = Bg * (4 + (steps_all + 3) // 4)   # keyframe budget: S-A turns about one frame in five into a keyframe
        need = B * n_frames * frame_bytes + G * reserve_frames * footprint + B * 2 * footprint + B * scratch_per_frame
        if need <= 0.85 * free_b:
            break
    if need > 0.9 * free_b:
        raise SystemExit("bench.py: %d sequences x %d steps need %.0f GB of HBM (input frames stay resident, keyframes keep their frame); "
                         "%.0f GB are free - use fewer --steps or --seqs" % (B, K, need / 1e9, free_b / 1e9))
    args.seqs = B
    threads = args.threads or 1
    trk.set_mapper(args.mapper)
    farm = trk.TrackerFarm(local_rank, G, Bg, W_IMG, H_IMG, TUM_CAM, host_threads_per_group=threads)
    farm.set_fibers(fibers)
    if DISTORTION is not None:
        farm.set_distortion(DISTORTION)
    ctxs = [CtxView(pkg, farm.ctx_handle(g)) for g in range(G)]
    ctx = ctxs[0]

    my_seqs = shard.sequences_for_rank(rank, world, B)   # independent sequences: no data-path collective
    SEQ_SEED = seq_seed_fn(shard, wl, B, world)
    buf = ctx.malloc(B * n_frames * frame_bytes)
    for k in range(n_frames):                 # frame-major layout: step k reads B consecutive frames
        views = [make_view(pkg, se3_exp(shard.sequence_twist(g) * k), SEQ_SEED(g), k) for g in my_seqs]
        ctx.render(views, buf + k * B * frame_bytes)
    ptrs = (buf + (np.arange(n_frames, dtype=np.uint64)[:, None] * B + np.arange(B, dtype=np.uint64)[None, :]) * frame_bytes).astype(np.uint64)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # CPU baseline first (rank 0, one core, bounded sample): it is independent of the GPU run, and on a freshly started box
    # it also keeps the timed region away from the start-up activity of the container (measured: the first seconds of a
    # fresh box cost the 16 host threads up to 40 % of their CPU share)
    cpu, cpu_sample, cpu_one, cpu_all = None, None, None, None
    if rank == 0 and world == 1 and args.cpu_frames > 0:   # at N = 1 only: with more ranks the host cores belong to their farms
        n_cpu = args.cpu_frames
        cbuf = ctx.malloc(n_cpu * frame_bytes)
        views = [make_view(pkg, se3_exp(shard.sequence_twist(0) * k), SEQ_SEED(0), k) for k in range(n_cpu)]
        ctx.render(views, cbuf)
        host = ctx.download(cbuf, n_cpu * frame_bytes).reshape(n_cpu, H_IMG, W_IMG)
        cpu_sample = [host[k].copy() for k in range(n_cpu)]
        if DISTORTION is not None:   # main.cc:133: the reference undistorts before the window it times (the oracle's cv::undistort restatement)
            import oraclelib as _ol
            _orc = _ol.Oracle()
            cpu_sample = [_orc.undistort(im, TUM_CAM, np.array(DISTORTION)) for im in cpu_sample]
        cpu_one = cpu_baseline(cpu_sample, args.mapper, 1)
        del host
        if os.environ.get("SDVL_BENCH_CPU_ORDER", "after") == "before":
            cpu_all = cpu_baseline(cpu_sample, args.mapper, max(1, min(ncpu, 16)))
    # (the all-core leg runs AFTER the timed region: 16 threads at full load right before it cost the GPU run ~5-10 % -
    #  151 k against 170 k on the same box - whatever the host does to a process that has just used its whole CPU share)

    # FAST keypoints per frame (the byte model of the selection kernels reads every per-cell list once): counted on four frames
    # of this workload before anything is timed
    n_kp_measured = 10000
    try:
        probe = pkg.Context(local_rank)
        counts = []
        for k in (0, 1):
            for q in (0, min(B - 1, 7)):
                img = ctx.download(int(ptrs[k, q]), frame_bytes).reshape(H_IMG, W_IMG)
                pf_ = probe.frame(img)
                got, _ = probe.fast_cells([pf_], det_params)
                counts.append(len(got[0][0]))
                pf_.close()
        probe.close()
        n_kp_measured = int(round(sum(counts) / len(counts)))
    except Exception as e:   # the probe is a measurement aid; the byte model then keeps its round-2 constant
        sys.stderr.write("bench.py: keypoint-count probe failed (%s); using %d\n" % (e, n_kp_measured))

    workers = args.workers or max(1, G // max(1, fibers))
    farm.reserve(reserve_frames)              # every keyframe keeps its HBM frame: no hipMalloc inside the run
    # Kernel timing (round 4).  A dispatch that carries start / stop events costs the host ~12 us instead of ~4 and, with EVERY dispatch of
    # 16 streams carrying them, the whole farm ~10 % of its throughput (345 k against 385 k tracked frames/s on one box).  So: every
    # dispatch is timed during the WARM-UP steps (kernel_ms_per_step, and which kernel has the most dispatch time); in the TIMED region
    # only that kernel's launches carry events — roofline.avg_launch_us is measured live over the timed region, as the contract asks,
    # without taxing the number it stands beside.  SDVL_BENCH_TIME_ALL=1: events on everything in the timed region too (rounds 1-3);
    # SDVL_BENCH_NO_KERNEL_TIMING=1: none at all.
    no_timing = bool(os.environ.get("SDVL_BENCH_NO_KERNEL_TIMING"))
    time_all = bool(os.environ.get("SDVL_BENCH_TIME_ALL")) or Wm < 3   # (too few warm-up steps to rank the kernels fairly: time everything)
    farm.run(ptrs[:1], workers)               # bootstrap keyframe (untimed)
    warm_timers = {}
    Wt = 0                                    # warm-up steps with every dispatch timed: the LAST ones — the first frames behind the
    if Wm > 0:                                # bootstrap run without a motion model and their alignment iterates three times as long
        Wt = max(1, min(3, Wm - 1))
        if Wm > Wt:
            farm.run(ptrs[1:1 + Wm - Wt], workers)
        for c in ctxs:
            c.timing(not no_timing)
        farm.run(ptrs[1 + Wm - Wt:1 + Wm], workers)   # (outside the timed region)
        for c in ctxs:
            for name, (ms, n) in c.timing_get().items():
                a = warm_timers.get(name, (0.0, 0))
                warm_timers[name] = (a[0] + ms, a[1] + n)
    dom_name = max(warm_timers.items(), key=lambda kv: kv[1][0])[0] if warm_timers else None
    farm.stage_times(reset=True)
    for c in ctxs:
        c.timing(not no_timing, None if (time_all or not dom_name) else dom_name)
    stats_buf = farm.alloc_stats(K)
    import resource
    def throttled():
        try:
            kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
            return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
        except (OSError, ValueError):
            return 0, 0
    thr0 = throttled()
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    flt0 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    barrier()
    t0 = time.perf_counter()
    stats = farm.run(ptrs[1 + Wm:], workers, stats_buf)  # the K timed steps: group-steps are scheduled onto the worker threads
    barrier()
    elapsed = time.perf_counter() - t0
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    flt1 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    thr1 = throttled()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    sys.stderr.write("host CPU in the timed region: %.2f s user + %.2f s system over %.2f s wall = %.1f CPUs busy (%d usable)\n" %
                     (ru1.ru_utime - ru0.ru_utime, ru1.ru_stime - ru0.ru_stime, elapsed, cpu_s / elapsed, ncpu))
    sys.stderr.write("host memory: max RSS %.0f -> %.0f MB, minor page faults in the timed region: %d; CPU quota throttling in the timed region: "
                     "%d periods, %.1f ms\n" % (rss0 / 1024, rss1 / 1024, flt1 - flt0, thr1[0] - thr0[0], (thr1[1] - thr0[1]) / 1e3))
    timers = {}
    for c in ctxs:
        for name, (ms, n) in c.timing_get().items():
            a = timers.get(name, (0.0, 0))
            timers[name] = (a[0] + ms, a[1] + n)
        c.timing(False)
    stage_s, stage_n = farm.stage_times()
    tracked = 0
    n_c = n_s = n_f = n_ia = n_lk = n_m = n_kf = 0
    for st in stats:
        n_m += st.matches
        n_kf += st.keyframe
        tracked += int(st.quality != 2)
        n_c += st.n_corners
        n_s += st.search_requests
        n_f += st.align_features
        n_ia += st.align_iters
        n_lk += st.lk_iters

    tracked_all, elapsed_max = shard.reduce_throughput(tracked, elapsed, dist if distributed else None, "cuda")

    # ---- second leg, HOST-FED: the same trackers continue, but their next frames come from pinned HOST memory and every group
    # uploads them on its own stream inside the step — what SDVL::HandleFrame(const cv::Mat&) receives (sdvl.cc:55-59).  This is
    # the end-to-end drop-in rate; `value` above is the contract's number (inputs resident when the timed region starts).
    host_fed = None
    Kh = max(0, args.host_steps)
    if Kh > 0:
        host_free = host_budget_per_rank(world)        # every rank of the node pins its own pool out of the same memory
        while Kh > 4 and host_free is not None and B * Kh * frame_bytes + 12e9 > 0.85 * host_free:
            Kh //= 2                                  # fewer host-fed steps rather than an out-of-memory kill
        if distributed:                               # all ranks run the same number of steps (the smallest any of them can afford)
            kh_t = torch.tensor([Kh], dtype=torch.int64, device="cuda")
            dist.all_reduce(kh_t, op=dist.ReduceOp.MIN)
            Kh = int(kh_t[0])
        try:
            hbuf = torch.empty(B * Kh * frame_bytes, dtype=torch.uint8, pin_memory=True)
        except RuntimeError as e:
            hbuf = None
            sys.stderr.write("bench.py: host-fed leg skipped, cannot pin %.1f GB of host memory (%s)\n" % (B * Kh * frame_bytes / 1e9, e))
            if distributed:                           # the other ranks are about to enter the leg's barrier: fail the job loudly instead of hanging it
                raise SystemExit("bench.py: rank %d cannot pin its host-fed pool; rerun with --host-steps 0 or fewer steps" % rank)
        if hbuf is not None:
            lib_hip = pkg.load_library()
            for k in range(Kh):                       # the frames that follow the resident leg's last one, rendered into the (now free) input area
                ka = n_frames + k
                views = [make_view(pkg, se3_exp(shard.sequence_twist(g) * ka), SEQ_SEED(g), ka) for g in my_seqs]
                ctx.render(views, buf)
                ctx.check(lib_hip.sdvl_device_download(ctx.h, C.c_void_p(buf), C.c_int64(B * frame_bytes), C.c_void_p(hbuf.data_ptr() + k * B * frame_bytes)))
            hptrs = (hbuf.data_ptr() + (np.arange(Kh, dtype=np.uint64)[:, None] * B + np.arange(B, dtype=np.uint64)[None, :]) * frame_bytes).astype(np.uint64)
            farm.set_input_ring(not os.environ.get("SDVL_BENCH_NO_INPUT_RING"))
            farm.set_host_input(True)     # allocates the groups' input rings
            hstats_buf = farm.alloc_stats(Kh)
            farm.stage_times(reset=True)
            for c in ctxs:
                c.timing(not os.environ.get("SDVL_BENCH_NO_KERNEL_TIMING"))
            barrier()
            thr_h0, ru_h0 = throttled(), resource.getrusage(resource.RUSAGE_SELF)
            t0 = time.perf_counter()
            hstats = farm.run(hptrs, workers, hstats_buf)
            own_h = time.perf_counter() - t0          # this rank's own time (its link rate); the job's time ends behind the barrier
            barrier()
            elapsed_h = time.perf_counter() - t0
            thr_h1, ru_h1 = throttled(), resource.getrusage(resource.RUSAGE_SELF)
            host_cpu_h = {"cpus_busy": round(((ru_h1.ru_utime - ru_h0.ru_utime) + (ru_h1.ru_stime - ru_h0.ru_stime)) / elapsed_h, 2),
                          "system_share": round((ru_h1.ru_stime - ru_h0.ru_stime) / max(1e-9, (ru_h1.ru_utime - ru_h0.ru_utime) + (ru_h1.ru_stime - ru_h0.ru_stime)), 3),
                          "usable": ncpu, "quota_throttled_ms": round((thr_h1[1] - thr_h0[1]) / 1e3, 1), "minor_faults": ru_h1.ru_minflt - ru_h0.ru_minflt}
            farm.set_host_input(False)
            h_timers = {}
            for c in ctxs:
                for name, (ms, n_l) in c.timing_get().items():
                    a = h_timers.get(name, (0.0, 0))
                    h_timers[name] = (a[0] + ms, a[1] + n_l)
                c.timing(False)
            h_stage_s, h_stage_n = farm.stage_times()
            feed_call_s, feed_wait_s, feed_calls, work_wait_s, late_acq, feed_thr_s = farm.feed_stats()
            tracked_h = sum(int(st.quality != 2) for st in hstats)
            th_all, eh_max = shard.reduce_throughput(tracked_h, elapsed_h, dist if distributed else None, "cuda")
            link_lo, link_hi = shard.reduce_min_max(B * Kh * frame_bytes / own_h / 1e9, dist if distributed else None, "cuda")
            host_fed = {"value": round(th_all / eh_max, 2), "unit": "frames/s", "steps": Kh, "ms_per_step": round(eh_max / Kh * 1e3, 3),
                        "pcie_h2d_gb_per_s": round(world * B * Kh * frame_bytes / eh_max / 1e9, 2),
                        "pcie_h2d_gb_per_s_per_gpu": {"min": round(link_lo, 2), "max": round(link_hi, 2)},
                        "host_cpu": host_cpu_h,
                        "feeder": {"transfers": feed_calls, "s_in_transfer_calls": round(feed_call_s, 4), "s_waiting_for_a_free_slot": round(feed_wait_s, 4),
                                   "s_waiting_for_its_own_transfers": round(feed_thr_s, 4), "wall_s": round(elapsed_h, 4),
                                   "group_steps_begun_before_their_images_arrived": late_acq, "group_steps": feed_calls,
                                   "worker_s_waiting_for_images_sum_over_groups": round(work_wait_s, 4)},
                        "kernel_ms_per_step": {k: round(v[0] / Kh, 4) for k, v in sorted(h_timers.items())},
                        "host_stage_ms_per_group_step": {k: round(v / max(1, h_stage_n) * 1e3, 3) for k, v in h_stage_s.items() if v > 0},
                        "input": "pinned host memory, %d B per frame; ONE feeder thread sends the images of the next steps (a ring of three steps per group in HBM, "
                                 "at most two transfers queued on the device) on a copy stream of its own priority class; a group submits a step once its images "
                                 "have arrived (SDVL_BENCH_NO_INPUT_RING=1: uploaded inside the step)" % frame_bytes}
    # ---- third leg, SUSTAINED: S-A is 300 frames per sequence (main.cc:126-159 loops over the whole sequence).  The
    # first leg keeps every input frame resident (B x frames x 307 KB) and so measures a burst of a few dozen frames; here FRESH
    # trackers run whole sequences.  What stays resident is what the path itself retains: every keyframe keeps its HBM frame
    # (pyramid + corner list + descriptors + bins; the detection scratch left the frame in round 3) and its host-side features.
    # Input: a pool of D distinct sequences in pinned host memory (tracker i follows sequence i mod D; D x frames x 307 KB), fed
    # through the farm's input ring like the host-fed leg — every tracker's frames cross the link, nothing is shared on the device.
    sustained = None
    NF = args.sustained_frames
    lost_mix = None
    if NF > Wm + 2 or args.lost_mix_steps > 0:
        lib_hip = pkg.load_library()
        ctx.check(lib_hip.sdvl_device_free(ctx.h, C.c_void_p(buf)))
        farm.close()                        # the first legs' trackers, keyframes and rings go
        farm = None
        torch.cuda.synchronize()
    # ---- LOST-MIX leg: what a farm of real cameras sees — now and then a camera is covered or looks at a blank wall.  Until round 5 one
    # tracker with lost_frames >= 3 sent its whole batch of 256 to the host-driven path (14 k frame-steps/s); now its Relocalize
    # (sdvl.cc:73-89,205-238) runs inside the tabled step.  Input: D distinct sequences resident in HBM (tracker i follows i mod D) and ONE
    # featureless frame; a blinded tracker resumes, five steps later, with the frame it would have seen when it was blinded.
    if args.lost_mix_steps > 0:
        M, base_steps, blind_len, period, D = args.lost_mix_steps, 40, 5, 50, min(B, 32)
        NFp = 1 + Wm + base_steps + M
        farm = trk.TrackerFarm(local_rank, G, Bg, W_IMG, H_IMG, TUM_CAM, host_threads_per_group=threads)
        farm.set_fibers(fibers)
        if DISTORTION is not None:
            farm.set_distortion(DISTORTION)
        ctx3 = CtxView(pkg, farm.ctx_handle(0))
        pool = ctx3.malloc(D * NFp * frame_bytes)
        for k in range(NFp):
            views = [make_view(pkg, se3_exp(shard.sequence_twist(d) * k), SEQ_SEED(d), k) for d in range(D)]
            ctx3.render(views, pool + k * D * frame_bytes)
        grey_t = torch.full((frame_bytes,), 127, dtype=torch.uint8, device="cuda")   # (device memory for one frame; torch is the plumbing here)
        torch.cuda.synchronize()
        grey = grey_t.data_ptr()
        lp = np.zeros((NFp, B), np.uint64)
        first_mixed = 1 + Wm + base_steps
        n_blinded = 0
        for i in range(B):
            kk = 0
            for k in range(NFp):
                s_mix = k - first_mixed
                e = s_mix // period if s_mix >= 0 else -1
                blind = e >= 0 and (i % 20) == (10 * e) % 20 and (s_mix - e * period) < blind_len
                if blind:
                    lp[k, i] = grey
                    n_blinded += 1
                else:
                    lp[k, i] = pool + (kk * D + i % D) * frame_bytes
                    kk += 1
        farm.reserve(Bg * (NFp // 4 + 8))
        farm.run(lp[:1 + Wm], workers)                           # bootstrap keyframe + warm-up, untimed
        lbuf = farm.alloc_stats(max(base_steps, M))
        barrier()
        t0 = time.perf_counter()
        st_a = farm.run(lp[1 + Wm:first_mixed], workers, lbuf)
        barrier()
        el_a = time.perf_counter() - t0
        tracked_a = sum(int(st.quality != 2) for st in st_a[:base_steps * B])
        farm.stage_times(reset=True)
        barrier()
        t0 = time.perf_counter()
        st_m = farm.run(lp[first_mixed:], workers, lbuf)
        barrier()
        el_m = time.perf_counter() - t0
        mix = st_m[:M * B]
        tracked_m = sum(int(st.quality != 2) for st in mix)
        lm_stage_s, lm_stage_n = farm.stage_times()
        lost_mix = {"value": round(tracked_m / el_m, 2), "unit": "tracked frames/s", "steps": M, "ms_per_step": round(el_m / M * 1e3, 3),
                    "undisturbed_value": round(tracked_a / el_a, 2), "undisturbed_steps": base_steps,
                    "ratio_to_undisturbed": round((tracked_m / el_m) / max(1e-9, tracked_a / el_a), 4),
                    "frame_steps_per_s": round(M * B / el_m, 2),
                    "blinded_tracker_frames": n_blinded, "lost_frames": M * B - tracked_m,
                    "relocalized": sum(int(st.relocalized) for st in mix),
                    "frames_on_the_host_driven_path": sum(int(st.host_path != 0) for st in mix),
                    "relocalize_host_ms_per_group_step": round(lm_stage_s.get("relocalize", 0.0) / max(1, lm_stage_n) * 1e3, 3),
                    "schedule": "every %d steps 1 tracker in 20 sees %d featureless frames: TRACKING_BAD x3, two steps of Relocalize over its keyframes that fail, "
                                "one that succeeds" % (period, blind_len),
                    "input": "%d distinct sequences x %d frames resident in HBM (tracker i follows sequence i mod %d), one shared featureless frame" % (D, NFp, D)}
        ctx3.check(lib_hip.sdvl_device_free(ctx3.h, C.c_void_p(pool)))
        farm.close()
        farm = None
        del grey_t
        torch.cuda.synchronize()
    if NF > Wm + 2:
        import resource as _res
        free0, total0 = torch.cuda.mem_get_info()
        D = min(B, int(os.environ.get("SDVL_BENCH_SUSTAINED_DISTINCT", "32")))
        max_kf = args.sustained_max_keyframes
        kf_budget = NF // 4 + 8             # S-A turns about one frame in five into a keyframe
        if max_kf > 0:
            kf_budget = min(kf_budget, max_kf + 8)   # PlaneMap::LimitKeyframes hands the culled keyframes' HBM frames back to the pool
        need2 = G * Bg * kf_budget * footprint + 3 * B * frame_bytes + B * scratch_per_frame
        # host side: every keyframe keeps ~270 KB of Feature / Point objects (the reference's own representation)
        host_need = B * min(NF / 4.3, kf_budget) * 290e3 + D * NF * frame_bytes
        host_free = host_memory_available()
        if need2 > 0.9 * free0:
            sys.stderr.write("bench.py: sustained leg skipped: %d sequences x ~%d keyframes x %.2f MB = %.0f GB of HBM, %.0f GB free\n" %
                             (B, kf_budget, footprint / 1e6, need2 / 1e9, free0 / 1e9))
        elif host_free is not None and host_need > 0.85 * host_free:
            sys.stderr.write("bench.py: sustained leg skipped: %d sequences x %d frames need about %.0f GB of host memory (keyframe objects), %.0f GB available\n" %
                             (B, NF, host_need / 1e9, host_free / 1e9))
        else:
            if max_kf > 0:
                trk.configure(dict(trk.TUM_OVERRIDES, **dict(wl["over"], **{"SDVL.max_keyframes": max_kf})))
            farm = trk.TrackerFarm(local_rank, G, Bg, W_IMG, H_IMG, TUM_CAM, host_threads_per_group=threads)
            farm.set_fibers(fibers)
            if DISTORTION is not None:
                farm.set_distortion(DISTORTION)
            ctx2 = CtxView(pkg, farm.ctx_handle(0))
            spool = torch.empty(D * NF * frame_bytes, dtype=torch.uint8, pin_memory=True)
            tmp = ctx2.malloc(D * frame_bytes)
            for k in range(NF):
                views = [make_view(pkg, se3_exp(shard.sequence_twist(d) * k), SEQ_SEED(d), k) for d in range(D)]
                ctx2.render(views, tmp)
                ctx2.check(lib_hip.sdvl_device_download(ctx2.h, C.c_void_p(tmp), C.c_int64(D * frame_bytes), C.c_void_p(spool.data_ptr() + k * D * frame_bytes)))
            ctx2.check(lib_hip.sdvl_device_free(ctx2.h, C.c_void_p(tmp)))
            sptrs = (spool.data_ptr() + (np.arange(NF, dtype=np.uint64)[:, None] * D + (np.arange(B, dtype=np.uint64) % D)[None, :]) * frame_bytes).astype(np.uint64)
            farm.set_input_ring(True)
            farm.set_host_input(True)
            farm.reserve(Bg * kf_budget)
            farm.run(sptrs[:1 + Wm], workers)                      # bootstrap keyframe + warm-up, untimed
            Ks = NF - 1 - Wm
            half = Ks // 2                                         # memory is read at the middle of the timed region and at its end
            # the timed region in chunks of ~100 steps: frame-steps/s and the tracked share of each — a long sequence may lose its scene
            # (S-A's plane leaves the view after ~900 frames: from then on every tracker relocalises at every frame)
            n_chunks = max(2, min(16, Ks // 100))
            n_chunks += n_chunks % 2                               # (even: the middle of the region is a chunk boundary)
            bounds = [half * c // (n_chunks // 2) for c in range(n_chunks // 2)] + [half + (Ks - half) * c // (n_chunks // 2) for c in range(n_chunks // 2 + 1)]
            sbuf = farm.alloc_stats(max(bounds[c + 1] - bounds[c] for c in range(n_chunks)))

            def mem_now():
                torch.cuda.synchronize()
                fr, _ = torch.cuda.mem_get_info()
                rss = 0.0
                try:
                    for line in open("/proc/self/status"):
                        if line.startswith("VmRSS:"):
                            rss = int(line.split()[1]) / 1e6
                except (OSError, ValueError):
                    pass
                return round((total0 - fr) / 1e9, 1), round(rss, 1)
            elapsed_s, tracked_s, kf_s, phases, mem_mid = 0.0, 0, 0, [], None
            for c in range(n_chunks):
                a, b = bounds[c], bounds[c + 1]
                barrier()
                t0 = time.perf_counter()
                sst = farm.run(sptrs[1 + Wm + a:1 + Wm + b], workers, sbuf)
                barrier()
                el = time.perf_counter() - t0
                trk_c = sum(int(st.quality != 2) for st in sst[:(b - a) * B])
                kf_s += sum(int(st.keyframe) for st in sst[:(b - a) * B])
                hp_c = sum(int(st.host_path != 0) for st in sst[:(b - a) * B])
                elapsed_s += el
                tracked_s += trk_c
                phases.append({"first_frame": 1 + Wm + a, "steps": b - a, "tracked_fraction": round(trk_c / ((b - a) * B), 4),
                               "frame_steps_per_s": round((b - a) * B / el, 1), "frames_on_the_host_driven_path": hp_c})
                if b == half:
                    mem_mid = mem_now()
            mem_end = mem_now()
            sustained = {"value": round(tracked_s / elapsed_s, 2), "unit": "frames/s", "frames_per_sequence": NF, "timed_steps": Ks, "ms_per_step": round(elapsed_s / Ks * 1e3, 3),
                         "sequences_per_gpu": B, "tracked_fraction": round(tracked_s / (B * Ks), 5),
                         "max_keyframes": max_kf if max_kf > 0 else "the workload's (never reached)",
                         "map": "BOUNDED-MAP variant: SDVL.max_keyframes = %s (the reference's cfg files say 1000, config.cc:63 defaults to 100) and the plane-map stub deletes a culled "
                                "keyframe's points with it, which Map::LimitKeyframes / EmptyTrash do not; cpu_baseline runs the oracle at the cfg's 1000" % (max_kf if max_kf > 0 else "the workload's"),
                         "keyframes_made_per_sequence": round(1 + kf_s / B * (NF - 1) / Ks, 1), "hbm_bytes_per_keyframe": int(footprint), "corner_capacity": corner_cap,
                         "hbm_used_gb_at_frame_%d" % (1 + Wm + half): mem_mid[0], "hbm_used_gb_at_end": mem_end[0], "hbm_total_gb": round(total0 / 1e9, 1),
                         "host_rss_gb_at_frame_%d" % (1 + Wm + half): mem_mid[1], "host_rss_gb_at_end": mem_end[1],
                         "host_max_rss_gb": round(_res.getrusage(_res.RUSAGE_SELF).ru_maxrss / 1e6, 1),
                         "pcie_h2d_gb_per_s": round(B * Ks * frame_bytes / elapsed_s / 1e9, 2),
                         "phases": phases,
                         "lost_phase_frame_steps_per_s": (lambda lp: round(sum(p_["steps"] for p_ in lp) * B / sum(p_["steps"] * B / p_["frame_steps_per_s"] for p_ in lp), 1) if lp else None)(
                             [p_ for p_ in phases if p_["tracked_fraction"] < 0.02]),
                         "lost_phase_note": "chunks of the timed region in which under 2 % of the frames were tracked: every tracker runs Relocalize (sdvl.cc:205-238) "
                                            "over all its keyframes at every frame, inside the tabled step",
                         "input": "host-fed through the input ring from %d distinct sequences x %d frames in pinned host memory (tracker i follows sequence i mod %d)" % (D, NF, D)}
            farm.set_host_input(False)
            del spool
    if cpu_sample is not None:
        fps1, n_tracked1, secs1 = cpu_one
        n_thr = max(1, min(ncpu, 16))
        # the headline CPU figure: the oracle built HERE with the reference's own flags (CMakeLists.txt:20: -O3 -march=native), all
        # usable cores; beside it the same with the checker's flags (-march=x86-64-v3 -ffp-contract=off: the build every parity test
        # compares with).  Neither calls OpenCV's SIMD pyrDown / FAST as the reference would: "kind" stays "port".
        try:
            fps, n_tracked, secs = cpu_baseline(cpu_sample, args.mapper, n_thr, ref_flags=True)
            fps1r = cpu_baseline(cpu_sample, args.mapper, 1, ref_flags=True)[0] if n_thr > 1 else fps
            flags = "-O3 -march=native -msse3 (the reference's CMakeLists.txt:20), built on this host"
            fps_chk = (cpu_all if cpu_all else (cpu_baseline(cpu_sample, args.mapper, n_thr) if n_thr > 1 else cpu_one))[0]
        except Exception as e:   # no compiler on the box: the checker's build is all there is
            sys.stderr.write("bench.py: timing build of the oracle failed (%s); cpu_baseline uses the checker's build\n" % e)
            fps, n_tracked, secs = cpu_all if cpu_all else (cpu_baseline(cpu_sample, args.mapper, n_thr) if n_thr > 1 else cpu_one)
            fps1r, fps_chk, flags = fps1, fps, "-O3 -march=x86-64-v3 -ffp-contract=off (the checker's build)"
        cpu = {"value": round(fps, 2), "unit": "tracked frames/s", "cores": n_thr, "kind": "port", "flags": flags,
               "sample": "sequence 0 of the same workload, %d tracked frames per tracker: %d independent trackers on %d host threads after the GPU "
                         "run (%.1f s; %d usable CPUs); one tracker on one core before it (%.1f s)" % (n_tracked1, n_thr, n_thr, secs, ncpu, secs1),
               "one_core": round(fps1r, 2),
               "checker_build": {"flags": "-O3 -march=x86-64-v3 -ffp-contract=off", "value": round(fps_chk, 2), "one_core": round(fps1, 2)}}

    if latency and cpu:
        for b in ("b1", "b1_lookahead", "b16", "b16_batched"):
            if b + "_frames_per_s" in latency:
                latency[b + "_vs_cpu_one_core"] = round(latency[b + "_frames_per_s"] / cpu["one_core"], 2)
        latency["target"] = "north_star: >= 30 x the CPU reference path for one camera = %.0f frames/s here" % (30 * cpu["one_core"])
    if rank == 0:
        frames_rank = B * K
        # the dominant kernel = the one with the most dispatch time in the timed region, nothing else (round 2 broke near-ties by
        # byte count, which named the kernel with the larger fraction: VERDICT r02)
        # per-kernel dispatch time per step: from the timed region when everything was timed there, otherwise from the warm-up steps
        all_ms_per_step = ({k: v[0] / K for k, v in timers.items()} if (time_all or not warm_timers)
                           else {k: v[0] / Wt for k, v in warm_timers.items()})
        dom = None
        if timers:
            dn = dom_name if (dom_name in timers and not time_all) else max(timers.items(), key=lambda kv: kv[1][0])[0]
            dom = (dn, timers[dn])
        roofline = None
        valu = pmc_valu(all_ms_per_step, B)
        if dom:
            name, (ms, launches) = dom
            avg_s = ms / max(1, launches) * 1e-3
            # measured per-frame averages (corners, features, requests, GN evaluations, LK iterations, FAST keypoints) feed the
            # §8(d) formula
            per_frame = algorithmic_bytes_per_frame(name, n_c / frames_rank, n_f / frames_rank, n_s / frames_rank,
                                                    n_ia / frames_rank, n_lk / max(1, n_s), n_m / frames_rank, n_kp_measured)
            launches_per_step = launches / K          # all groups together
            # descriptors of whole frames, Shi-Tomasi scores and the filter gather run on the frames that become keyframes only
            frames_per_step = n_kf / K if name in ("orb_describe", "shi_tomasi", "filter_gather", "filter_select", "filter_describe") else B
            frames_per_launch = frames_per_step / launches_per_step
            roofline = {"bound": "hbm", "kernel": name, "selection_rule": "argmax of kernel_ms_per_step (dispatch time, HIP events on the kernel's own stream)",
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_us": round(avg_s * 1e6, 2), "frames_per_launch": round(frames_per_launch, 1),
                        "achieved": None, "frac": None, "traffic": None, "traffic_source": None, "algorithmic_bytes_per_launch": None}
            # the counters need rocprofv3 passes of their own (tools/profile_bench.sh): the line quotes the committed ones when they were taken
            # on this workload and texture, and says whether the kernels have changed since (traffic_stale)
            stale, stale_detail = profile_staleness()
            same_input = True     # _pmc_rows only finds passes of this workload and texture (profiles/rNN/<tag>/)
            if per_frame is not None:
                bytes_per_launch = per_frame * frames_per_launch
                achieved = bytes_per_launch / avg_s / 1e9
                traffic, traffic_src = pmc_traffic_bytes(name, frames_per_launch) if same_input else (None, None)
                roofline.update({"achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                                 "traffic_stale": stale, "traffic_stale_detail": stale_detail,
                                 "algorithmic_bytes_per_launch": int(bytes_per_launch),
                                 "algorithmic_bytes_note": "SURVEY §8(d) per-frame bytes of this kernel x frames per launch; search_points adds the 31x31 ORB window of "
                                                           "every compared corner (961 B x matches): descriptors are computed inside the search" if name == "search_points" else
                                                           "SURVEY §8(d) per-frame bytes of this kernel x frames per launch (FAST keypoints per frame measured: %d)" % n_kp_measured})
            # the roof that governs: none of these kernels streams, they issue tens to hundreds of VALU instructions per byte.  Wave-level
            # VALU instructions per launch (SQ_INSTS_VALU of the committed SQ pass, scaled to this run's frames per launch) over the
            # launch's duration, against 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave instruction
            if valu and same_input:
                vk = valu["kernels"].get(name)
                roofline["valu"] = {
                    "peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS, "source": valu["source"] + " (SQ_INSTS_VALU, own --pmc pass)",
                    "peak_note": "nominal: 256 CUs x 4 SIMDs x 2.4 GHz / 2; streams of independent instructions reach (tools/valu_peak_probe.cpp)",
                    "measured_issue_rates": measured_issue_rates(),
                    "kernel_insts_per_launch": int(vk["insts_per_dispatch"] * frames_per_launch / valu["frames_per_dispatch"]) if vk else None,
                    "kernel_frac": round(vk["insts_per_dispatch"] * frames_per_launch / valu["frames_per_dispatch"] / avg_s / VALU_PEAK_WAVE_INSTS, 4) if vk else None,
                    "path_insts_per_frame": int(valu["path_insts_per_frame"]),
                    "path_frac": round(valu["path_insts_per_frame"] * (tracked_all / elapsed_max / world) / VALU_PEAK_WAVE_INSTS, 4),
                    "insts_per_frame": {k: int(v["insts_per_frame"]) for k, v in sorted(valu["kernels"].items(), key=lambda kv: -kv[1]["insts_per_frame"])[:8]}}
            # Beside the contract's kernel (most dispatch TIME in the timed region) the line names the kernel with the most vector
            # instructions per frame (committed SQ pass) with the same figures, its launch duration from the warm-up steps in which every
            # dispatch carried events: the two differ when a short kernel launched several times per step (pyr_down) collects more queueing
            # time than the kernel that does the most work.
            if valu and valu["kernels"] and warm_timers:
                hv = max(valu["kernels"].items(), key=lambda kv: kv[1]["insts_per_frame"])[0]
                if hv in warm_timers and warm_timers[hv][1] > 0:
                    hv_avg_s = warm_timers[hv][0] / warm_timers[hv][1] * 1e-3
                    hv_fpl = B / (warm_timers[hv][1] / max(1, Wt))
                    hv_pf = algorithmic_bytes_per_frame(hv, n_c / frames_rank, n_f / frames_rank, n_s / frames_rank, n_ia / frames_rank, n_lk / max(1, n_s),
                                                        n_m / frames_rank, n_kp_measured)
                    if hv_pf:
                        hv_traffic, _ = pmc_traffic_bytes(hv, hv_fpl)
                        roofline["heaviest_by_instructions"] = {
                            "kernel": hv, "insts_per_frame": int(valu["kernels"][hv]["insts_per_frame"]),
                            "share_of_path_insts": round(valu["kernels"][hv]["insts_per_frame"] / valu["path_insts_per_frame"], 3),
                            "avg_launch_us": round(hv_avg_s * 1e6, 2), "frames_per_launch": round(hv_fpl, 1),
                            "algorithmic_bytes_per_launch": int(hv_pf * hv_fpl), "achieved": round(hv_pf * hv_fpl / hv_avg_s / 1e9, 2),
                            "frac": round(hv_pf * hv_fpl / hv_avg_s / 1e9 / HBM_PEAK_GBS, 6), "traffic": hv_traffic,
                            "valu_frac": round(valu["kernels"][hv]["insts_per_dispatch"] * hv_fpl / valu["frames_per_dispatch"] / hv_avg_s / VALU_PEAK_WAVE_INSTS, 4),
                            "timed": "the last %d warm-up step(s), every dispatch timed" % Wt}
            # the whole path against HBM: all kernels' algorithmic bytes per tracked frame x frames/s
            path_bytes = 0.0
            for k in all_ms_per_step:
                pf = algorithmic_bytes_per_frame(k, n_c / frames_rank, n_f / frames_rank, n_s / frames_rank, n_ia / frames_rank, n_lk / max(1, n_s),
                                                 n_m / frames_rank, n_kp_measured)
                if pf is None:
                    continue
                share = (n_kf / frames_rank) if k in ("orb_describe", "shi_tomasi", "filter_gather", "filter_select", "filter_describe") else 1.0
                path_bytes += pf * share      # (pyr_down's figure already covers its four launches)
            roofline["path_hbm"] = {"algorithmic_bytes_per_frame": int(path_bytes), "achieved": round(path_bytes * (tracked_all / elapsed_max / world) / 1e9, 1),
                                    "frac": round(path_bytes * (tracked_all / elapsed_max / world) / 1e9 / HBM_PEAK_GBS, 5)}
        if roofline is not None and host_fed:
            # the link is the roof of the drop-in figure: SDVL::HandleFrame takes a HOST image (sdvl.cc:55-59); 307,200 B per frame over
            # PCIe Gen5 x16.  peak = what ONE lone transfer stream of large pieces gets on these boxes (profiles/r03/pcie_probe.txt: 57 GB/s;
            # the link's nominal 63 GB/s beside it)
            per_gpu = host_fed["pcie_h2d_gb_per_s"] / world
            roofline["link"] = {"bound": "pcie_h2d", "achieved": round(per_gpu, 2), "peak": PCIE_LONE_STREAM_GBS, "unit": "GB/s per GPU",
                                "frac": round(per_gpu / PCIE_LONE_STREAM_GBS, 4), "nominal_peak": 63.0, "per_gpu_min_max": host_fed["pcie_h2d_gb_per_s_per_gpu"],
                                "frames_per_s_at_peak_per_gpu": int(PCIE_LONE_STREAM_GBS * 1e9 / frame_bytes),
                                "note": "value_host_fed / n_gpus x %d B per frame; the resident `value` is %.2f x what this link can feed: single-GPU kernel work "
                                        "moves the resident figure only" % (frame_bytes, (tracked_all / elapsed_max) / max(1.0, host_fed["value"]))}
        value = tracked_all / elapsed_max
        out = {
            "metric": "tracked frames/sec (%dx%d, 5-lvl pyr, %s)" % (W_IMG, H_IMG, FEATS_LABEL), "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(elapsed_max / K * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64", "data": "synthetic",
            "value_host_fed": host_fed["value"] if host_fed else None, "host_fed": host_fed,
            "value_sustained": sustained["value"] if sustained else None, "sustained": sustained,
            "value_lost_mix": lost_mix["value"] if lost_mix else None, "lost_mix": lost_mix,
            "value_hw_queues_16": hw_queues["value"] if hw_queues else None, "hw_queues": hw_queues,
            "config": {"workload": "%s: synthetic %s %dx%d mono, textured plane z=2m, %d independent sequences per GPU, "
                                   "one tracked frame per sequence per step%s" % (args.workload, {"S-A": "TUM fr1-like", "S-B": "EuRoC MH_01-like (config_euroc.cfg)", "S-C": "roofline case"}[args.workload], W_IMG, H_IMG, B, "; map = reference mapper run inside the step (sequential mode)" if args.mapper else ""),
                       "input": "hbm_resident (frames rendered into HBM before the timed region; the host-fed rate is value_host_fed)",
                       "distortion": ("%s: %s; frames rendered through the lens, Camera::UndistortImage inside every step (cached map, fused into the upload)" %
                                      (args.distortion, DISTORTION)) if DISTORTION is not None else "none (pinhole frames: config/*.cfg's d1 = 0 case, camera.cc:46)",
                       "texture": args.texture + (": piecewise-smooth shading + soft-edged shapes at three scales (csrc/sdvl_synth.h SDVL_TEXTURE_CAMERA)" if TEXTURE else
                                                  ": five octaves of value noise, a FAST corner on every second tested pixel (rounds 1-4)"),
                       "look_ahead": not os.environ.get("SDVL_NO_LOOKAHEAD"), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                       "look_ahead_note": "the resident leg names step k+1's images before step k (SDVLBatch::SetNextImages): their pyramids and FAST are queued "
                                          "behind step k's chain; a live SDVL::HandleFrame caller has no next frame - the latency block is measured without it",
                       "chunks": ("%d chunks, texture seeds %d..%d; sequence g follows chunk shard.chunk_for_sequence(g) - one chunk per rank with --gpus %d" %
                                  (wl["chunks"], wl["seed0"], wl["seed0"] + wl["chunks"] - 1, wl["chunks"])) if "chunks" in wl else None,
                       "sequences_per_gpu": B, "groups_per_gpu": G, "sequences_per_group": Bg, "host_threads_per_group": threads, "host_worker_threads": workers, "group_steps_per_worker": fibers, "numa_node": numa_node, "parallelism": "sequences sharded over %d GPU(s)" % world,
                       "cross_gpu_relocalisation": "declined: a tracker's keyframes live in the HBM of the GPU that tracked them and Relocalize "
                                                   "(sdvl.cc:205-238) is one launch over all of them there; the round-robin partition and the min-index "
                                                   "reduce of a split exist (shard.keyframe_positions_for_rank / first_success, gloo + RCCL tests) and are not used",
                       "features_per_frame": round(n_f / frames_rank, 1), "corners_per_frame": round(n_c / frames_rank, 1),
                       "search_requests_per_frame": round(n_s / frames_rank, 1), "gn_evaluations_per_frame": round(n_ia / frames_rank, 1),
                       "lk_iterations_per_request": round(n_lk / max(1, n_s), 2), "fast_keypoints_per_frame": n_kp_measured,
                       "matches_per_frame": round(n_m / frames_rank, 1), "keyframes_per_frame": round(n_kf / frames_rank, 3)},
            "roofline": roofline, "cpu_baseline": cpu, "latency": latency,
            "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(all_ms_per_step.items())},
            "kernel_timing": {"timed_region": "every dispatch" if time_all else ("none" if no_timing else "launches of %s only (the roofline's kernel)" % dom_name),
                              "kernel_ms_per_step_from": "the timed region" if (time_all or not warm_timers) else "the last %d warm-up step(s), every dispatch timed" % Wt,
                              "note": "dispatch events on every launch of 16 streams cost ~10 % of the throughput (DESIGN 7): rounds 1-3 timed everything inside the timed region"},
            "host_stage_ms_per_group_step": {k: round(v / max(1, stage_n) * 1e3, 3) for k, v in stage_s.items()},
            "host_cpu": {"cpus_busy": round(cpu_s / elapsed, 2), "usable": ncpu, "quota_throttled_ms": round((thr1[1] - thr0[1]) / 1e3, 1),
                         "minor_faults": flt1 - flt0},
            "speedup_vs_cpu_1core": round(value / cpu["one_core"], 2) if cpu else None,
            "speedup_vs_cpu_all_cores": round(value / cpu["value"], 2) if cpu else None,
        }
        print(json.dumps(out))
    if farm is not None:
        farm.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
