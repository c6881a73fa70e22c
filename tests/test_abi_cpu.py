"""CPU-side checks of the boundary: the C-ABI library loads without a GPU and exports every symbol that
include/sdvl_hip.h declares; compute entry points refuse to run (no CPU fallback)."""
import ctypes as C
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sdvl():
    return importlib.import_module("slam-sdvl_amd")


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sdvl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdvl_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(sdvl):
    lib = sdvl.load_library()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libsdvl_hip.so does not export %s" % s
    assert sorted(sdvl.ABI_SYMBOLS) == syms


def test_struct_layouts_match_header(sdvl):
    # sizes the C side static-asserts implicitly by use; a mismatch here would corrupt batched records
    assert C.sizeof(sdvl.Keypoint) == 8
    assert C.sizeof(sdvl.AlignFeature) == 56
    assert C.sizeof(sdvl.AlignJob) == 8 + 8 + 8 + 56
    assert C.sizeof(sdvl.SearchReq) == 8 + 8 + 56 + 56 + 16 + 24 + 16 + 16 + 8 + 32
    assert C.sizeof(sdvl.SearchRes) == 16 + 6 * 4
    assert C.sizeof(sdvl.AlignResult) == 56 + 16 + 4 + 32 + 4 + 8


def test_no_cpu_fallback_without_gpu(sdvl):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sdvl.SdvlError):
        sdvl.Context(0)


def test_fast_num_cells_matches_survey(sdvl):
    lib = sdvl.load_library()
    dp = sdvl.default_detect_params()
    cpl = (C.c_int * 4)()
    tot = C.c_int()
    for (w, h), want in {(640, 480): [300, 80, 20], (752, 480): [360, 96, 24], (1280, 960): [1200, 300, 80]}.items():
        assert lib.sdvl_fast_num_cells(w, h, C.byref(dp), cpl, C.byref(tot)) == 0
        assert [cpl[i] for i in range(3)] == want and tot.value == sum(want)      # SURVEY §8 header


def test_orb_sincos_rounds_like_libm():
    """The ORB kernels take cos/sin of the float orientation angle in double and round to float
    (extra/orb_detector.cc:361-362).  The device uses csrc/sdvl_math.h `sincos_2pi` (plain IEEE double arithmetic, the
    same on host and device); its host build must round to the same floats as libm does, on angles all over [0, 2 pi]
    and hard against the quadrant boundaries."""
    import math
    import numpy as np
    trk = importlib.import_module("slam-sdvl_amd.tracker")
    lib = trk.load_host_library()
    fn = lib.sdvlh_sincos_2pi
    fn.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    fn.restype = None
    rng = np.random.default_rng(20260400)
    deg = np.concatenate([rng.uniform(0.0, 360.0, 60000), np.arange(0, 361, dtype=np.float64)]).astype(np.float32)
    ang = (deg.astype(np.float64) * np.float32(math.pi / np.float32(180.0))).astype(np.float32)   # orb_detector.cc:360
    near = []
    for k in range(5):                     # floats either side of k * pi/2
        c = np.float32(k * math.pi / 2)
        x = c
        for _ in range(200):
            near.append(x)
            x = np.nextafter(x, np.float32(10.0))
        x = c
        for _ in range(200):
            near.append(x)
            x = np.nextafter(x, np.float32(-1.0))
    ang = np.concatenate([ang, np.array([a for a in near if a >= 0], np.float32)])
    s, c = C.c_double(), C.c_double()
    worst = 0.0
    for a in ang.tolist():
        fn(a, C.byref(s), C.byref(c))
        assert np.float32(s.value) == np.float32(math.sin(a)) and np.float32(c.value) == np.float32(math.cos(a)), a
        worst = max(worst, abs(s.value - math.sin(a)), abs(c.value - math.cos(a)))
    assert worst < 3e-16


def test_cpp_example_fails_loudly_without_a_gpu():
    """the C++ example of the drop-in (host/track_sequence) has no CPU path either"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(ROOT, "slam-sdvl_amd", "host", "track_sequence")
    assert os.path.exists(exe), "build() makes it (make -C slam-sdvl_amd/host)"
    r = subprocess.run([exe, "--synthetic", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
