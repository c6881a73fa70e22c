"""The constants the tests, the bench and the host layer carry are the reference's: every value in tracker.TUM_OVERRIDES and
oraclelib.*_CAM / *_DIST equals what host/config.cc parses out of the reference's own configuration files.  Runs only where
/root/reference exists (the build container); the GPU box has no reference tree and skips it."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import oraclelib as ol

REF = "/root/reference/config"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def snapshot(cfg):
    trk = importlib.import_module("slam-sdvl_amd.tracker")
    lib = trk.load_host_library()
    lib.sdvlh_config_reset()
    lib.sdvlh_config_read.argtypes = [C.c_char_p]
    assert lib.sdvlh_config_read(os.path.join(REF, cfg).encode()) == 0
    out = (C.c_double * 20)()
    lib.sdvlh_config_snapshot(out)
    lib.sdvlh_config_reset()
    v = list(out)
    cam = dict(width=v[0], height=v[1], cam=np.array(v[2:6]), dist=np.array(v[6:11]))
    keys = ["SDVL.cell_size", "SDVL.min_avg_shift", "SDVL.max_matches", "SDVL.max_keyframes", "SDVL.use_orb", "SDVL.fast_threshold",
            "SDVL.lost_ratio", "SDVL.num_features", "SDVL.min_matches"]
    return cam, dict(zip(keys, v[11:]))


@pytest.mark.parametrize("cfg,cam,dist,size", [("config_tum_f1.cfg", ol.TUM_CAM, ol.TUM_DIST, (640, 480)),
                                               ("config_tum_f2.cfg", ol.TUM2_CAM, ol.TUM2_DIST, (640, 480)),
                                               ("config_euroc.cfg", ol.EUROC_CAM, ol.EUROC_DIST, (752, 480))])
def test_camera_constants_equal_the_reference_files(cfg, cam, dist, size):
    got, _ = snapshot(cfg)
    assert (got["width"], got["height"]) == size
    assert np.array_equal(got["cam"], cam) and np.array_equal(got["dist"], dist)


def test_tum_overrides_equal_config_tum_f1():
    trk = importlib.import_module("slam-sdvl_amd.tracker")
    _, sdvl = snapshot("config_tum_f1.cfg")
    for k, v in trk.TUM_OVERRIDES.items():
        assert sdvl[k] == float(v), k
    # f2 carries the same SDVL block (only the camera differs); EuRoC lowers min_matches to 5 (SURVEY §8 header)
    _, sdvl2 = snapshot("config_tum_f2.cfg")
    assert {k: sdvl2[k] for k in trk.TUM_OVERRIDES} == {k: float(v) for k, v in trk.TUM_OVERRIDES.items()}
    _, sdvle = snapshot("config_euroc.cfg")
    assert sdvle["SDVL.min_matches"] == 5 and sdvl["SDVL.min_matches"] == 20


def test_bench_workloads_use_these_constants():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_consts", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert np.array_equal(np.array(m.WORKLOADS["S-A"]["cam"]), ol.TUM_CAM)
    assert np.allclose(np.array(m.WORKLOADS["S-C"]["cam"]), 2.0 * ol.TUM_CAM)      # S-C: the same camera at twice the resolution
