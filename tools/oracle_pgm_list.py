#!/usr/bin/env python3
"""The CPU oracle (oracle/, test infrastructure) on a list of binary PGM frames with one of the reference's configuration
files — the other half of the real-data recipe in INTEGRATION.md:

    slam-sdvl_amd/host/track_sequence --list frames.txt --config config_tum_f2.cfg      (MI355X front-end)
    python tools/oracle_pgm_list.py   --list frames.txt --config config_tum_f2.cfg      (CPU restatement)

Both print one line per frame: index state quality matches attempts inliers pose(7).  The camera block of the file gives the
intrinsics and the distortion (Camera::UndistortImage runs before HandleFrame, main.cc:133), its SDVL.* keys override the
defaults of config.cc:55-85.  The first frame is bootstrapped from the scene plane (--plane nx ny nz d, default z = 2)."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as ol  # noqa: E402

KEYS = {"SDVL.pyramid_levels": "pyramid_levels", "SDVL.cell_size": "cell_size", "SDVL.max_fast_levels": "max_fast_levels",
        "SDVL.fast_threshold": "fast_threshold", "SDVL.num_features": "num_features", "SDVL.use_orb": "use_orb", "SDVL.orb_size": "orb_size",
        "SDVL.patch_size": "patch_size", "SDVL.max_align_its": "max_align_its", "SDVL.search_size": "search_size",
        "SDVL.align_patch_size": "align_patch_size", "SDVL.max_alignLevel": "max_align_level", "SDVL.min_alignLevel": "min_align_level",
        "SDVL.max_img_align_its": "max_img_align_its", "SDVL.min_feature_score": "min_feature_score", "SDVL.max_matches": "max_matches",
        "SDVL.min_matches": "min_matches", "SDVL.max_failed": "max_failed", "SDVL.max_optim_pose_its": "max_optim_pose_its",
        "SDVL.max_ransac_points": "max_ransac_points", "SDVL.max_ransac_its": "max_ransac_its", "SDVL.min_keyframe_its": "min_keyframe_its",
        "SDVL.inlier_error_threshold": "inlier_error_threshold", "SDVL.lost_ratio": "lost_ratio"}


def read_cfg(path):
    out = {}
    for line in open(path):
        line = line.split("#")[0].strip()
        if not line or line.startswith("%") or ":" not in line:
            continue
        k, v = line.split(":", 1)
        v = v.strip()
        if v.startswith('"'):
            continue
        try:
            out[k.strip()] = float(v)
        except ValueError:
            pass
    return out


def read_pgm(path):
    data = open(path, "rb").read()
    assert data[:2] == b"P5", path
    tok, pos = [], 2
    while len(tok) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tok.append(int(data[pos:end]))
        pos = end
    w, h, mx = tok
    assert mx == 255, path
    return np.frombuffer(data, np.uint8, w * h, pos + 1).reshape(h, w)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", required=True)
    ap.add_argument("--config", required=True)
    ap.add_argument("--plane", type=float, nargs=4, default=[0, 0, 1, 2.0])
    ap.add_argument("--mapper", action="store_true")
    a = ap.parse_args()
    cfg = read_cfg(a.config)
    orc = ol.Oracle()
    for k, field in KEYS.items():
        if k in cfg:
            is_int = dict(ol.Params._fields_)[field] is ctypes.c_int
            setattr(orc.params, field, int(cfg[k]) if is_int else float(cfg[k]))
    w, h = int(cfg.get("Camera.width", 640)), int(cfg.get("Camera.height", 480))
    cam = np.array([cfg["Camera.fx"], cfg["Camera.fy"], cfg["Camera.u0"], cfg["Camera.v0"]])
    dist = np.array([cfg.get("Camera.d%d" % i, 0.0) for i in range(1, 6)])
    trk = orc.tracker(w, h, cam, plane=a.plane)
    if a.mapper:
        trk.use_mapper(True)
    files = [l.strip() for l in open(a.list) if l.strip() and not l.startswith("#")]
    for k, f in enumerate(files):
        img = read_pgm(f)
        assert img.shape == (h, w), (f, img.shape)
        if dist[0] != 0.0:                       # Camera::SetDistortions tests d1 only (camera.cc:46)
            img = orc.undistort(img, cam, dist)
        st = trk.handle_frame(img)
        print(k, st.state, st.quality, st.matches, st.attempts, st.inliers, " ".join("%.17g" % v for v in st.pose))
    trk.close()


if __name__ == "__main__":
    main()
