#!/usr/bin/env python3
"""fast_cells_kernel on frames of different corner density: the synthetic plane of bench.py (a FAST corner on every second
pixel), white noise, and a smooth image with a few hundred small shapes (a few % corners, like a camera frame).
    python tools/fast_density_probe.py [frames per launch = 64]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

sdvl = importlib.import_module("slam-sdvl_amd")
import oraclelib as ol  # noqa: E402  (renderer only)
from test_gpu_parity import sparse_corner_image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = sdvl.Context(0)
ctx.timing_enable(True)
synth = ol.Synth()
orc_pose = np.array([1, 0, 0, 0, 0.05, 0.02, 0.0])
images = {
    "synthetic plane (bench.py)": synth.render(orc_pose, ol.TUM_CAM, 640, 480),
    "camera texture (bench.py --texture camera)": synth.render(orc_pose, ol.TUM_CAM, 640, 480, texture=1),
    "white noise": np.random.default_rng(1).integers(0, 256, (480, 640), dtype=np.uint8),
    "smooth + 500 small shapes": sparse_corner_image(3, 480, 640),
}
dp = sdvl.default_detect_params()
for name, img in images.items():
    frames = [ctx.frame(img) for _ in range(n)]
    ctx.timing_reset()
    for _ in range(5):
        got, _ = ctx.fast_cells(frames, dp, cap=60000)
    t = ctx.timing_get()["fast_cells"]
    kp = len(got[0][0])
    print("%-46s %7d keypoints after NMS per frame   fast_cells %7.1f us per %d frames" % (name, kp, t[0] / t[1] * 1e3, n))
    for f in frames:
        f.close()
