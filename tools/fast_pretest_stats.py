#!/usr/bin/env python3
"""How much of a frame could a cheap pre-test spare fast_cells?  (why the kernel stays at ~150 us per 256 frames.)
On frame 0 of the camera-like texture (or --plane), per detection level: the share of tested pixels that are FAST-9 corners, that pass the
compass pre-test (two ADJACENT compass pixels both brighter / darker than the centre by more than t — necessary for a 9-arc; what the
candidate-list path of fast_cells tests), that pass the stronger octant test (4 consecutive of the 8 even ring positions; also necessary)
and the conjunction with the same test on the odd positions — and the share of 4-pixel groups (the unit a lane of the dense path scores)
that hold at least one such pixel.  CPU only (numpy + the oracle's FAST for the corner counts).
    python tools/fast_pretest_stats.py [--plane]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oraclelib as ol  # noqa: E402

texture = 0 if "--plane" in sys.argv else 1
orc, syn = ol.Oracle(), ol.Synth()
img = syn.render(ol.trajectory_pose(orc, 0), ol.TUM_CAM, 640, 480, texture=texture)
pyr = orc.pyramid(img)
t = 10
OFFS16 = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def consecutive(flags, k):
    n, out = len(flags), np.zeros_like(flags[0])
    for s in range(n):
        a = flags[s]
        for j in range(1, k):
            a = a & flags[(s + j) % n]
        out |= a
    return out


def groups4(m):
    w4 = (m.shape[1] // 4) * 4
    return m[:, :w4].reshape(m.shape[0], -1, 4).any(2).mean()


print("texture: %s, threshold %d" % ("plane (rounds 1-4)" if texture == 0 else "camera-like (round 5)", t))
for l in range(3):
    p = pyr[l].astype(np.int32)
    H, W = p.shape
    v = p[3:-3, 3:-3]
    ring = [p[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] - v for dx, dy in OFFS16]
    corner = consecutive([r > t for r in ring], 9) | consecutive([r < -t for r in ring], 9)
    compass = consecutive([ring[i] > t for i in (0, 4, 8, 12)], 2) | consecutive([ring[i] < -t for i in (0, 4, 8, 12)], 2)
    octant = consecutive([ring[i] > t for i in range(0, 16, 2)], 4) | consecutive([ring[i] < -t for i in range(0, 16, 2)], 4)
    odd = consecutive([ring[i] > t for i in range(1, 16, 2)], 4) | consecutive([ring[i] < -t for i in range(1, 16, 2)], 4)
    assert corner.sum() == len(orc.fast(pyr[l], t, False)), "the numpy FAST-9 test disagrees with the oracle"
    print("level %d: corners %.3f | pass compass %.3f, octant %.3f, octant & odd %.3f | 4-pixel groups with a pixel that passes: compass %.3f, "
          "octant %.3f, octant & odd %.3f, with a corner %.3f" % (l, corner.mean(), compass.mean(), octant.mean(), (octant & odd).mean(), groups4(compass),
                                                                groups4(octant), groups4(octant & odd), groups4(corner)))
