"""A/B of one sequence with the reference mapper: device-resident tables vs the host path, frame by frame (diagnostic)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as ol  # noqa: E402

trk = importlib.import_module("slam-sdvl_amd.tracker")
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
mapper = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
orc, syn = ol.Oracle(), ol.Synth()
trk.configure()
lib = trk.load_host_library()
dev = trk.HostDevice(0)
trk.set_mapper(mapper)
batches = []
for on in (1, 0):
    batches.append(trk.TrackerBatch(dev, 1, 640, 480, ol.TUM_CAM))
trk.set_mapper(False)
lib.sdvlh_batch_point_digest.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for k in range(n_frames):
    im = syn.render(ol.trajectory_pose(orc, k), ol.TUM_CAM, 640, 480, frame_id=k)
    rows = []
    for on, b in zip((1, 0), batches):
        lib.sdvlh_set_track_tables(on)
        g = b.step_host([im])[0]
        d = (C.c_double * 6)()
        lib.sdvlh_batch_point_digest(b.h, 0, d)
        ms = b.map_stats(0) if mapper else {}
        rows.append(((g.quality, g.keyframe, g.matches, g.attempts, g.inliers, g.outliers, g.align_meas, g.search_requests), list(d), ms, np.array(g.pose[:])))
    same = rows[0][0] == rows[1][0] and rows[0][1] == rows[1][1] and rows[0][2] == rows[1][2]
    print(k, "OK " if same else "DIFF", rows[0][0], ["%.6f" % x for x in rows[0][1]], float(np.abs(rows[0][3] - rows[1][3]).max()))
    if not same:
        print("   host path:", rows[1][0], ["%.6f" % x for x in rows[1][1]])
        print("   tables :", rows[0][2])
        print("   host   :", rows[1][2])
