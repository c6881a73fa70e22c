import importlib, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oraclelib as ol
from oraclelib import TUM_CAM, XI, trajectory_pose
importlib.import_module("slam-sdvl_amd")
trk = importlib.import_module("slam-sdvl_amd.tracker")
orc, synth = ol.Oracle(), ol.Synth()

def run(name, blind, mode, B=16, n_frames=20):
    trk.configure()
    dev = trk.HostDevice(0)
    xis = [XI * (1.0 + 0.1 * i) * (1 if i % 2 == 0 else -1) for i in range(B)]
    seeds = [20260001 + i for i in range(B)]
    batch = trk.TrackerBatch(dev, B, 640, 480, TUM_CAM)
    oracles = [orc.tracker(640, 480, TUM_CAM) for _ in range(B)]
    shown = [0] * B
    bad = 0
    for k in range(n_frames):
        imgs = []
        for i in range(B):
            if i in blind and k in blind[i]:
                if mode == "grey":
                    imgs.append(np.full((480, 640), 127, np.uint8))
                else:
                    imgs.append(synth.render(trajectory_pose(orc, 40 + k, xis[i]), TUM_CAM, 640, 480, seed=seeds[i] + 5000, frame_id=1000 + k))
            else:
                imgs.append(synth.render(trajectory_pose(orc, shown[i], xis[i]), TUM_CAM, 640, 480, seed=seeds[i], frame_id=shown[i]))
                shown[i] += 1
        got = batch.step_host(imgs)
        for i in range(B):
            w, g = oracles[i].handle_frame(imgs[i]), got[i]
            a = (g.state, g.quality, g.matches, g.attempts, g.inliers, g.keyframe, g.relocalized, g.n_corners, g.host_path)
            b = (w.state, w.quality, w.matches, w.attempts, w.inliers, w.keyframe, w.relocalized, w.n_corners, 0)
            d = np.abs(np.array(g.pose[:]) - np.array(w.pose[:])).max()
            if i in blind:
                print(name, "k", k, "trk", i, "blind" if k in blind[i] else "", a, b, "pose d %.2e" % d, "DIFF" if a != b else "")
            if a != b or (d > 1e-4 and not (i in blind and k in blind[i])):
                bad += 1
    print(name, "mismatches", bad)
    batch.close()
    for o in oracles: o.close()
    dev.close()

run("A_only11_grey", {11: range(9, 14)}, "grey")
run("B_both_grey", {3: range(8, 13), 11: range(9, 14)}, "grey")
run("C_both_scene", {3: range(8, 13), 11: range(9, 14)}, "scene")
