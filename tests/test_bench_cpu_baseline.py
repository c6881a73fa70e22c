"""bench.py's `cpu_baseline` leg on the CPU: the oracle tracker timed on one thread and on several (SURVEY §8d (ii):
N cores = N independent trackers).  The trackers must not share state: every thread reports the same tracked count."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as ol  # noqa: E402


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cpu_baseline_one_and_many_threads():
    bench = _bench_module()
    orc, syn = ol.Oracle(), ol.Synth()
    frames = [syn.render(ol.trajectory_pose(orc, k), ol.TUM_CAM, 640, 480, frame_id=k) for k in range(5)]
    fps1, tracked1, secs1 = bench.cpu_baseline(frames, False, 1)
    assert tracked1 == 4 and fps1 > 0 and secs1 > 0           # frame 0 is the bootstrap keyframe
    fps3, tracked3, secs3 = bench.cpu_baseline(frames, False, 3)
    assert tracked3 == 3 * tracked1                             # independent trackers: each tracks every frame
    assert np.isfinite(fps3) and fps3 > 0 and abs(fps3 - tracked3 / secs3) < 1e-6 * fps3
