"""bench.py's one-line JSON contract, exercised on the GPU box with a tiny workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, env=None, want_stderr=False):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--seqs", "32", "--cpu-frames", "12", "--sustained-frames", "40", "--latency-frames", "12",
                          "--lost-mix-steps", "20", *extra], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    if want_stderr:
        lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
        return json.loads(lines[0]), out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def check(d, steps=4, warmup=1, mapper=False):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("tracked frames/sec") and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["input"].startswith("hbm_resident")
    # the host-fed leg (frames uploaded inside the step) rides along and is never the headline value
    assert d["value_host_fed"] == d["host_fed"]["value"] and d["value_host_fed"] > 0
    assert d["host_fed"]["pcie_h2d_gb_per_s"] > 0 and d["host_fed"]["steps"] >= 1
    assert d["value"] > 0 and abs(d["value"] - 32 * steps / (d["ms_per_step"] * steps / 1e3)) / d["value"] < 0.02   # every frame tracked
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["traffic"] is None or r["traffic"] > 0
    # the roofline names THE dominant kernel: the one with the most dispatch time in the timed region, whatever its byte count
    ms = d["kernel_ms_per_step"]
    assert r["kernel"] == max(ms, key=ms.get), (r["kernel"], ms)
    assert abs(r["avg_launch_us"] * 1e-6 * r["achieved"] * 1e9 - r["algorithmic_bytes_per_launch"]) / r["algorithmic_bytes_per_launch"] < 0.01
    assert r["path_hbm"]["frac"] > 0
    # round 4: every dispatch is timed during the warm-up steps; inside the timed region only the roofline's kernel carries events
    kt = d["kernel_timing"]
    if warmup >= 3:
        assert kt["timed_region"] == "launches of %s only (the roofline's kernel)" % r["kernel"] and "warm-up" in kt["kernel_ms_per_step_from"]
    else:   # too few warm-up steps to rank the kernels: every dispatch of the timed region carries events, as in rounds 1-3
        assert kt["timed_region"] == "every dispatch" and kt["kernel_ms_per_step_from"] == "the timed region"
    # the link is the roof of the drop-in (host-fed) figure
    ln = r["link"]
    assert ln["bound"] == "pcie_h2d" and ln["peak"] == 57.0 and abs(ln["frac"] - ln["achieved"] / 57.0) < 1e-3
    assert abs(ln["achieved"] - d["host_fed"]["pcie_h2d_gb_per_s"]) < 0.02 and ln["per_gpu_min_max"]["min"] > 0
    assert d["config"]["cross_gpu_relocalisation"].startswith("declined")
    if "valu" in r:   # present when an SQ pass is committed under profiles/
        assert r["valu"]["peak_wave_insts_per_s"] == 256 * 4 * 2.4e9 / 2 and 0 < r["valu"]["path_frac"] < 1
    if d["sustained"] is not None:   # whole sequences on fresh trackers, host-fed (skipped with the reference's mapper in the step)
        u = d["sustained"]
        assert d["value_sustained"] == u["value"] > 0 and u["frames_per_sequence"] == 40 and u["timed_steps"] == 40 - 1 - warmup
        assert u["tracked_fraction"] > 0.99 and 2 <= u["keyframes_made_per_sequence"] <= 20 and u["hbm_bytes_per_keyframe"] < 800000
        assert u["max_keyframes"] == 48 and u["hbm_used_gb_at_end"] > 0 and u["host_rss_gb_at_end"] > 0
    # round 5: the counters the line quotes say whether the kernels changed since they were taken (None: no stamp committed)
    assert "traffic_stale" in r and r["traffic_stale"] in (None, True, False)
    if r["traffic_stale"] is not None:
        assert "changed_files" in r["traffic_stale_detail"]
    assert d["config"]["texture"].split(":")[0] in ("plane", "camera") and d["config"]["look_ahead"] in (True, False)
    # round 5: the reference's own shape of use — one camera through SDVL::HandleFrame — and 16 cameras, threads and batched
    lat = d["latency"]
    if not mapper:
        # presence and positivity only: 12-frame runs on a loaded box order these rates any way they like
        assert lat is not None and lat["b1_tracked"] == 11 and lat["b1_frames_per_s"] > 0 and lat["b16_tracked"] == 16 * 11
        assert lat["b16_batched_frames_per_s"] > 0
        assert lat["b1_lookahead_frames_per_s"] > 0                              # the sequence-from-disk form (SDVL::SetNextImage), reported apart
        assert abs(lat["b1_vs_cpu_one_core"] - lat["b1_frames_per_s"] / d["cpu_baseline"]["one_core"]) < 0.02
    # round 6: the lost-mix leg — blinded trackers relocalise inside the tabled step, nobody takes the host-driven form
    lm = d["lost_mix"]
    if lm is not None:
        assert d["value_lost_mix"] == lm["value"] > 0 and lm["steps"] == 20 and lm["undisturbed_value"] > 0
        assert lm["frames_on_the_host_driven_path"] == 0 and lm["blinded_tracker_frames"] == 10 and lm["relocalized"] == 2, lm
    # round 6: the resident leg again on 16 hardware queues (a child process with GPU_MAX_HW_QUEUES=16), reported beside `value`, never as it
    hq = d["hw_queues"]
    if not mapper:
        assert hq is not None and d["value_hw_queues_16"] == hq["value"] > 0 and hq["gpu_max_hw_queues"] == 16 and hq["steps"] == steps
        assert d["config"]["gpu_max_hw_queues"].startswith("runtime default")   # `value` itself was measured on the runtime's default
    else:
        assert hq is None and d["value_hw_queues_16"] is None
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "tracked frames/s" and "sample" in c
    assert 0 < c["one_core"] <= c["value"] * 1.05   # the all-core figure is at least the one-core figure
    assert "march=native" in c["flags"] and c["checker_build"]["value"] > 0   # timed with the reference's own flags; the checker's build beside it
    assert d["value"] > 2 * c["one_core"]      # even this tiny batch (32 sequences, launch-latency bound) beats a CPU core several times over


def test_bench_line_contract():
    check(run_bench())


def test_bench_line_contract_with_the_drivers_warmup():
    """--warmup 5 as the driver runs it: every dispatch timed over the last three warm-up steps, the roofline's kernel alone after that"""
    d = run_bench("--warmup", "5")
    check(d, warmup=5)


def test_bench_line_contract_mapper_and_fibers():
    check(run_bench("--mapper", "--groups", "4", "--workers", "2", "--fibers", "2", "--latency-frames", "0"), mapper=True)


def test_bench_diagnostic_switches(tmp_path):
    """the farm's three diagnostic switches (README): SDVL_NO_LOOKAHEAD (the resident leg does not queue the next step's pyramids ahead),
    SDVL_FARM_TIMELINE (one line per group-step), SDVL_PROFILE (sampling profile of the worker threads on stderr)"""
    tl = tmp_path / "timeline.txt"
    d, err = run_bench("--sustained-frames", "0", "--latency-frames", "0", "--lost-mix-steps", "0",
                       env={"SDVL_NO_LOOKAHEAD": "1", "SDVL_FARM_TIMELINE": str(tl), "SDVL_PROFILE": "1"}, want_stderr=True)
    assert d["config"]["look_ahead"] is False and d["value"] > 0
    assert tl.exists() and tl.stat().st_size > 0
    assert "samples of 1 ms CPU" in err


def test_bench_line_contract_s_b_on_the_camera_texture():
    """BASELINE config 4's workload (S-B: 752x480, config_euroc.cfg) on the camera-like texture"""
    d = run_bench("--workload", "S-B", "--texture", "camera", "--sustained-frames", "0")
    check(d)
    assert d["metric"].startswith("tracked frames/sec (752x480") and d["config"]["workload"].startswith("S-B")
    assert d["config"]["texture"].startswith("camera") and "20260010" in d["config"]["chunks"]
    assert 2000 < d["config"]["fast_keypoints_per_frame"] < 8000


def test_bench_line_contract_through_a_lens():
    """--distortion tum_f1: frames rendered through config_tum_f1.cfg's lens, Camera::UndistortImage inside every step (cached map)"""
    d = run_bench("--distortion", "tum_f1", "--sustained-frames", "0")
    check(d)
    assert d["config"]["distortion"].startswith("tum_f1") and d["kernel_ms_per_step"].get("undistort", 0) > 0
    assert d["host_fed"]["kernel_ms_per_step"].get("undistort", 0) > 0


def test_bench_line_contract_s_c():
    """BASELINE config 5's workload (S-C: 1280x960, 4000 features, 1000 matches), 16 sequences"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "S-C", "--seqs", "16", "--steps", "4", "--warmup", "2", "--cpu-frames", "6",
                          "--host-steps", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert d["metric"].startswith("tracked frames/sec (1280x960") and d["config"]["workload"].startswith("S-C") and d["config"]["sequences_per_gpu"] == 16
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    assert d["config"]["features_per_frame"] > 400 and d["config"]["matches_per_frame"] > 400     # really the 1000-match configuration
