"""bench.py's one-line JSON contract, exercised on the GPU box with a tiny workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--seqs", "32", "--cpu-frames", "12",
                          *extra], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def check(d, steps=4, warmup=1):
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
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "tracked frames/s" and "sample" in c
    assert 0 < c["one_core"] <= c["value"] * 1.05   # the all-core figure is at least the one-core figure
    assert d["value"] > 30 * c["value"] * 0.0 and d["value"] > c["value"]      # even this tiny batch beats one CPU core


def test_bench_line_contract():
    check(run_bench())


def test_bench_line_contract_mapper_and_fibers():
    check(run_bench("--mapper", "--groups", "4", "--workers", "2", "--fibers", "2"))
