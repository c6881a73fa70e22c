"""bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run), driven on CPU: two ranks over
gloo through the real sharding + shard.reduce_throughput path (SDVL_BENCH_DRY=1 replaces only the GPU work), and the
refusal to measure fewer GPUs than asked for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SDVL_BENCH_DRY")}
    env.update(kw)
    return env


def test_launcher_runs_two_ranks_over_gloo():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--seqs", "8"], env=_env(SDVL_BENCH_DRY="1"),
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["tracked"] == 2 * 8 * 3                         # SUM over ranks of (sequences per GPU x steps)
    assert d["sequences"] == [0, 7]                          # rank 0 owns the first 8 global sequences
    assert d["cpu_share"] >= 1
    # MAX over ranks: rank 1 sleeps 40 ms, rank 0 20 ms
    assert d["ms_per_step"] * 3 >= 39.0
    assert abs(d["value"] - d["tracked"] / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 0.02
    # round 4: the N > 1 line carries the DROP-IN figure too — the host-fed leg runs on every rank (8 steps each by default), its
    # tracked frames are summed, its time is the slowest rank's, and the ranks' own link rates come back as min / max
    hf = d["host_fed"]
    assert d["value_host_fed"] == hf["value"] and hf["steps"] == 8 and hf["tracked"] == 2 * 8 * 8
    lo, hi = hf["pcie_h2d_gb_per_s_per_gpu"]["min"], hf["pcie_h2d_gb_per_s_per_gpu"]["max"]
    assert 0 < lo < hi                                      # rank 1 sleeps twice as long as rank 0: its rate is the minimum
    assert hi / lo > 1.3
    link = d["roofline"]["link"]
    assert link["bound"] == "pcie_h2d" and link["peak"] == 57.0 and link["per_gpu_min_max"] == hf["pcie_h2d_gb_per_s_per_gpu"]
    assert abs(link["achieved"] - hf["pcie_h2d_gb_per_s"] / 2) < 1e-5 and abs(link["frac"] - link["achieved"] / 57.0) < 1e-6
    assert lo * 0.9 <= link["achieved"] <= hi * 1.1          # the job's per-GPU rate is set by the slowest rank's time


def test_host_fed_leg_can_be_switched_off_on_several_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--seqs", "8", "--host-steps", "0"],
                         env=_env(SDVL_BENCH_DRY="1"), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["value_host_fed"] is None and d["host_fed"] is None and d["roofline"]["link"] is None


def test_launcher_refuses_fewer_gpus_than_asked():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the refusal cannot be provoked")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode != 0
    assert "refusing" in out.stderr and not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


def test_world_size_must_match_gpus():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "2", "--warmup", "1"],
                         env=_env(SDVL_BENCH_DRY="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in out.stderr


def test_a_failing_rank_fails_the_launch():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "0", "--warmup", "0", "--seqs", "8"], env=_env(SDVL_BENCH_DRY="1"),
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0                               # steps = 0: every rank divides by zero -> the launcher reports it
    assert "exited with" in out.stderr


def _dry(*args):
    out = subprocess.run([sys.executable, BENCH, *args], env=_env(SDVL_BENCH_DRY="1"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])


def test_eight_ranks_s_b_every_rank_tracks_chunk_rank_mod_4():
    """the shape the driver's 8-GPU node runs, dry.  S-B (BASELINE config 4): four chunks, seeds 20260010..13; with more
    ranks than chunks rank r tracks chunk r mod 4.  Every rank plans with an eighth of the node's free host memory."""
    d = _dry("--gpus", "8", "--workload", "S-B", "--steps", "2", "--warmup", "1", "--seqs", "8", "--host-steps", "0")
    assert d["n_gpus"] == 8 and d["tracked"] == 8 * 8 * 2 and len(d["per_rank"]) == 8
    for r, pr in enumerate(d["per_rank"]):
        assert pr["rank"] == r and pr["first_sequence"] == 8 * r
        assert pr["seeds"] == [20260010 + r % 4], (r, pr["seeds"])
        assert pr["host_budget_bytes"] is None or pr["host_budget_bytes"] <= d["host_memory_available_bytes"] * 1.25 / 8   # (free memory moves a little between the ranks' readings)


def test_eight_ranks_s_c_seeds_follow_the_survey():
    """S-C (BASELINE config 5): 64 sequences per GPU x 8 GPUs, seeds 20260100 + rank*64 + i"""
    d = _dry("--gpus", "8", "--workload", "S-C", "--steps", "1", "--warmup", "1", "--seqs", "64", "--host-steps", "0")
    assert d["n_gpus"] == 8 and d["tracked"] == 8 * 64
    for r, pr in enumerate(d["per_rank"]):
        assert pr["seed_of_first"] == 20260100 + r * 64 and pr["seed_of_last"] == 20260100 + r * 64 + 63 and len(pr["seeds"]) == 64
