"""`python bench.py --steps K --warmup W` (what the driver runs at N = 1) prints ONE JSON line with the contract's keys, BASELINE.json's
metric, a roofline object for the dominant kernel and a cpu_baseline object; the numbers in it are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_bench_line_follows_the_contract():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ROME_BENCH_SHARED_DEVICE", "ROME_BENCH_FORCE_EXCHANGE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-modes", "--cpu-seconds", "4"],
                       env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and p.stdout.rstrip().splitlines()[-1] == lines[0]     # ONE line, and it is the last thing on stdout
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "convolutions/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None      # BASELINE.md publishes no number
    assert d["dtype"] == "f64" and "synthetic" in d["data"]
    cfg = d["config"]
    assert "Manhattan-3500" in cfg["workload"] and cfg["particles"] == 100 and "model" not in cfg
    assert cfg["convolutions_per_step_per_gpu"] == 10907          # 2 x 5453 Pose2Pose2 directions + the PriorPose2 row
    # value = units / the timed region: whole-job convolutions over the median block
    assert abs(d["value"] - 10907 / (d["ms_per_step"] * 1e-3)) <= 1e-9 * d["value"]
    assert len(d["timed_blocks_ms_per_step"]) == 9 and sorted(d["timed_blocks_ms_per_step"])[4] == pytest.approx(d["ms_per_step"])
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.3 < r["frac"] < 1.0
    # algorithmic bytes: 48 B per relative particle (fixed 24 + proposal 24) + 24 B per prior particle, per launch
    assert r["algorithmic_bytes_per_launch"] == 10906 * 100 * 48 + 100 * 24
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms_per_launch"] * 1e-3) / 1e9)
    assert r["kernel_ms_per_launch"] <= d["ms_per_step"] * 1.05   # the kernel's period cannot exceed the step it is part of
    assert "traffic" in r                                          # counter-measured bytes of the stored PMC pass, or null
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "convolutions/s" and c["sample"]
    assert d["value"] > 20 * c["value"]                            # north_star's target: >= 20x the CPU reference path
