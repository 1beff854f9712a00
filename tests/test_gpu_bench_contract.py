"""`python bench.py --steps K --warmup W` (what the driver runs at N = 1) prints ONE JSON line with the contract's keys, BASELINE.json's
metric, a roofline object for the dominant kernel and a cpu_baseline object; the numbers in it are consistent with each other."""
import json
import os
import subprocess
import sys

import numpy as np
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
    # ONE period behind `value` and `frac` (the line quoted 0.838 from a separate event-bracketed block while value came from the median block)
    assert abs(r["frac"] - r["algorithmic_bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e9 / r["peak"]) < 0.03 * r["frac"]
    assert "analytic root" in cfg["solver"] and "inflate_cycles inert" in cfg["solver"]
    assert "traffic" in r                                          # counter-measured bytes of the stored PMC pass, or null
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "convolutions/s" and c["sample"]
    assert d["value"] > 20 * c["value"]                            # north_star's target: >= 20x the CPU reference path


@pytest.mark.timeout(1500)
def test_bench_line_carries_the_functor_iterating_solvers_and_the_tree_solve():
    """the full default line (what the driver records): roofline_by_solver for gauss_newton / nelder_mead (north_star: "residual + numerical
    root-find") and solve.from_tree (the solveTree!-shaped wall-clock from NO parametric start)"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ROME_BENCH_SHARED_DEVICE", "ROME_BENCH_FORCE_EXCHANGE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-seconds", "4"],
                       env=env, capture_output=True, text=True, timeout=1400, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    rb = d["roofline_by_solver"]
    for name, bpp in (("closed_form", 48), ("newton", 48), ("gauss_newton", 72), ("nelder_mead", 72)):
        r = rb[name]
        assert r["bytes_per_particle"] == bpp and r["frac"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms_per_launch"] * 1e-3) / 1e9 / 8000.0)
    assert rb["gauss_newton"]["frac"] < rb["newton"]["frac"] and rb["nelder_mead"]["frac"] < rb["gauss_newton"]["frac"]
    ft = d["solve"]["from_tree"]
    assert "error" not in ft, ft
    el = ft["elimination"]
    assert el["rounds"] < 40 and el["compositions"] > 5000 and el["factor_edges"] == 5453 and el["launch_steps"] < 120, el
    rr = [p_["rms_to_parametric_m"] for p_ in ft["passes"]]
    # median raw RMS <= 2.3 m (the reference's own level on its Manhattan-500 solve), stable over the passes
    assert len(rr) == 8 and np.median(rr) <= 2.3 and max(rr[2:]) / min(rr[2:]) <= 2.0, rr
    assert np.median(ft["independent_single_passes_rms_to_parametric_m"]) <= 2.3
    assert max(p_["rms_after_rigid_alignment_m"] for p_ in ft["passes"]) < 1.5
    assert ft["seconds_per_pass"] < 0.2 and ft["wall_clock_s_first_pass"] < 2.5
    cf = d["solve"]["from_tree_clique_forms"]
    assert "error" not in cf, cf
    assert cf["tree"]["levels"] > 20 and cf["tree"]["cliques"] > 2500 and sum(cf["frontier_width_by_level"]) == cf["tree"]["cliques"]
    assert np.median(rr) < cf["rms_to_parametric_m_after_init"]       # the elimination solve improves on what initAll! leaves
