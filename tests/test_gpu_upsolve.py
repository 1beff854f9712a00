"""rome_clique_upsolve (IIF upGibbsCliqueDensity, device-resident) against the oracle's restatement of the same loop, clique by
clique on the hexagon (BASELINE configs[0]; windows of test/testHexagonal2D_CliqByCliq.jl:37-79)."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

import rome_jl_amd as R   # noqa: E402
import oracle as ro       # noqa: E402
from solve_ref import upsolve_ref   # noqa: E402

CLIQUES = [["x0", "x1"], ["x2"], ["x3"], ["x4"], ["x5"], ["x6", "l1"]]


def _hex(N=100, seed=5):
    fg = R.generateGraph_Hexagonal(N=N)
    R.dead_reckon_init(fg, seed=seed)
    fg.initVariable("l1", np.array([[20.0], [0.0]]) + np.random.default_rng(1).standard_normal((2, N)))
    return fg


def _wrapdiff(a, b, dim):
    d = a - b
    if dim == 3:
        d[2] = np.arctan2(np.sin(d[2]), np.cos(d[2]))
    return d


@pytest.mark.parametrize("schedule", ["sequential", "jacobi"])
def test_hexagon_clique_by_clique_equals_the_oracle_loop(schedule):
    N = 100
    fg_d, fg_o = _hex(N), _hex(N)
    for p in range(2):   # two upward passes over the cliques
        for ci, fr in enumerate(CLIQUES):
            seed = 1000 + 17 * p + ci
            res = R.upGibbsCliqueDensity(fg_d, fr, gibbsIters=3, schedule=schedule, seed=seed)
            ref = upsolve_ref(R, fg_o, fr, N, seed=seed, gibbs_iters=3, schedule=schedule)
            for l in fr:
                pts, bw = res[l]
                dim = pts.shape[0]
                d = _wrapdiff(pts.copy(), ref[l], dim)
                assert np.mean(np.abs(d) < 1e-6) > 0.9, (p, fr, l, np.mean(np.abs(d) < 1e-6))
                md = np.array([np.mean(d[k]) for k in range(dim)])
                assert np.abs(md).max() < 1e-3, (p, fr, l, md)          # north_star tolerance on the belief means
                bo = ro.kde_bandwidths(ref[l][None], 0b100 if dim == 3 else 0)[0]
                assert np.allclose(bw, bo, rtol=2e-2), (l, bw, bo)      # the manikde! bandwidth setValKDE! stores
                fg_o.initVariable(l, ref[l])
    # statistical windows of the reference's clique-by-clique hexagon test (test/testHexagonal2D_CliqByCliq.jl:37-79)
    want = {"x0": (0, 0), "x1": (10, 0), "x2": (15, 8.66), "x3": (10, 17.32), "x4": (0, 17.32), "x5": (-5, 8.66), "x6": (0, 0), "l1": (20, 0)}
    for l, (x, y) in want.items():
        pts = fg_d.getVal(l)
        inside = np.mean((np.abs(pts[0] - x) < 3.0) & (np.abs(pts[1] - y) < 3.0))
        assert inside > 0.55, (l, inside)


def test_upsolve_messages_and_layouts_and_errors():
    N = 100
    fg = _hex(N)
    # an upward message on x1 (a tight density at a shifted position) pulls the product of x1 towards it
    msg = fg.getVal("x1").copy(); msg[0] += 1.0; msg[:2] = msg[:2].mean(1, keepdims=True) + 0.05 * (msg[:2] - msg[:2].mean(1, keepdims=True))
    a = R.upGibbsCliqueDensity(fg, ["x1"], gibbsIters=2, seed=9, setvals=False)["x1"][0]
    b = R.upGibbsCliqueDensity(fg, ["x1"], gibbsIters=2, seed=9, setvals=False, messages={"x1": [msg]})["x1"][0]
    ref = upsolve_ref(R, fg, ["x1"], N, seed=9, gibbs_iters=2, messages={"x1": [msg]})["x1"]
    assert np.mean(np.abs(_wrapdiff(b.copy(), ref, 3)) < 1e-6) > 0.9
    assert abs(b[0].mean() - msg[0].mean()) < abs(a[0].mean() - msg[0].mean())
    # rows not grouped by target in update order / a row targeting a variable that is not updated -> ROME_ERR_INVALID_ARG
    from rome_jl_amd.clique import CliqueBatch
    pairs = [(fl, d) for d in ("x1", "x2") for fl, labels, _ in fg.factors if d in labels]
    batch = CliqueBatch(fg, pairs)
    o = R.make_opts(N=N, seed=3)
    with pytest.raises(R.RomeError):
        batch.upsolve(o, ["x2", "x1"])        # rows are grouped x1-first
    with pytest.raises(R.RomeError):
        batch.upsolve(o, ["x1"])              # rows targeting x2 have no updated variable
    ok = batch.upsolve(o, ["x1", "x2"])
    assert set(ok) == {"x1", "x2"} and np.isfinite(ok["x1"][0]).all() and (ok["x1"][1] > 0).all()
    with pytest.raises(R.RomeError):
        batch.upsolve(R.make_opts(N=N, seed=3), ["x1", "x1"])


def test_upsolve_pcie_inclusive_time_per_clique():
    """ms per clique, host beliefs in -> new host beliefs out (PCIe and Python included); written to gpurun_out/ for profiles/."""
    N = 100
    fg = _hex(N)
    R.upGibbsCliqueDensity(fg, ["x0", "x1"], seed=1, setvals=False)
    ts = []
    for rep in range(20):
        t0 = time.perf_counter()
        for fr in CLIQUES:
            R.upGibbsCliqueDensity(fg, fr, gibbsIters=3, seed=rep, setvals=False)
        ts.append((time.perf_counter() - t0) / len(CLIQUES))
    ms = 1e3 * float(np.median(ts))
    tj = []
    for rep in range(20):
        t0 = time.perf_counter()
        for fr in CLIQUES:
            R.upGibbsCliqueDensity(fg, fr, gibbsIters=3, seed=rep, setvals=False, schedule="jacobi")
        tj.append((time.perf_counter() - t0) / len(CLIQUES))
    msj = 1e3 * float(np.median(tj))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r03_clique_upsolve.txt"), "w") as f:
        f.write("rome_clique_upsolve, hexagon cliques %s, N=100, gibbsIters=3, host beliefs in -> host beliefs out (PCIe + Python included):\n"
                "  sequential schedule %.3f ms per clique (median of 20 passes)\n  jacobi schedule     %.3f ms per clique\n" % (CLIQUES, ms, msj))
    assert ms < 50.0
