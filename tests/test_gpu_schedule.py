"""Ordered up-solve schedules (rome_jl_amd.schedule.OrderedSolve): the initAll!-style init pass and Gauss-Seidel sweeps, device-resident
(DeviceStore + one UpsolvePlan per independent group), against the oracle's restatement of the same schedule (the stand-ins of
tests/dist_standin.py drive tests/solve_ref.py::upsolve_ref with the same groups, `usable` sets and Philox stream ids)."""
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
from rome_jl_amd.clique import DeviceStore   # noqa: E402
from rome_jl_amd.schedule import OrderedSolve   # noqa: E402
from dist_standin import OracleStore, OraclePlan   # noqa: E402


def _wd(a, b):
    d = a - b
    if d.shape[0] == 3:
        d[2] = np.arctan2(np.sin(d[2]), np.cos(d[2]))
    return d


def _compare(fg, kind, n_sweeps, seed):
    N = fg.N
    dev = OrderedSolve(DeviceStore(fg, upload=False), kind=kind)
    orc = OrderedSolve(OracleStore(R, fg), kind=kind, plan_cls=OraclePlan)
    assert [tuple(g) for g in dev.init_groups] == [tuple(g) for g in orc.init_groups]
    o = R.make_opts(N=N, seed=seed)
    dev.init(o); orc.init(o)
    worst = []
    for stage in range(n_sweeps + 1):
        if stage:
            dev.sweep(R.make_opts(N=N, seed=seed + stage)); orc.sweep(R.make_opts(N=N, seed=seed + stage))
        fr, dm = [], []
        for l in fg.variables:
            d = _wd(dev.store.get(l), orc.store.get(l))
            fr.append(np.mean(np.abs(d) < 1e-6)); dm.append(np.abs(d.mean(axis=1)).max())
        worst.append((float(np.mean(fr)), float(np.max(dm))))
    return worst


@pytest.mark.parametrize("kind", ["colour", "levels"])
def test_hexagon_schedule_equals_the_oracle_schedule(kind):
    fg = R.generateGraph_Hexagonal(N=100)           # no beliefs at all: the init pass creates them
    worst = _compare(fg, kind, 2, 11)
    for frac, dmean in worst:
        assert frac > 0.9 and dmean < 1e-3, worst   # north_star tolerance on the belief means
    # and the windows of test/testHexagonal2D_CliqByCliq.jl:37-79 after init + 2 sweeps on the device
    dev = OrderedSolve(DeviceStore(fg, upload=False), kind=kind)
    dev.init(R.make_opts(N=100, seed=5)); dev.sweep(R.make_opts(N=100, seed=6), 2)
    want = {"x0": (0, 0), "x1": (10, 0), "x2": (15, 8.66), "x3": (10, 17.32), "x4": (0, 17.32), "x5": (-5, 8.66), "x6": (0, 0), "l1": (20, 0)}
    for l, (x, y) in want.items():
        p = dev.store.get(l)
        assert np.mean((np.abs(p[0] - x) < 3.0) & (np.abs(p[1] - y) < 3.0)) > 0.55, (l, p[:2].mean(1))


def test_manhattan_prefix_schedule_equals_the_oracle_schedule():
    """first 120 edges of manhattan.g2o (loop closures included): init pass + one coloured sweep, device == oracle"""
    fg = R.loadG2o(os.path.join(ROOT, "tests", "golden", "manhattan.g2o"), N=64, max_edges=120)
    worst = _compare(fg, "colour", 1, 21)
    for frac, dmean in worst:
        assert frac > 0.9 and dmean < 1e-3, worst


def test_beehive_multihypo_schedule_runs_from_nothing():
    """BASELINE configs[3]: the beehive with ambiguous re-sightings through the ordered schedule (multihypo rows inside the plans)"""
    fg = R.synth_beehive_mh(20, N=100)
    dev = OrderedSolve(DeviceStore(fg, upload=False), kind="colour")
    dev.init(R.make_opts(N=100, seed=2)); dev.sweep(R.make_opts(N=100, seed=3), 3)
    sim = fg._sim
    err = [np.hypot(*(dev.store.get(l)[:2].mean(1) - np.asarray(sim[l])[:2])) for l in fg.variables if fg.variables[l] is R.Pose2]
    assert np.isfinite(err).all() and np.median(err) < 3.0, np.median(err)


def test_manhattan3500_from_nothing_sweeps_stall_clique_tree_stays_elimination_solves():
    """Manhattan-3500 from NOTHING (no dead reckoning, no parametric start), against the MAP (solveGraphParametric incl. its polish):
      * the init pass leaves metres and coloured Gauss-Seidel sweeps stay there (profiles/r04_ordered_solve.txt);
      * the clique-form tree solve (rome_jl_amd.tree, relative messages) does not improve on its init pass after rigid alignment -- one-shot
        outward clique solves, belief-weighted down pass (profiles/r06_tree_forms.txt);
      * variable elimination in relative-factor algebra (rome_jl_amd.elimination) solves it from the factors alone: every single pass
        <= 2.3 m raw (the reference's own solveTree! sits 2.26 m from the MAP on its Manhattan-500 graph), <= 1.0 m after alignment."""
    from rome_jl_amd.tree import TreeSolver
    from rome_jl_amd.elimination import RelativeEliminationSolver
    N = 100
    g2o = os.path.join(ROOT, "tests", "golden", "manhattan.g2o")
    fg = R.loadG2o(g2o, N=N)
    xp = R.solveGraphParametric(R.dead_reckon_init(R.loadG2o(g2o, N=N), seed=1))
    labels = list(fg.variables)
    mp = np.array([xp[l] for l in labels])

    def rms(aligned=False):
        m, _ = R.belief_stats(np.stack([fg.getVal(l) for l in labels]))
        A, B = m[:, :2], mp[:, :2]
        if aligned:
            A, B = A - A.mean(0), B - B.mean(0)
            U, _, Vt = np.linalg.svd(A.T @ B); Rr = (U @ Vt).T
            if np.linalg.det(Rr) < 0:
                Rr = (U @ np.diag([1.0, -1.0]) @ Vt).T
            A = A @ Rr.T
        return float(np.sqrt(np.mean(np.sum((A - B) ** 2, axis=1))))
    osv = R.initAllOrdered(fg, seed=1)
    r_init, a_init = rms(), rms(True)
    osv.sweep(R.make_opts(N=N, seed=100), 5); osv.store.download(fg)
    r_sweeps = rms()
    assert 3.0 < r_init < 8.0 and r_sweeps > 3.0              # the stall
    ts = TreeSolver(fg, messages="relative")
    st = ts.stats()
    assert st["levels"] < 80 and st["width_max"] > 500 and st["unreached"] == 0, st
    ts.upload()
    ali, secs = [], []
    for ps in range(3):
        store_ctx = ts.store.ctx
        store_ctx.synchronize(); t0 = time.perf_counter()
        ts.solve(R.make_opts(N=N, seed=100 + ps)); store_ctx.synchronize()
        secs.append(time.perf_counter() - t0)
        ts.download()
        ali.append(rms(True))
    assert 0.4 < np.median(ali) < 3.0 and max(secs) < 1.5, (a_init, ali, secs)
    es = RelativeEliminationSolver(fg)
    raw, ali2, secs2 = [], [], []
    for ps in range(6):
        es.reset()
        es.store.ctx.synchronize(); t0 = time.perf_counter()
        es.solve(R.make_opts(N=N, seed=100 + ps)); es.store.ctx.synchronize()
        secs2.append(time.perf_counter() - t0)
        es.download(fg)
        raw.append(rms()); ali2.append(rms(True))
    assert np.median(raw) <= 2.3 and max(raw) <= 3.0 and max(ali2) <= 1.0 and np.median(ali2) < min(a_init, np.median(ali)), (r_init, a_init, raw, ali2, ali)
    assert np.median(secs2) < 0.2, secs2


def test_init_all_ordered_keeps_existing_beliefs_and_solve_graph_takes_it():
    """R.initAllOrdered = IIF initAll! semantics on the device: variables that have a belief are left alone, the others get the product
    of ALL usable factors; R.solveGraph(init="ordered") strings it in front of the sweeps (hexagon windows of
    test/testHexagonal2D_CliqByCliq.jl:37-79 after the solve)."""
    N = 100
    fg = R.generateGraph_Hexagonal(N=N)
    rng = np.random.default_rng(0)
    x0 = np.array([[0.0], [0.0], [0.0]]) + 0.05 * rng.standard_normal((3, N))
    fg.initVariable("x0", x0)
    R.initAllOrdered(fg, seed=4)
    assert np.array_equal(fg.getVal("x0"), x0)                      # kept
    assert all(fg.isInitialized(l) for l in fg.variables)
    assert np.abs(fg.getVal("x3")[:2].mean(1) - [10.0, 17.32]).max() < 4.0 and np.abs(fg.getVal("l1")[:2].mean(1) - [20.0, 0.0]).max() < 3.0
    fg2 = R.generateGraph_Hexagonal(N=N)
    R.solveGraph(fg2, n_sweeps=4, init="ordered", bandwidth="lcv", product="gibbs", seed=77)
    want = {"x0": (0, 0), "x1": (10, 0), "x2": (15, 8.66), "x3": (10, 17.32), "x4": (0, 17.32), "x5": (-5, 8.66), "x6": (0, 0), "l1": (20, 0)}
    for l, (x, y) in want.items():
        p = fg2.getVal(l)
        assert np.mean((np.abs(p[0] - x) < 3.0) & (np.abs(p[1] - y) < 3.0)) > 0.55, (l, p[:2].mean(1))
