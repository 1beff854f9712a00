"""rome_jl_amd.elimination on the CPU: the host structure (rounds, merges, compositions, back substitution order) executed through the
oracle-backed stand-ins of tests/dist_standin.py -- small graphs whose answers are known."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import rome_jl_amd as R   # noqa: E402
from rome_jl_amd.elimination import RelativeEliminationSolver   # noqa: E402
from dist_standin import OracleTreeBackend   # noqa: E402


def _mean(es, l):
    b = es.store.get(l)
    return np.array([b[0].mean(), b[1].mean(), np.arctan2(np.sin(b[2]).sum(), np.cos(b[2]).sum())])


def _square(N=100, priors=("x0",)):
    """four poses on a 10 m square, odometry + the closing factor (the loop is consistent: the answer is the square itself)"""
    fg = R.initfg(N)
    truth = {"x0": (0, 0, 0), "x1": (10, 0, np.pi / 2), "x2": (10, 10, np.pi), "x3": (0, 10, -np.pi / 2)}
    for l in truth:
        fg.addVariable(l, R.Pose2)
    for l in priors:
        fg.addFactor([l], R.PriorPose2(R.MvNormal(np.array(truth[l], float), np.diag([0.01, 0.01, 0.0004]))))
    z = R.MvNormal(np.array([10.0, 0.0, np.pi / 2]), np.diag([0.04, 0.04, 0.0025]))
    for a, b in (("x0", "x1"), ("x1", "x2"), ("x2", "x3"), ("x3", "x0")):
        fg.addFactor([a, b], R.Pose2Pose2(z))
    return fg, truth


def test_a_consistent_loop_is_recovered_and_passes_pool():
    fg, truth = _square()
    es = RelativeEliminationSolver(fg, backend=OracleTreeBackend(R))
    st = es.stats()
    assert st["factor_edges"] == 4 and st["prior_variables"] == 1 and st["compositions"] >= 1 and st["merges"] >= 1, st   # the loop closes in a merge
    es.solve(R.make_opts(N=fg.N, seed=3), passes=3)
    assert es.passes_pooled == 3
    for l, t in truth.items():
        d = _mean(es, l) - np.array(t, float)
        d[2] = np.arctan2(np.sin(d[2]), np.cos(d[2]))
        assert np.abs(d[:2]).max() < 0.25 and abs(d[2]) < 0.05, (l, d)
    # the closing factor halves the variance of the far corner against dead reckoning along one side
    assert es.store.get("x2")[0].std() < 0.5


def test_two_priors_are_both_used():
    fg, truth = _square(priors=("x0", "x2"))
    es = RelativeEliminationSolver(fg, backend=OracleTreeBackend(R))
    assert es.stats()["prior_variables"] == 2
    es.solve(R.make_opts(N=fg.N, seed=5))
    for l, t in truth.items():
        d = _mean(es, l) - np.array(t, float)
        assert np.abs(d[:2]).max() < 0.2, (l, d)
    assert np.hypot(*(es.store.get("x2")[:2].std(axis=1))) < 0.2          # x2 carries its own prior: tight


def test_scope_and_errors():
    fg, _ = _square()
    assert RelativeEliminationSolver.covers(fg)
    hx = R.generateGraph_Hexagonal(N=16)                                  # landmark + bearing-range factors
    assert not RelativeEliminationSolver.covers(hx) and "l1" in RelativeEliminationSolver.covers(hx, why=True)
    with pytest.raises(TypeError):
        RelativeEliminationSolver(hx, backend=OracleTreeBackend(R))
    fg2, _ = _square(priors=())
    assert RelativeEliminationSolver.covers(fg2, why=True) == "no prior"
    fg3, _ = _square()
    fg3.addVariable("x9", R.Pose2)                                         # connected to nothing
    with pytest.raises(ValueError):
        RelativeEliminationSolver(fg3, backend=OracleTreeBackend(R))


def test_structures_differ_and_rounds_are_independent_sets():
    fg = R.synth_manhattan(P=400, loops=200, seed=3, N=16)
    es = RelativeEliminationSolver(fg, backend=OracleTreeBackend(R), structures=2)
    nbr = {v: set() for v in fg.variables}
    for _, ls, _ in fg.factors:
        if len(ls) == 2:
            nbr[ls[0]].add(ls[1]); nbr[ls[1]].add(ls[0])
    # the first round of either structure eliminates mutually non-adjacent variables of the ORIGINAL graph
    for sched in es.schedules:
        down_last = [spec for kind, spec in sched if kind == "plan"][-1]      # the last plan of the schedule = the first round, back-substituted
        first = [l for l in down_last.order if l in fg.variables]
        assert len(first) > 20
        for a in first:
            assert not (nbr[a] & set(first)), a
    assert [len(s) for s in es.schedules][0] > 10
