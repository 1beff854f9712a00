"""multihypo on a Pose2Pose2 factor (IIF `addFactor!(fg, [:a; :b1; :b2], Pose2Pose2(z), multihypo=[1; w; 1-w])`): the keyword is IIF's
and applies to any factor; every use inside the reference is on a bearing-range factor (test/testMultimodalRangeBearing.jl:53), whose
rule -- a categorical draw per particle, entropy for the particles of the other hypothesis -- is applied here unchanged.
Device (rome_conv_pose2pose2_mh through the C ABI) against the oracle (ro_conv_pose2pose2_mh) and against what the rule implies."""
import numpy as np
import pytest

import oracle as ro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _pkg():
    global R
    import rome_jl_amd
    R = rome_jl_amd
    R.default_context()
    yield


def _problem(N, rng):
    a = np.array([0.0, 0.0, 0.2])[:, None] + 0.05 * rng.standard_normal((3, N))
    b1 = np.array([1.0, 0.3, 0.3])[:, None] + 0.05 * rng.standard_normal((3, N))
    b2 = np.array([6.0, -4.0, -2.0])[:, None] + 0.05 * rng.standard_normal((3, N))
    mu = np.array([[1.0, 0.1, 0.1]]); cov = np.diag([0.01, 0.01, 0.0025])[None]
    return a, b1, b2, mu, cov


@pytest.mark.parametrize("solver", [0, 1, 2])
@pytest.mark.parametrize("N", [100, 257])
def test_device_equals_oracle_both_directions(solver, N):
    rng = np.random.default_rng(10 * N + solver)
    a, b1, b2, mu, cov = _problem(N, rng)
    L = np.array([ro.cholesky_lower(cov[0])])
    bel = np.stack([a, b1, b2])
    o = R.make_opts(N=N, solver=solver, seed=5, stream_offset=9)
    oo = ro.make_opts(N=N, solver=solver, seed=5, stream_offset=9)
    tol = 1e-9 if solver != 2 else 1e-5
    # direction 1: solve a, the fixed pose is b1 (p = 0.7) or b2 per particle
    got = R.conv_pose2pose2(o, mu, cov, b1[None], a[None], dirs=[1], alt=b2[None], hypo_w=[0.7])[0]
    ref = ro.conv_pose2pose2(oo, mu, L, bel, [1], [0], [1], alt_var=[2], hypo_w=[0.7])[0]
    d = got - ref; d[2] = np.arctan2(np.sin(d[2]), np.cos(d[2]))
    assert (np.abs(d).max(0) < tol).mean() > (0.999 if solver != 2 else 0.97)
    near_b1 = np.hypot(got[0] - 0.0, got[1] - 0.0) < 1.0            # a = b1 ⊖ z sits near the origin, a = b2 ⊖ z far away
    assert abs(near_b1.mean() - 0.7) < 4 * np.sqrt(0.21 / N) + 0.01
    # direction 0: solve b1 from a; particles drawn for b2 keep their value up to the hypothesis entropy
    got0 = R.conv_pose2pose2(o, mu, cov, a[None], b1[None], dirs=[0], alt=b2[None], hypo_w=[0.7])[0]
    ref0 = ro.conv_pose2pose2(oo, mu, L, bel, [0], [1], [0], alt_var=[2], hypo_w=[0.7])[0]
    d0 = got0 - ref0; d0[2] = np.arctan2(np.sin(d0[2]), np.cos(d0[2]))
    assert (np.abs(d0).max(0) < tol).mean() > (0.999 if solver != 2 else 0.97)
    nh = 3.0 * np.hypot(b1[0].mean() - b2[0].mean(), b1[1].mean() - b2[1].mean())   # spreadNH = 3 (IIF default)
    moved = np.abs(got0[:2] - b1[:2]).max(0)
    solved = moved < 1.0                                                           # (the entropy box is ±nh/2 ≈ ±9.6 wide)
    assert abs(solved.mean() - 0.7) < 4 * np.sqrt(0.21 / N) + 0.05
    assert np.abs(got0[:2] - b1[:2]).max() <= nh / 2 + 1e-9


def test_approxconv_on_a_graph():
    N = 200
    rng = np.random.default_rng(2)
    a, b1, b2, mu, cov = _problem(N, rng)
    fg = R.initfg(N)
    for l, v in (("a", a), ("b1", b1), ("b2", b2)):
        fg.addVariable(l, R.Pose2); fg.initVariable(l, v)
    fl = fg.addFactor(["a", "b1", "b2"], R.Pose2Pose2(R.MvNormal(mu[0], cov[0])), multihypo=[1.0, 0.5, 0.5])
    pa = R.approxConv(fg, fl, "a", seed=4)
    frac = (np.hypot(pa[0], pa[1]) < 1.0).mean()
    assert 0.35 < frac < 0.65                                   # two modes, one per hypothesis
    far = pa[:, np.hypot(pa[0], pa[1]) >= 1.0]
    assert np.abs(np.hypot(far[0] - 6.0, far[1] + 4.0) - np.hypot(1.0, 0.1)).max() < 0.6   # the b2 mode: |a - b2| = |z_t|
    pb2 = R.approxConv(fg, fl, "b2", seed=4)                     # solve the second candidate: about half the particles land on a ⊕ z
    assert 0.3 < (np.hypot(pb2[0] - 1.0, pb2[1] - 0.3) < 0.6).mean() < 0.7
    with pytest.raises(R.RomeError):
        R.conv_pose2pose2(R.make_opts(N=N), mu, cov, b1[None], a[None], dirs=[1], alt=b2[None], hypo_w=[1.5])


def _ring_with_ambiguous_closure(N):
    """seven poses around a ring, a prior on x0, and a loop closure from x6 whose other end is x0 (p = 0.7) or x3 (p = 0.3)"""
    fg = R.initfg(N)
    cov = np.diag([0.01, 0.01, 0.0025])
    fg.addVariable("x0", R.Pose2)
    fg.addFactor(["x0"], R.PriorPose2(R.MvNormal([0.0, 0.0, 0.0], np.diag([0.01, 0.01, 0.0025]))))
    for k in range(1, 7):
        fg.addVariable("x%d" % k, R.Pose2)
        fg.addFactor(["x%d" % (k - 1), "x%d" % k], R.Pose2Pose2(R.MvNormal([2.0, 0.0, np.pi / 3.5], cov)))
    fl = fg.addFactor(["x6", "x0", "x3"], R.Pose2Pose2(R.MvNormal([2.0, 0.0, np.pi / 3.5], cov)), multihypo=[1.0, 0.7, 0.3])
    R.dead_reckon_init(fg, seed=5)
    return fg, fl


def test_whole_graph_tables_and_solve_loop_equal_the_oracle():
    """The graph tables carry the hypotheses (alternative and probability per row, one more row for the second candidate): the
    device sweep equals the per-factor path row by row, and DeviceGraph.solve equals the oracle's restatement of the same loop."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from solve_ref import solve_ref
    N, S = 100, 4
    fg, fl = _ring_with_ambiguous_closure(N)
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    tb = dg.tab["p2p2"]
    assert tb["mh"] and tb["E"] == 1 and tb["C"] == 2 * 7 + 1 + 1
    rows = tb["rows4"].cpu().numpy(); alt = tb["alt"].cpu().numpy(); w = tb["w"].cpu().numpy()
    i = lambda l: dg.packed.index[l]
    f = int(rows[-1, 0])
    assert list(rows[-1]) == [f, 0, i("x6"), i("x3")] and alt[-1] == i("x0") and abs(w[-1] - 0.3) < 1e-15       # the second candidate's row
    assert list(rows[2 * f]) == [f, 0, i("x6"), i("x0")] and alt[2 * f] == i("x3") and abs(w[2 * f] - 0.7) < 1e-15
    assert list(rows[2 * f + 1]) == [f, 1, i("x0"), i("x6")] and alt[2 * f + 1] == i("x3")
    o = R.make_opts(N=N, solver=1, seed=77)
    prop = dg.sweep_pose2pose2(o).cpu().numpy()
    bel = dg.bel[R.Pose2].cpu().numpy()
    for r in (2 * f, 2 * f + 1, len(rows) - 1):                                  # the three rows of the ambiguous factor, one by one
        orow = R.make_opts(N=N, solver=1, seed=77, stream_offset=r)
        fac = fg.getFactor(fl)[2]
        one = R.conv_pose2pose2(orow, [fac.Z.mu], [fac.Z.cov], bel[rows[r, 2]][None], bel[rows[r, 3]][None], dirs=[int(rows[r, 1])],
                                alt=bel[alt[r]][None], hypo_w=[w[r]])[0]
        assert np.array_equal(one, prop[r])
    dg.solve(o, n_sweeps=S)
    b2, _ = solve_ref(R, fg, S, N, seed=77)
    d = dg.bel[R.Pose2].cpu().numpy() - b2; d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    assert np.mean(np.abs(d) < 1e-8) > 0.95
    # the true closure (x0) dominates: x6 stays where the odometry chain and the 0.7-hypothesis put it
    m, _ = dg.belief_stats(R.Pose2)
    assert np.all(np.isfinite(m.cpu().numpy()))


def test_target_sharded_sweep_carries_the_hypotheses_through_the_sort():
    """Strong-scaling driver on the ambiguous ring: rows are sorted by target variable and the alternative / probability columns
    travel with them; a sorted row is the per-factor call with its own Philox stream (= its sorted position)."""
    import torch
    from rome_jl_amd.distributed import TargetShardedSweep
    N = 64
    fg, fl = _ring_with_ambiguous_closure(N)
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    o = R.make_opts(N=N, solver=1, seed=3)
    sh = TargetShardedSweep(dg, o, None, 1, 0)
    sh.step(); sh.wait(); torch.cuda.synchronize()
    prop = sh.prop.cpu().numpy(); rows = sh.rows4.cpu().numpy(); alt = sh.alt.cpu().numpy(); w = sh.w.cpu().numpy()
    bel = dg.bel[R.Pose2].cpu().numpy()
    fac = fg.getFactor(fl)[2]
    hit = 0
    for j in range(len(rows)):
        if alt[j] < 0:
            continue
        orow = R.make_opts(N=N, solver=1, seed=3, stream_offset=j)
        one = R.conv_pose2pose2(orow, [fac.Z.mu], [fac.Z.cov], bel[rows[j, 2]][None], bel[rows[j, 3]][None], dirs=[int(rows[j, 1])],
                                alt=bel[alt[j]][None], hypo_w=[w[j]])[0]
        assert np.array_equal(one, prop[j]); hit += 1
    assert hit == 3 and np.all(np.diff(rows[:, 3]) >= 0)
