"""`multihypo=[1, w1, w2]` bearing-range factors (IIF addFactor! kwarg; test/testMultimodalRangeBearing.jl:53,108):
HIP path vs oracle, and the reference test's statistical expectations at the convolution level."""
import numpy as np
import pytest

import oracle as ro

pytestmark = pytest.mark.gpu
R = None


@pytest.fixture(scope="module", autouse=True)
def _pkg():
    global R
    import rome_jl_amd
    R = rome_jl_amd
    R.default_context()
    yield


def _wd(a, b):
    d = a - b
    d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    return d


@pytest.mark.parametrize("direction", [0, 1])
@pytest.mark.parametrize("solver", [0, 1])
def test_multihypo_vs_oracle(direction, solver):
    rng = np.random.default_rng(5 + direction)
    C_, N = 11, 100
    mu = np.stack([rng.uniform(-3, 3, C_), rng.uniform(8, 25, C_)], 1); sigma = np.stack([rng.uniform(0.01, 0.1, C_), rng.uniform(0.2, 1.0, C_)], 1)
    pose = rng.standard_normal((C_, 3, N)) * np.array([0.3, 0.3, 0.1])[None, :, None] + rng.standard_normal((C_, 3, 1)) * [[10], [10], [3]]
    l1 = rng.standard_normal((C_, 2, N)) * 0.5 + rng.standard_normal((C_, 2, 1)) * 15
    l2 = rng.standard_normal((C_, 2, N)) * 0.5 + rng.standard_normal((C_, 2, 1)) * 15
    w = rng.uniform(0.2, 0.8, C_)
    o = R.make_opts(N=N, solver=solver, seed=17, stream_offset=5)
    oo = ro.make_opts(N=N, solver=solver, seed=17, stream_offset=5)
    if direction == 1:   # solve the pose: the landmark is drawn per particle
        out, st = R.conv_pose2point2br(o, 1, mu, sigma, l1, pose, alt=l2, hypo_w=w, want_status=True)
        ref, rst = ro.conv_pose2point2br(oo, 1, mu, sigma, np.concatenate([l1, l2]), pose, np.arange(C_), np.arange(C_),
                                         alt_var=C_ + np.arange(C_), hypo_w=w, want_status=True)
        assert np.abs(_wd(out, ref)).max() < 1e-8 and (st == rst).all()
        # each pose proposal satisfies the sighting against ONE of the two landmarks, in proportion ~w
        d1 = np.hypot(out[:, 0] - l1[:, 0], out[:, 1] - l1[:, 1]); d2 = np.hypot(out[:, 0] - l2[:, 0], out[:, 1] - l2[:, 1])
        near1 = np.abs(d1 - mu[:, 1:2]) < 6 * sigma[:, 1:2]; near2 = np.abs(d2 - mu[:, 1:2]) < 6 * sigma[:, 1:2]
        assert (near1 | near2).mean() > 0.999
        excl = (near1 ^ near2).mean(axis=1) > 0.9          # rows where the two rings do not overlap
        f1 = (near1 & ~near2).sum(axis=1) / np.maximum((near1 ^ near2).sum(axis=1), 1)
        assert excl.sum() >= 3 and np.abs(f1[excl] - w[excl]).max() < 0.25
    else:                # solve landmark l1: particles of the other hypothesis only get spreadNH entropy
        out = R.conv_pose2point2br(o, 0, mu, sigma, pose, l1, alt=l2, hypo_w=w)
        ref = ro.conv_pose2point2br(oo, 0, mu, sigma, pose, np.concatenate([l1, l2]), np.arange(C_), np.arange(C_),
                                    alt_var=C_ + np.arange(C_), hypo_w=w)
        assert np.abs(out - ref).max() < 1e-8
        rngs = np.hypot(out[:, 0] - pose[:, 0], out[:, 1] - pose[:, 1])
        on_ring = np.abs(rngs - mu[:, 1:2]) < 6 * sigma[:, 1:2]
        assert (on_ring.mean(axis=1) > w - 0.25).all()      # the selected share is on the ring (the rest may land there by chance)


def test_reference_multimodal_setup_statistics():
    """test/testMultimodalRangeBearing.jl:27-82 at the convolution level: l1 ~ N((10,0),I), l2 ~ N((30,0),I),
    p2br = (bearing N(0,0.1), range N(20,1)), multihypo=[1;0.5;0.5]: pose proposals land on the 20 m rings around BOTH
    landmarks; with the heading near 0 that is x ≈ -10 and x ≈ +10 (the reference asserts ≥ 2 of its 4 x-windows are hit)."""
    N = 100
    rng = np.random.default_rng(1)
    fg = R.initfg(N)
    fg.addVariable("l1", R.Point2); fg.addVariable("l2", R.Point2); fg.addVariable("x0", R.Pose2)
    fg.initVariable("l1", np.array([[10.0], [0.0]]) + rng.standard_normal((2, N)))
    fg.initVariable("l2", np.array([[30.0], [0.0]]) + rng.standard_normal((2, N)))
    fg.initVariable("x0", np.array([[0.0], [0.0], [0.0]]) + rng.standard_normal((3, N)) * np.array([[20.0], [1.0], [0.05]]))
    fl = fg.addFactor(["x0", "l1", "l2"], R.Pose2Point2BearingRange(R.Normal(0, 0.1), R.Normal(20.0, 1.0)), multihypo=[1.0, 0.5, 0.5])
    X0 = R.approxConv(fg, fl, "x0", solver=R.SOLVER_NEWTON, seed=3)
    wins = [(-20, 0), (0, 20), (20, 40), (40, 60)]
    assert sum(int(((X0[0] > a) & (X0[0] < b)).sum() > 0) for a, b in wins) >= 2
    d1 = np.hypot(X0[0] - fg.getVal("l1")[0], X0[1] - fg.getVal("l1")[1]); d2 = np.hypot(X0[0] - fg.getVal("l2")[0], X0[1] - fg.getVal("l2")[1])
    on1 = np.abs(d1 - 20) < 5; on2 = np.abs(d2 - 20) < 5
    assert 20 < on1.sum() < 80 and 20 < on2.sum() < 80 and (on1 | on2).all()
    # landmark direction (second testset of the reference, :108-130): some but not all of the l2 proposals sit 20 m from x0
    fg.initVariable("x0", rng.standard_normal((3, N)) * np.array([[1.0], [1.0], [0.01]]))
    L2 = R.approxConv(fg, fl, "l2", solver=R.SOLVER_NEWTON, seed=4)
    m = ((L2[0] > 10) & (L2[0] < 30)).sum()
    assert 5 < m < 95


def test_multihypo_in_device_graph_solve_vs_oracle():
    from solve_ref import solve_ref
    N = 100
    rng = np.random.default_rng(2)
    fg = R.initfg(N)
    fg.addVariable("x0", R.Pose2); fg.addFactor(["x0"], R.PriorPose2(R.MvNormal([0.0, 0, 0], np.diag([1.0, 1.0, 0.01]) ** 2)))
    fg.addVariable("x1", R.Pose2); fg.addFactor(["x0", "x1"], R.Pose2Pose2(R.MvNormal([5.0, 0, 0], np.diag([0.1, 0.1, 0.01]) ** 2)))
    fg.addVariable("l1", R.Point2); fg.addVariable("l2", R.Point2)
    fg.addFactor(["x0", "l1"], R.Pose2Point2BearingRange(R.Normal(0, 0.05), R.Normal(20.0, 0.5)))
    fg.addFactor(["x1", "l1", "l2"], R.Pose2Point2BearingRange(R.Normal(0, 0.05), R.Normal(15.0, 0.5)), multihypo=[1.0, 0.7, 0.3])
    R.dead_reckon_init(fg, seed=3)
    fg.initVariable("l2", np.array([[35.0], [3.0]]) + rng.standard_normal((2, N)))
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    assert dg.tab["br"]["F"] == 2 and dg.tab["br"]["F0"] == 3 and dg.tab["br"]["mh"]
    dg.solve(R.make_opts(N=N, solver=1, seed=21), n_sweeps=3)
    b2, bl = solve_ref(R, fg, 3, N, seed=21)
    g2 = dg.bel[R.Pose2].cpu().numpy(); gl = dg.bel[R.Point2].cpu().numpy()
    assert np.mean(np.abs(_wd(g2, b2)) < 1e-7) > 0.97 and np.mean(np.abs(gl - bl) < 1e-7) > 0.97
    m2, _ = R.belief_stats(g2); mo, _ = R.belief_stats(b2)
    assert np.abs(m2[:, :2] - mo[:, :2]).max() < 1e-3


@pytest.mark.parametrize("kind", ["p2p2", "p3p3"])
def test_nullhypo_vs_oracle_and_fraction(kind):
    """IIF nullhypo=0.5 (test/testPose3Pose3NH.jl:118): about half of the proposals follow the factor, the rest keep
    their start value plus spreadNH entropy.  GPU = oracle on the same Philox streams."""
    rng = np.random.default_rng(3)
    C_, N = 6, 100
    if kind == "p2p2":
        mu = np.tile([-35.0, 0, 0], (C_, 1)); cov = np.tile(np.diag([0.5, 0.5, 0.01]), (C_, 1, 1))
        fixed = rng.standard_normal((C_, 3, N)) * np.array([1.0, 1.0, 0.05])[None, :, None] + np.array([50.0, 0, 0])[None, :, None]
        target = rng.standard_normal((C_, 3, N)) * np.array([1.0, 1.0, 0.05])[None, :, None]
        dirs = np.zeros(C_, np.int32)
        out = R.conv_pose2pose2(R.make_opts(N=N, solver=1, seed=9, nullhypo=0.5), mu, cov, fixed, target, dirs=dirs)
        L = np.array([ro.cholesky_lower(c) for c in cov])
        ref = ro.conv_pose2pose2(ro.make_opts(N=N, solver=1, seed=9, nullhypo=0.5), mu, L, np.concatenate([fixed, target]),
                                 np.arange(C_), C_ + np.arange(C_), dirs)
        d = out - ref; d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
        assert np.abs(d).max() < 1e-9
        follows = np.abs(out[:, 0] - 15.0) < 6.0           # 50 - 35 = 15 along x
    else:
        mu = np.tile([-35.0, 0, 0, 0, 0, 0], (C_, 1)); cov = np.tile(np.diag([0.5] * 3 + [0.01] * 3), (C_, 1, 1))
        sc = np.array([1.0, 1.0, 1.0, 0.05, 0.05, 0.05])[None, :, None]
        fixed = rng.standard_normal((C_, 6, N)) * sc + np.array([50.0, 0, 0, 0, 0, 0])[None, :, None]
        target = rng.standard_normal((C_, 6, N)) * sc
        dirs = np.zeros(C_, np.int32)
        out = R.conv_pose3pose3(R.make_opts(N=N, solver=1, seed=9, nullhypo=0.5), mu, cov, fixed, target, dirs=dirs)
        L = np.array([ro.cholesky_lower(c) for c in cov])
        ref = ro.conv_pose3pose3(ro.make_opts(N=N, solver=1, seed=9, nullhypo=0.5), mu, L, np.concatenate([fixed, target]),
                                 np.arange(C_), C_ + np.arange(C_), dirs)
        assert np.abs(out - ref).max() < 1e-8
        follows = np.abs(out[:, 0] - 15.0) < 6.0
    frac = follows.mean(axis=1)
    assert (frac > 0.3).all() and (frac < 0.7).all()
    # the null-hypothesis share stays spread around the START belief (x ≈ 0), not at the factor's solution
    assert (np.abs(out[:, 0][~follows]).mean() < 8.0)


@pytest.mark.parametrize("direction", [0, 1])
@pytest.mark.parametrize("solver", [0, 1])
def test_nullhypo_bearingrange_vs_oracle(direction, solver):
    """nullhypo on Pose2Point2BearingRange, both directions: GPU = oracle; the null share keeps its start value (+ entropy)."""
    rng = np.random.default_rng(11)
    C_, N = 5, 100
    mu = np.tile([0.3, 12.0], (C_, 1)); sigma = np.tile([0.03, 0.4], (C_, 1))
    poses = rng.standard_normal((C_, 3, N)) * np.array([0.3, 0.3, 0.05])[None, :, None]
    lms = rng.standard_normal((C_, 2, N)) * 0.4 + np.array([40.0, -25.0])[None, :, None]   # start belief far from the solution
    fixed, target = (poses, lms) if direction == 0 else (lms, poses + np.array([30.0, 10.0, 0.4])[None, :, None])
    o = R.make_opts(N=N, solver=solver, seed=5, nullhypo=0.4)
    out = R.conv_pose2point2br(o, direction, mu, sigma, fixed, target)
    ref = ro.conv_pose2point2br(ro.make_opts(N=N, solver=solver, seed=5, nullhypo=0.4), direction, mu, sigma, fixed, target,
                                np.arange(C_), np.arange(C_), factor=np.arange(C_))
    d = out - ref
    if direction == 1:
        d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    assert np.abs(d).max() < 1e-8
    if direction == 0:
        follows = np.hypot(out[:, 0], out[:, 1]) < 16.0          # landmarks 12 m from poses near the origin
    else:
        follows = np.abs(np.hypot(out[:, 0] - 40.0, out[:, 1] + 25.0) - 12.0) < 2.5   # poses on the 12 m ring about the landmark
    frac = follows.mean(axis=1)
    assert (frac > 0.4).all() and (frac < 0.8).all(), frac


def test_pose3pose3_nullhypo_against_reference_validation_samples():
    """test/testPose3Pose3NH.jl:20-145 with the reference's own validation samples (its data files test/X1ptst.csv, X2ptst.csv,
    copied to tests/golden/): x1 -25-> x2 -25-> x3 chain, then Pose3Pose3([-35,0,...]) over [x3, x1] with nullhypo=0.5.  The
    proposals on x1 are a mixture of the factor's solution (x ≈ 15) and of x1's own belief plus spreadNH entropy (x ≈ 0);
    on x3: x ≈ 35 and x ≈ 50.  Mixture fractions, mode locations and the order of magnitude of the null share's
    translation spread must match the reference samples (their rotation spread predates the scalar-spread entropy of current
    IncrementalInference and is not compared)."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    X1ref = np.loadtxt(os.path.join(gold, "pose3pose3nh_X1ptst.csv"), delimiter=",")
    X2ref = np.loadtxt(os.path.join(gold, "pose3pose3nh_X2ptst.csv"), delimiter=",")
    N = 100
    cov = np.diag(np.square([1, 1, 1, 0.01, 0.01, 0.01]))
    fg = R.initfg(N)
    fg.addVariable("x1", R.Pose3)
    f1 = fg.addFactor(["x1"], R.PriorPose3(R.MvNormal(np.zeros(6), cov)))
    fg.initVariable("x1", R.approxConv(fg, f1, "x1", seed=1))
    fg.addVariable("x2", R.Pose3); f12 = fg.addFactor(["x1", "x2"], R.Pose3Pose3(R.MvNormal([25.0, 0, 0, 0, 0, 0], cov)))
    fg.initVariable("x2", R.approxConv(fg, f12, "x2", seed=2))
    fg.addVariable("x3", R.Pose3); f23 = fg.addFactor(["x2", "x3"], R.Pose3Pose3(R.MvNormal([25.0, 0, 0, 0, 0, 0], cov)))
    fg.initVariable("x3", R.approxConv(fg, f23, "x3", seed=3))
    m, _ = R.belief_stats(np.stack([fg.getVal("x2"), fg.getVal("x3")]))
    assert np.allclose(m[0, :3], [25, 0, 0], atol=5.0) and np.allclose(m[1, :3], [50, 0, 0], atol=5.0)      # :101-105
    f31 = fg.addFactor(["x3", "x1"], R.Pose3Pose3(R.MvNormal([-35.0, 0, 0, 0, 0, 0], cov)))
    X1 = R.approxConv(fg, f31, "x1", seed=4, nullhypo=0.5)
    X3 = R.approxConv(fg, f31, "x3", seed=5, nullhypo=0.5)
    for got, ref, active_x, null_x in ((X1, X1ref, 15.0, 0.0), (X3, X2ref, 35.0, 50.0)):
        ga, ra = np.abs(got[0] - active_x) < 7, np.abs(ref[0] - active_x) < 7
        assert abs(ga.mean() - 0.5) < 0.2 and abs(ra.mean() - 0.5) < 0.2
        assert np.abs(got[:3, ga].mean(axis=1) - ref[:3, ra].mean(axis=1)).max() < 2.5
        assert np.abs(got[:3, ~ga].mean(axis=1) - ref[:3, ~ra].mean(axis=1)).max() < 2.5
        assert abs(got[0, ~ga].mean() - null_x) < 2.5
        ratio = got[:3, ~ga].std(axis=1) / ref[:3, ~ra].std(axis=1)
        assert (ratio > 1 / 3.0).all() and (ratio < 3.0).all(), ratio
        # the factor-following share is as tight as the reference's in translation
        ratio_a = got[:3, ga].std(axis=1) / ref[:3, ra].std(axis=1)
        assert (ratio_a > 1 / 3.0).all() and (ratio_a < 3.0).all(), ratio_a


def test_priorpoint2_nonparametric_sampling_and_solve_loop_vs_oracle():
    """PriorPoint2 (src/factors/Point2D.jl:8-18) in the non-parametric path: the per-factor entry = the oracle's sampler; in the solve
    loop a landmark prior contributes one proposal row to the product of its landmark (device loop = oracle loop), and it pins the
    landmark: the setup of test/testBearingRange2D.jl:314-330 (landmark prior at (20, 0), a sighting from x0 straight ahead at 20 m)."""
    from solve_ref import solve_ref
    import oracle as ro
    N = 100
    cov = np.array([[0.04, 0.01], [0.01, 0.09]])
    got = R.sample_priorpoint2(R.make_opts(N=N, seed=8, stream_offset=77), [[20.0, 1.0]], [cov])[0]
    ref = ro.sample_priorpoint2(ro.make_opts(N=N, seed=8, stream_offset=77), [[20.0, 1.0]], [ro.cholesky_lower(cov)])[0]
    assert np.abs(got - ref).max() < 1e-12 and abs(got[0].mean() - 20.0) < 0.1 and abs(got[1].std() - 0.3) < 0.08
    fg = R.initfg(N)
    fg.addVariable("x0", R.Pose2); fg.addFactor(["x0"], R.PriorPose2(R.MvNormal([0.0, 0, 0], np.diag([0.5, 0.5, 0.05]) ** 2)))
    fg.addVariable("l1", R.Point2); fg.addFactor(["l1"], R.PriorPoint2(R.MvNormal([20.0, 0.0], np.diag([0.1, 0.1]) ** 2)))
    fg.addFactor(["x0", "l1"], R.Pose2Point2BearingRange(R.Normal(0, 0.1), R.Normal(20.0, 0.1)))
    assert np.allclose(R.approxConv(fg, "l1f1", "l1", seed=3).mean(1), [20.0, 0.0], atol=0.05)
    R.initAll(fg, seed=4)
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    assert dg.n_prop[R.Point2] == 2                      # the sighting's proposal + the prior's
    dg.solve(R.make_opts(N=N, solver=1, seed=31), n_sweeps=4, bandwidth="lcv", product="gibbs")
    b2, bl = solve_ref(R, fg, 4, N, seed=31, bandwidth="lcv", product="gibbs")
    gl = dg.bel[R.Point2].cpu().numpy(); g2 = dg.bel[R.Pose2].cpu().numpy()
    assert np.mean(np.abs(gl - bl) < 1e-6) > 0.9 and np.mean(np.abs(_wd(g2, b2)) < 1e-6) > 0.9
    assert np.abs(gl[0].mean(1) - [20.0, 0.0]).max() < 0.2 and gl[0].std(1).max() < 0.3      # pinned by its prior
    assert np.abs(g2[0, :2].mean(1)).max() < 0.5


def test_nullhypo_factors_in_the_device_graph_tables():
    """`addFactor(..., nullhypo=p)` (test/testPose3Pose3NH.jl:118) on the GRAPH path: DeviceGraph carries one probability per table row;
    every row of a whole-graph sweep = the per-factor call with the factor's nullhypo, bit for bit (Pose2Pose2, both bearing-range
    directions, the fused sweep entry), and DeviceGraph.solve = the oracle's restatement of the loop."""
    from solve_ref import solve_ref
    N = 100
    rng = np.random.default_rng(5)

    def graph():
        fg = R.initfg(N)
        fg.addVariable("x0", R.Pose2); fg.addFactor(["x0"], R.PriorPose2(R.MvNormal([0.0, 0, 0], np.diag([0.1, 0.1, 0.01]) ** 2)))
        cov = np.diag([0.1, 0.1, 0.01]) ** 2
        for k in range(1, 5):
            fg.addVariable("x%d" % k, R.Pose2)
            fg.addFactor(["x%d" % (k - 1), "x%d" % k], R.Pose2Pose2(R.MvNormal([5.0, 0, 0.3], cov)), nullhypo=0.3 if k == 2 else None)
        fg.addFactor(["x4", "x0"], R.Pose2Pose2(R.MvNormal([-3.0, -9.0, -1.2], cov)), nullhypo=0.5)     # a doubtful loop closure
        fg.addVariable("l1", R.Point2)
        fg.addFactor(["x0", "l1"], R.Pose2Point2BearingRange(R.Normal(0, 0.05), R.Normal(10.0, 0.3)))
        fg.addFactor(["x3", "l1"], R.Pose2Point2BearingRange(R.Normal(1.0, 0.05), R.Normal(12.0, 0.3)), nullhypo=0.4)
        R.dead_reckon_init(fg, seed=3)
        return fg
    fg = graph()
    assert len(fg.nullhypo) == 3
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    assert dg.tab["p2p2"]["nh"] is not None and dg.tab["br"]["nh"] is not None
    o = R.make_opts(N=N, seed=13)
    dg.conv_step(o, 0)
    p2 = dg.prop[R.Pose2].cpu().numpy(); pl = dg.prop[R.Point2].cpu().numpy()
    pk = dg.packed
    C2 = dg.tab["p2p2"]["C"]
    for f, fl in enumerate(pk.p2p2["labels"]):
        labels = fg.getFactor(fl)[1]
        for d, tgt in ((0, labels[1]), (1, labels[0])):
            ref = R.approxConv(fg, fl, tgt, seed=13, stream_offset=dg.STREAM_P2P2 + 2 * f + d, nullhypo=fg.nullhypo.get(fl, 0.0))
            assert np.array_equal(p2[2 * f + d], ref), (fl, d)
    for k, fl in enumerate(pk.br["labels"]):
        labels = fg.getFactor(fl)[1]
        nh = fg.nullhypo.get(fl, 0.0)
        assert np.array_equal(p2[C2 + k], R.approxConv(fg, fl, labels[0], seed=13, stream_offset=dg.STREAM_BR1 + k, nullhypo=nh)), fl
        assert np.array_equal(pl[k], R.approxConv(fg, fl, labels[1], seed=13, stream_offset=dg.STREAM_BR0 + k, nullhypo=nh)), fl
    # the loop: device == oracle
    dg.solve(R.make_opts(N=N, solver=1, seed=21), n_sweeps=2, bandwidth="lcv", product="gibbs")
    b2, bl = solve_ref(R, graph(), 2, N, seed=21, bandwidth="lcv", product="gibbs")
    g2 = dg.bel[R.Pose2].cpu().numpy(); gl = dg.bel[R.Point2].cpu().numpy()
    assert np.mean(np.abs(_wd(g2, b2)) < 1e-6) > 0.9 and np.mean(np.abs(gl - bl) < 1e-6) > 0.9
