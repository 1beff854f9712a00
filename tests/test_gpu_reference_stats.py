"""The reference's statistical convolution tests (SURVEY §8(c) "Conv-stat" rows) run against the HIP path:
test/testBasicPose2Conv.jl:8-44 and test/TestPoseAndPoint2Constraints.jl:10-42,88-121 (convolution level)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
R = None


@pytest.fixture(scope="module", autouse=True)
def _pkg():
    global R
    import rome_jl_amd
    R = rome_jl_amd
    R.default_context()
    yield


@pytest.mark.parametrize("solver", [0, 1, 2])
def test_basic_pose2_convolution(solver):
    # testBasicPose2Conv.jl: x0 ~ 100 points 0.01*randn(3) about identity; Pose2Pose2(MvNormal([10;0;pi], 0.1*I))
    N = 100
    rng = np.random.default_rng(0)
    fg = R.initfg(N)
    fg.addVariable("x0", R.Pose2); fg.initVariable("x0", 0.01 * rng.standard_normal((3, N)))
    fg.addVariable("x1", R.Pose2)
    fl = fg.addFactor(["x0", "x1"], R.Pose2Pose2(R.MvNormal([10, 0, np.pi], 0.1 * np.eye(3))))
    X1 = R.approxConv(fg, fl, "x1", solver=solver, seed=123)
    mean, sd = R.belief_stats(X1[None])
    Rm = np.array([[np.cos(mean[0, 2]), -np.sin(mean[0, 2])], [np.sin(mean[0, 2]), np.cos(mean[0, 2])]])
    assert np.allclose(mean[0, :2], [10, 0], atol=0.2)                       # :28
    assert np.allclose(Rm, [[-1, 0], [0, -1]], atol=0.2)                     # :29
    assert np.allclose(np.diag(sd[0] ** 2), 0.15 * np.eye(3), atol=0.4)      # :31
    psi = X1[2]
    assert 20 < (psi > 2).sum() and 20 < (psi < -2).sum()                    # :38-39  bimodal at ±π
    # :41 asserts no sample in (-2, 2): with σ_θ = √0.1 that window starts 3.6 σ from ±π, i.e. the reference's own test
    # fails ~3 % of the time; allow one stray sample here.  :43 (|ψ| ≤ π on-manifold) is exact.
    assert ((psi > -2) & (psi < 2)).sum() <= 1 and (np.abs(psi) > 3.15).sum() < 1


def test_pose_and_point_convolutions():
    # TestPoseAndPoint2Constraints.jl:19-42 and :88-121
    N = 100
    fg = R.initfg(N)
    fg.addVariable("x0", R.Pose2)
    f1 = fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), np.diag([0.03, 0.03, 0.001]))))
    fg.initVariable("x0", R.approxConv(fg, f1, "x0", seed=1))
    fg.addVariable("x1", R.Pose2)
    f2 = fg.addFactor(["x0", "x1"], R.Pose2Pose2(R.MvNormal([50.0, 0.0, np.pi / 2], np.diag([3.0, 3.0, 0.01]))))
    pts = R.approxConv(fg, f2, "x1", seed=2)
    mean, _ = R.belief_stats(pts[None])
    assert np.allclose(mean[0, :2], [50, 0], atol=1)                         # :40
    th = mean[0, 2]
    assert np.allclose([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]], [[0, -1], [1, 0]], atol=0.5)   # :41
    # bearing-range with a Uniform(-pi, pi) bearing: all landmark proposals lie in a ring around (0,0)
    fg.addVariable("l1", R.Point2)
    f4 = fg.addFactor(["x0", "l1"], R.Pose2Point2BearingRange(R.Uniform(-np.pi, np.pi), R.Normal(10.0, 1.0)))
    lp = R.approxConv(fg, f4, "l1", seed=3)
    r = np.hypot(lp[0], lp[1])
    assert (r < 5.0).sum() == 0 and (r < 15.0).sum() == N                    # :104-105
    assert np.ptp(np.arctan2(lp[1], lp[0])) > 5.0                            # bearings cover the circle
    # ... and the pose direction runs (the reference only checks it does not throw, :108)
    fg.initVariable("l1", lp)
    xp = R.approxConv(fg, f4, "x0", seed=4)
    assert np.isfinite(xp).all()


@pytest.mark.parametrize("case", ["yaw", "pitch_from_pi"])
@pytest.mark.parametrize("solver", [0, 1])
def test_pose3_hexagon_chain_initialisation(case, solver):
    """test/testSimpleHexPose3.jl: a PriorPose3 and six Pose3Pose3 legs of 10 m turning π/3 (about z from identity, or about y
    from a start yawed by π), initialised leg by leg (`initAll!`): the chain closes on itself -- x6 comes back to x0 --
    and x3 is the opposite corner of the hexagon."""
    from scipy.spatial.transform import Rotation as Rot
    N = 100
    fg = R.initfg(N)
    fg.addVariable("x0", R.Pose3)
    yaw0 = 0.0 if case == "yaw" else np.pi
    f0 = fg.addFactor(["x0"], R.PriorPose3(R.MvNormal([0.0, 0, 0, 0, 0, yaw0], np.diag(np.square([0.1, 0.1, 0.1, 0.01, 0.01, 0.01])))))
    fg.initVariable("x0", R.approxConv(fg, f0, "x0", seed=1))
    turn = [0, 0, np.pi / 3] if case == "yaw" else [0, np.pi / 3, 0]
    for i in range(6):
        fg.addVariable("x%d" % (i + 1), R.Pose3)
        fl = fg.addFactor(["x%d" % i, "x%d" % (i + 1)],
                          R.Pose3Pose3(R.MvNormal([10.0, 0, 0] + turn, np.diag(np.square([0.5, 0.5, 0.5, 0.05, 0.05, 0.05])))))
        fg.initVariable("x%d" % (i + 1), R.approxConv(fg, fl, "x%d" % (i + 1), solver=solver, seed=10 + i))
    mean, sd = R.belief_stats(np.stack([fg.getVal("x%d" % i) for i in range(7)]))
    # exact (noise-free) chain
    T = [np.eye(4)]; T[0][:3, :3] = Rot.from_rotvec([0, 0, yaw0]).as_matrix()
    D = np.eye(4); D[:3, :3] = Rot.from_rotvec(turn).as_matrix(); D[:3, 3] = [10, 0, 0]
    for i in range(6):
        T.append(T[-1] @ D)
    assert np.allclose(T[6][:3, 3], 0, atol=1e-9)
    for i in range(7):
        assert np.linalg.norm(mean[i, :3] - T[i][:3, 3]) < 0.5 + 4 * np.linalg.norm(sd[i, :3]) / np.sqrt(N), (i, mean[i], T[i][:3, 3])
        dR = Rot.from_rotvec(mean[i, 3:]).as_matrix().T @ T[i][:3, :3]
        assert np.linalg.norm(Rot.from_matrix(dR).as_rotvec()) < 0.15, i
    assert np.linalg.norm(mean[3, :3] - T[3][:3, 3]) < 3.0 and np.linalg.norm(T[3][:3, 3]) > 19.9
    # uncertainty grows along the chain
    assert np.linalg.norm(sd[6, :3]) > np.linalg.norm(sd[1, :3]) > np.linalg.norm(sd[0, :3])


def test_bearingrange_sampling_windows():
    """test/testBearingRange2D.jl:12-41: 100 samples of Pose2Point2BearingRange(Normal(0,0.1), Normal(20,1)) have bearing mean within 0.1,
    std in (0.05, 0.2), range mean within 1.0 of 20, std in (0.5, 1.5).  The in-kernel sampler is observed through the closed-form
    convolution from the identity pose: landmark = ρ (cos b, sin b)."""
    N = 100
    for seed in (1, 2, 3):
        o = R.make_opts(N=N, solver=R.SOLVER_CLOSED_FORM, seed=seed)
        out = R.conv_pose2point2br(o, 0, [[0.0, 20.0]], [[0.1, 1.0]], np.zeros((1, 3, N)), np.zeros((1, 2, N)))[0]
        b, rho = np.arctan2(out[1], out[0]), np.hypot(out[0], out[1])
        assert abs(b.mean()) < 0.1 and 0.05 < b.std(ddof=1) < 0.2
        assert abs(rho.mean() - 20.0) < 1.0 and 0.5 < rho.std(ddof=1) < 1.5


@pytest.mark.parametrize("solver", [1, 2])
def test_pose_from_landmark_from_a_degenerate_start_spreads_around_the_ring(solver):
    """IIF calcStdBasicSpread: "if no std yet, set to 1" -- approxConv / initVariable start an uninitialised target from identical
    points.  For the bearing-range POSE direction (2 equations, 3 unknowns: a ring of roots around the landmark) the jitter of that
    start decides where on the ring a particle lands: without the fallback every particle would sit on the one ray through the
    common start point.  The jitter here is the full-width (32-bit) uniform of the oracle, not the cheap one of the unique-root
    kernels; device == oracle, the ring is covered over a wide arc, and the result is independent of N's slot packing."""
    import oracle as ro
    N = 100
    lm = np.array([[10.0], [0.0]]) + 0.05 * np.random.default_rng(1).standard_normal((2, N))
    o = R.make_opts(N=N, solver=solver, seed=12)
    out = R.conv_pose2point2br(o, 1, [[0.3, 10.0]], [[0.03, 0.5]], lm[None], np.zeros((1, 3, N)))[0]
    ref = ro.conv_pose2point2br(ro.make_opts(N=N, solver=solver, seed=12), 1, [[0.3, 10.0]], [[0.03, 0.5]], lm[None], np.zeros((1, 3, N)), [0], [0])[0]
    d = out - ref; d[2] = np.arctan2(np.sin(d[2]), np.cos(d[2]))
    assert np.median(np.abs(d)) < 1e-8 and (np.abs(d).max(axis=0) < 1e-5).mean() > 0.95
    ang = np.arctan2(out[1] - lm[1], out[0] - lm[0])           # where on the ring around the landmark
    rng_ = np.hypot(out[0] - lm[0], out[1] - lm[1])
    assert np.abs(rng_ - 10.0).max() < 2.5                     # on the ring (sigma_rho = 0.5)
    spread = np.sqrt(-2 * np.log(np.hypot(np.cos(ang).mean(), np.sin(ang).mean())))   # circular std
    assert spread > 0.35, spread                               # a wide arc, not one ray (collapsed: < 0.01)
    # the measured bearing holds at every root of the ring
    b = np.arctan2(lm[1] - out[1], lm[0] - out[0]) - out[2]
    assert np.abs(np.arctan2(np.sin(b - 0.3), np.cos(b - 0.3))).max() < 0.2


def test_generic_measurement_beliefs_through_presampled_measurements():
    """`Pose2Point2BearingRange{B<:SamplableBelief, R<:SamplableBelief}` (src/factors/BearingRange2D.jl:10-27) is generic in its
    bearing and range beliefs; `getSample` just draws from them.  Any such belief is served by handing the library the N measurement
    samples themselves (opts.presampled = ROME_NOISE_MEASUREMENTS): here a Rayleigh range and a two-component mixture bearing.
    Closed form = the exact landmark of every sampled measurement; Newton and Nelder-Mead agree with it; Pose2Pose2 likewise with a
    heavy-tailed (Student-t) odometry sample."""
    rng = np.random.default_rng(8)
    N, C_ = 100, 3
    b = np.where(rng.random((C_, N)) < 0.5, rng.normal(-0.4, 0.02, (C_, N)), rng.normal(0.6, 0.05, (C_, N)))   # mixture bearing
    rho = rng.rayleigh(8.0, (C_, N)) + 2.0                                                                        # Rayleigh range
    z = np.stack([b, rho], axis=1)                                                                                # (C, 2, N)
    pose = rng.normal(0, 1, (C_, 3, N)); pose[:, 2] *= 0.3
    want = np.stack([pose[:, 0] + rho * np.cos(pose[:, 2] + b), pose[:, 1] + rho * np.sin(pose[:, 2] + b)], axis=1)
    for solver, tol in ((0, 1e-10), (1, 1e-9), (2, 2e-2)):
        o = R.make_opts(N=N, solver=solver, seed=3, presampled=1)
        got = R.conv_pose2point2br(o, 0, np.zeros((C_, 2)), np.ones((C_, 2)), pose, np.zeros((C_, 2, N)), noise=z)
        err = np.abs(got - want)
        assert np.median(err) < tol and (err.max() < tol or solver == 2), (solver, err.max())
    # Pose2Pose2 with Student-t odometry samples: dir 0 root = p ∘ exp(z)
    zt = np.stack([10 + 0.1 * rng.standard_t(3, (C_, N)), 0.1 * rng.standard_t(3, (C_, N)), np.pi / 3 + 0.02 * rng.standard_t(3, (C_, N))], axis=1)
    o = R.make_opts(N=N, solver=1, seed=3, presampled=1)
    got = R.conv_pose2pose2(o, np.zeros((C_, 3)), np.tile(np.eye(3), (C_, 1, 1)), pose, np.zeros((C_, 3, N)), dirs=[0] * C_, noise=zt)
    c, s = np.cos(pose[:, 2]), np.sin(pose[:, 2])
    want = np.stack([pose[:, 0] + c * zt[:, 0] - s * zt[:, 1], pose[:, 1] + s * zt[:, 0] + c * zt[:, 1], pose[:, 2] + zt[:, 2]], axis=1)
    d = got - want; d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    assert np.abs(d).max() < 1e-9
    with pytest.raises(Exception):
        R.conv_pose2pose2(R.make_opts(N=N, presampled=7), np.zeros((1, 3)), np.eye(3)[None], pose[:1], np.zeros((1, 3, N)), dirs=[0])
