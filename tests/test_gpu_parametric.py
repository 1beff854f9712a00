"""Parametric path on the GPU: analytic batched Jacobians vs finite differences of the oracle's
residual functors, and the reference's deterministic parametric-solve pins (SURVEY §8(c) Param-* rows)."""
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


def _retract(kind_dim, x, d):
    from scipy.spatial.transform import Rotation as Rot
    if kind_dim == 6:
        return np.concatenate([x[:3] + d[:3], (Rot.from_rotvec(x[3:]) * Rot.from_rotvec(d[3:])).as_rotvec()])
    return x + d


def _fd(fun, xa, xb, dims, h=1e-6):
    """central differences of r(xa ⊕ δa, xb ⊕ δb) through the ORACLE residuals"""
    da, db = dims
    r0 = fun(xa, xb)
    Ja = np.zeros((r0.size, da)); Jb = np.zeros((r0.size, db)) if db else None
    for j in range(da):
        e = np.zeros(da); e[j] = h
        Ja[:, j] = (fun(_retract(da, xa, e), xb) - fun(_retract(da, xa, -e), xb)) / (2 * h)
    for j in range(db or 0):
        e = np.zeros(db); e[j] = h
        Jb[:, j] = (fun(xa, _retract(db, xb, e)) - fun(xa, _retract(db, xb, -e))) / (2 * h)
    return r0, Ja, Jb


def test_jacobians_vs_oracle_finite_differences():
    rng = np.random.default_rng(8)
    F = 40
    L = R._lib
    cases = [
        (L.FACTOR_POSE2POSE2, (3, 3), lambda z: (lambda a, b: ro.residual_pose2pose2([z], [a], [b])[0]), [3, 3, 1.5], [8, 8, 3], [8, 8, 3]),
        (L.FACTOR_PRIORPOSE2, (3, 0), lambda z: (lambda a, b: ro.residual_priorpose2([z], [a])[0]), [3, 3, 1.5], [3, 3, 1.5], None),
        (L.FACTOR_POSE2POINT2BR, (3, 2), lambda z: (lambda a, b: ro.residual_pose2point2br([z], [a], [b])[0]), None, [8, 8, 3], [9, 9]),
        (L.FACTOR_PRIORPOINT2, (2, 0), lambda z: (lambda a, b: z - a), [5, 5], [5, 5], None),
        (L.FACTOR_POSE3POSE3, (6, 6), lambda z: (lambda a, b: ro.residual_pose3pose3([z], [a], [b])[0]), [2, 2, 2, .5, .5, .5], [5, 5, 5, .7, .7, .7], [5, 5, 5, .7, .7, .7]),
        (L.FACTOR_PRIORPOSE3, (6, 0), lambda z: (lambda a, b: ro.residual_priorpose3([z], [a])[0]), [2, 2, 2, .5, .5, .5], [2, 2, 2, .5, .5, .5], None),
    ]
    for kind, (da, db), mk, zs, sa, sb in cases:
        dz = len(zs) if zs is not None else 2
        mu = rng.standard_normal((F, dz)) * (zs if zs is not None else 1.0)
        if zs is None:
            mu = np.stack([rng.uniform(-3, 3, F), rng.uniform(3, 20, F)], 1)
        xa = rng.standard_normal((F, da)) * sa
        xb = rng.standard_normal((F, db)) * sb if db else None
        dr = R.api._LIN_DIMS[kind][1]
        A = rng.standard_normal((F, dr, dr)) * 0.3 + np.eye(dr)
        r, Ja, Jb = R.linearize(kind, mu, A, xa, xb)
        for f in range(F):
            r0, ja, jb = _fd(mk(mu[f]), xa[f], None if xb is None else xb[f], (da, db))
            if kind == L.FACTOR_POSE2POINT2BR and abs(abs(r0[0]) - np.pi) < 1e-3:
                continue  # finite differences across the ±π cut of the bearing are meaningless
            assert np.allclose(r[f], A[f] @ r0, atol=1e-9), (kind, f)
            assert np.allclose(Ja[f], A[f] @ ja, atol=2e-6), (kind, f, np.abs(Ja[f] - A[f] @ ja).max())
            if db:
                assert np.allclose(Jb[f], A[f] @ jb, atol=2e-6), (kind, f)


def _pose2_close(x, ref, atol):
    d = np.asarray(x) - np.asarray(ref)
    d[2] = np.arctan2(np.sin(d[2]), np.cos(d[2]))
    return np.abs(d).max() < atol


def test_param_pose2_square_pin():
    # test/testParametric.jl:16-57 : prior (10,10,-π+1e-5), 4x Pose2Pose2 (10,0,π/2) ; MvNormal(μ, σ-vector)
    fg = R.initfg()
    fg.addVariable("x0", R.Pose2)
    fg.addFactor(["x0"], R.PriorPose2(R.MvNormal([10, 10, -np.pi + 1e-5], np.square([0.1, 0.1, 0.05]))))
    for i in range(4):
        fg.addVariable("x%d" % (i + 1), R.Pose2)
        fg.addFactor(["x%d" % i, "x%d" % (i + 1)], R.Pose2Pose2(R.MvNormal([10.0, 0, np.pi / 2], np.square([0.1, 0.1, 0.1]))))
    x = R.solveGraphParametric(fg)
    ref = [[10, 10, -np.pi], [0, 10, -np.pi / 2], [0, 0, 0], [10, 0, np.pi / 2], [10, 10, -np.pi]]
    for i in range(5):
        assert _pose2_close(x["x%d" % i], ref[i], 1e-3), (i, x["x%d" % i])


def test_param_bearingrange_pin():
    # test/testParametric.jl:155-181
    fg = R.initfg()
    fg.addVariable("x1", R.Pose2); fg.addVariable("l1", R.Point2); fg.addVariable("l2", R.Point2)
    fg.addFactor(["l1"], R.PriorPoint2(R.MvNormal([1.0, 1], np.square([0.01, 0.01]))))
    fg.addFactor(["l2"], R.PriorPoint2(R.MvNormal([1.0, -1], np.square([0.01, 0.01]))))
    fg.addFactor(["x1", "l1"], R.Pose2Point2BearingRange(R.Normal(np.pi / 4, 0.01), R.Normal(np.sqrt(2), 0.1)))
    fg.addFactor(["x1", "l2"], R.Pose2Point2BearingRange(R.Normal(3 * np.pi / 4, 0.01), R.Normal(np.sqrt(2), 0.1)))
    x = R.solveGraphParametric(fg)
    assert _pose2_close(x["x1"], [2, 0, np.pi / 2], 1e-3), x["x1"]
    assert np.allclose(x["l1"], [1, 1], atol=1e-3) and np.allclose(x["l2"], [1, -1], atol=1e-3)


def test_param_covariance_weighting_pin():
    # test/testParametricCovariances.jl:33-55 : two Pose2Pose2 with different Σ -> x1 = (1.05, 0, 0)
    fg = R.initfg()
    fg.addVariable("x0", R.Pose2); fg.addVariable("x1", R.Pose2)
    fg.addFactor(["x0"], R.PriorPose2(R.MvNormal([0.0, 0, 0], np.diag(np.square([0.1, 0.1, 0.01])))))
    fg.addFactor(["x0", "x1"], R.Pose2Pose2(R.MvNormal([1.1, 0, 0], np.diag(np.square([0.1, 0.1, 0.01])))))
    fg.addFactor(["x0", "x1"], R.Pose2Pose2(R.MvNormal([0.9, 0, 0], np.diag(np.square([np.sqrt(0.03), 0.1, 0.01])))))
    x = R.solveGraphParametric(fg)
    assert np.allclose(x["x0"], [0, 0, 0], atol=1e-4) and np.allclose(x["x1"], [1.05, 0, 0], atol=1e-4)


def test_param_pose3_loop_pin():
    # test/testPose3.jl:27-70 : prior pitch -π/4, 4x Pose3Pose3 (√2,0,0,0,0,π/2) closes the loop: p0 ≈ p4
    from scipy.spatial.transform import Rotation as Rot
    fg = R.initfg()
    fg.addVariable("x0", R.Pose3)
    fg.addFactor(["x0"], R.PriorPose3(R.MvNormal([0.0, 0, 0, 0, -np.pi / 4, 0], np.diag(np.square([0.1, 0.1, 0.1, 0.01, 0.01, 0.01])) ** 1)))
    odo = R.MvNormal([np.sqrt(2), 0, 0, 0, 0, np.pi / 2], np.diag(np.square([0.1, 0.1, 0.1, 0.01, 0.01, 0.01])))
    for i in range(4):
        fg.addVariable("x%d" % (i + 1), R.Pose3)
        fg.addFactor(["x%d" % i, "x%d" % (i + 1)], R.Pose3Pose3(odo))
    x = R.solveGraphParametric(fg)
    p0, p4 = x["x0"], x["x4"]
    assert np.abs(p0[:3] - p4[:3]).max() < 1e-3
    assert (Rot.from_rotvec(p0[3:]).inv() * Rot.from_rotvec(p4[3:])).magnitude() < 1e-3
    assert np.allclose(p0, [0, 0, 0, 0, -np.pi / 4, 0], atol=1e-6)


def test_param_manhattan_recovers_ground_truth():
    """g2o-shaped graph (config 2 shape, smaller): the parametric solve lands on the generator's ground
    truth within the measurement noise, and beats the dead-reckoned initial guess."""
    fg = R.synth_manhattan(P=400, loops=200, seed=3)
    x0 = R.initParametric(fg, refine=False)          # dead reckoning along the odometry
    x1 = R.initParametric(fg)                        # + the two linear solves (headings, then translations)
    x = R.solveGraphParametric(fg, init=x1)
    gt = fg.ground_truth

    def err(sol):  # RMS position error after removing the gauge (best rigid alignment, 2-D Kabsch)
        A = np.array([sol[l][:2] for l in gt]); B = np.array([gt[l][:2] for l in gt])
        ca, cb = A.mean(0), B.mean(0)
        U, _, Vt = np.linalg.svd((A - ca).T @ (B - cb))
        Rm = (U @ np.diag([1, np.sign(np.linalg.det(U @ Vt))]) @ Vt).T
        return np.sqrt(np.mean(np.sum(((A - ca) @ Rm.T + cb - B) ** 2, axis=1)))
    assert err(x) < 0.5 * err(x0) and err(x) < 0.3 and err(x1) < 1.2 * err(x) + 0.05, (err(x), err(x1), err(x0))
    xd = R.solveGraphParametric(fg, init=x0)         # the same optimum from the dead-reckoned start
    assert abs(err(xd) - err(x)) < 0.02
