"""BASELINE.json configs[2] and [4] shapes at size on the GPU (the bench line is configs[1]; these are
parity-test cases): device sweeps vs the oracle + size-independent properties."""
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


def _wd(a, b, ang):
    d = a - b
    for r in ang:
        d[:, r] = np.arctan2(np.sin(d[:, r]), np.cos(d[:, r]))
    return d


def test_mit_bearingrange_graph_sweeps_vs_oracle():
    import torch
    N = 100
    fg = R.synth_mit_br(P=808, n_landmarks=120, N=N)
    R.dead_reckon_init(fg, seed=4)
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    pk = dg.packed
    assert pk.p2p2["F"] == 807 and len(pk.labels[R.Point2]) == 120 and 240 <= pk.br["F"] <= 480
    o = R.make_opts(N=N, solver=1, seed=31)
    dg.conv_step(o, sweep=0)
    torch.cuda.synchronize()
    C2 = dg.tab["p2p2"]["C"]; Fb = pk.br["F"]
    prop2 = dg.prop[R.Pose2].cpu().numpy(); propl = dg.prop[R.Point2].cpu().numpy()
    bel2 = pk.beliefs(fg, R.Pose2); bell = pk.beliefs(fg, R.Point2)
    mk = lambda off: ro.make_opts(N=N, solver=1, seed=31, stream_offset=off)
    ref1 = ro.conv_pose2point2br(mk(dg.STREAM_BR1), 1, pk.br["mu"], pk.br["sigma"], bell, bel2, pk.br["point"], pk.br["pose"], want_status=True)
    ref0 = ro.conv_pose2point2br(mk(dg.STREAM_BR0), 0, pk.br["mu"], pk.br["sigma"], bel2, bell, pk.br["pose"], pk.br["point"])
    assert np.abs(propl[:Fb] - ref0).max() < 1e-8
    ok = ref1[1] == 0   # under-determined direction: compare where the oracle's min-norm Newton converged
    d1 = np.abs(_wd(prop2[C2:C2 + Fb], ref1[0], [2])).max(axis=1)
    assert ok.mean() > 0.99 and d1[ok].max() < 1e-7
    # property: landmark proposals satisfy the sampled bearing/range constraint => they lie on the range ring
    pose_of = bel2[pk.br["pose"]]
    rng_ = np.hypot(propl[:Fb, 0] - pose_of[:, 0], propl[:Fb, 1] - pose_of[:, 1])
    assert (np.abs(rng_ - pk.br["mu"][:, 1:2]) < 5 * pk.br["sigma"][:, 1:2] + 1e-9).mean() > 0.999
    # a few sweeps of the full loop keep landmarks near their generator truth
    dg.solve(o, n_sweeps=6)
    ml, _ = dg.belief_stats(R.Point2)
    truth = np.array([fg.ground_truth[l] for l in pk.labels[R.Point2]])
    err = np.hypot(*(ml.cpu().numpy() - truth).T)
    assert np.median(err) < 3.0


def test_helix3d_pose3pose3_sweep_vs_oracle_and_roundtrip():
    import torch
    N = 100
    P = 10000   # BASELINE.json configs[4]: 10k Pose3
    fg = R.synth_helix3d(P=P, N=N)
    R.dead_reckon_init_pose3(fg, seed=2)
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    tb = dg.tab["p3p3"]
    assert tb["F"] == (P - 1) + len(range(20, P, 5)) and tb["P"] == 1   # odometry + closures between adjacent turns
    o = R.make_opts(N=N, solver=1, seed=8)
    st = torch.zeros((tb["C"], N), dtype=torch.int32, device="cuda")
    prop = dg.sweep_pose3pose3(o, status=st)
    torch.cuda.synchronize()
    assert int(st.sum()) == 0                                   # every one of the ~2.4 M root-finds converged
    pk = dg.packed
    factor, dr, fixed, target = R.PackedGraph.conv_table(pk.p3p3)
    L = R.cholesky_lower(pk.p3p3["cov"])
    bel = pk.beliefs(fg, R.Pose3)
    n = 3000                                                    # oracle on a prefix of the table (same Philox streams)
    ref = ro.conv_pose3pose3(ro.make_opts(N=N, solver=1, seed=8), pk.p3p3["mu"], L, bel, fixed[:n], target[:n], dr[:n], factor=factor[:n])
    got = prop[:n].cpu().numpy()
    from scipy.spatial.transform import Rotation as Rot
    rv = lambda a: Rot.from_rotvec(a[:, 3:].transpose(0, 2, 1).reshape(-1, 3))
    # Manifolds' SO(3) log (restated on both sides) loses ~1/(π-θ) digits as the heading passes θ = π, which a helix
    # does once per turn: 1e-9 away from the cut, 1e-6 on it (north_star tolerance: 1e-3)
    ang = (rv(got).inv() * rv(ref)).magnitude(); theta = rv(ref).magnitude()
    assert np.abs(got[:, :3] - ref[:, :3]).max() < 1e-9 and ang[theta < 3.0].max() < 1e-9 and ang.max() < 1e-6
    # size-independent property on the WHOLE table: closed form == Newton
    prop0 = dg.sweep_pose3pose3(R.make_opts(N=N, solver=0, seed=8))
    a = prop.cpu().numpy(); b = prop0.cpu().numpy()
    ang = (rv(a).inv() * rv(b)).magnitude(); theta = rv(b).magnitude()
    assert np.abs(a[:, :3] - b[:, :3]).max() < 1e-9 and ang[theta < 3.0].max() < 1e-9 and ang.max() < 1e-6
    # ... and the functor ITERATION from the belief points (GAUSS_NEWTON: the packed sweep k_conv_flat<P3P3, 3>, residual on unit
    # quaternions) reaches the same roots on the whole table, every root-find converged
    st3 = torch.ones((tb["C"], N), dtype=torch.int32, device="cuda")
    prop3 = dg.sweep_pose3pose3(R.make_opts(N=N, solver=3, seed=8), status=st3)
    torch.cuda.synchronize()
    assert int(st3.sum()) == 0
    c = prop3.cpu().numpy()
    ang = (rv(c).inv() * rv(b)).magnitude()
    assert np.abs(c[:, :3] - b[:, :3]).max() < 1e-9 and ang[theta < 3.0].max() < 1e-9 and ang.max() < 1e-6
    # belief statistics of Pose3 beliefs vs oracle on a few variables
    mean, sd = dg.belief_stats(R.Pose3)
    for v in (0, 17, 4242, P - 1):
        m, s = ro.belief_spread(bel[v])
        assert np.abs(mean[v].cpu().numpy() - m).max() < 1e-9 and np.abs(sd[v].cpu().numpy() - s).max() < 1e-9


def test_manhattan_m3500_dataset_parametric_solve_and_sweep():
    """BASELINE configs[1] on the real data (tests/golden/manhattan.g2o = the reference's examples/manhattan.g2o): the parametric
    solve converges from dead reckoning to the well-known compact M3500 map, and a non-parametric sweep over all 10 906
    (factor, direction) convolutions + the prior runs clean around it."""
    import os
    import torch
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fg = R.loadG2o(os.path.join(gold, "manhattan.g2o"))
    R.dead_reckon_init(fg, seed=1)
    xp = R.solveGraphParametric(fg)
    X = np.array([xp["x%d" % k] for k in range(3500)])
    assert np.abs(X[:, :2]).max() < 70.0                      # the optimised M3500 map spans about [-50, 25] x [-60, 15] m
    # whitened residuals of the solution: chi2 per scalar residual well below 1 (3.6e3 over 16 359 residuals measured; the
    # reference's residual takes the translation difference in the WORLD frame, SURVEY A.2, so this is not g2o's 146)
    e = np.array([[int(l[0][1:]), int(l[1][1:])] for _, l, f in fg.factors if isinstance(f, R.Pose2Pose2)])
    mu = np.array([f.Z.mu for _, _, f in fg.factors if isinstance(f, R.Pose2Pose2)])
    Wi = np.array([np.linalg.inv(f.Z.cov) for _, _, f in fg.factors if isinstance(f, R.Pose2Pose2)])
    p, q = X[e[:, 0]], X[e[:, 1]]
    c, s = np.cos(p[:, 2]), np.sin(p[:, 2])
    r = np.stack([p[:, 0] + c * mu[:, 0] - s * mu[:, 1] - q[:, 0], p[:, 1] + s * mu[:, 0] + c * mu[:, 1] - q[:, 1],
                  np.arctan2(np.sin(p[:, 2] + mu[:, 2] - q[:, 2]), np.cos(p[:, 2] + mu[:, 2] - q[:, 2]))], 1)
    chi2 = np.einsum("fi,fij,fj->", r, Wi, r)
    assert chi2 < 0.4 * 3 * len(e), chi2
    dg = R.DeviceGraph(fg); dg.init_from_means(xp)
    tb = dg.tab["p2p2"]
    status = torch.zeros((tb["C"], 100), dtype=torch.int32, device="cuda")
    out = dg.sweep_pose2pose2(R.make_opts(N=100, solver=1, seed=3500), status=status)
    torch.cuda.synchronize()
    assert tb["C"] == 10907 and int(status.sum()) == 0 and bool(torch.isfinite(out).all())
    # every proposal sits at its target's parametric estimate within the factor + belief noise
    m, sd = R.belief_stats(out.cpu().numpy()[:2000])
    tgt = X[tb["target"].cpu().numpy()[:2000]]
    d = m - tgt; d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    assert np.percentile(np.hypot(d[:, 0], d[:, 1]), 99) < 1.0 and np.percentile(np.abs(d[:, 2]), 99) < 0.3


def test_mit_dataset_with_landmarks_sweeps_and_solve():
    """BASELINE configs[2] on the real pose graph (tests/golden/MIT.g2o = the reference's examples/MIT.g2o, 808 poses / 827
    EDGE_SE2, no landmarks in the file): landmarks are synthesised around the parametric solution and sighted with
    Pose2Point2BearingRange; both bearing-range directions run over the graph and a few solve iterations keep the landmarks
    at their true places."""
    import os
    import torch
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    N = 100
    fg = R.loadG2o(os.path.join(gold, "MIT.g2o"), N=N)
    R.dead_reckon_init(fg, seed=2)
    xp = R.solveGraphParametric(fg)
    truth = R.add_synthetic_landmarks(fg, xp, 150, 808)
    R.dead_reckon_init(fg, seed=2)
    dg = R.DeviceGraph(fg)
    dg.upload_beliefs(fg)
    dg.init_from_means({**xp, **truth})
    pk = dg.packed
    assert pk.p2p2["F"] == 827 and len(pk.labels[R.Pose2]) == 808 and len(pk.labels[R.Point2]) == 150 and pk.br["F"] >= 300
    o = R.make_opts(N=N, solver=1, seed=808)
    st1 = torch.zeros((pk.br["F"], N), dtype=torch.int32, device="cuda")
    p1 = dg.sweep_bearingrange(o, 1, status=st1); p0 = dg.sweep_bearingrange(o, 0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(p1).all()) and bool(torch.isfinite(p0).all()) and float((st1 != 0).float().mean()) < 0.01
    dg.solve(o, n_sweeps=6)
    ml, _ = dg.belief_stats(R.Point2)
    err = np.hypot(*(ml.cpu().numpy() - np.array([truth[l] for l in pk.labels[R.Point2]])).T)
    assert np.median(err) < 1.0 and np.percentile(err, 95) < 4.0, (np.median(err), np.percentile(err, 95))
    m2, _ = dg.belief_stats(R.Pose2)
    X = np.array([xp[l] for l in pk.labels[R.Pose2]])
    assert np.median(np.hypot(*(m2.cpu().numpy()[:, :2] - X[:, :2]).T)) < 1.0


def test_helix3d_parametric_gauss_newton():
    """BASELINE configs[4]: synthetic SE(3) helix, Pose3Pose3 odometry + closures between adjacent turns, solved with the batched
    residual/Jacobian kernel (`rome_linearize`) + sparse Levenberg-Marquardt: from a dead-reckoned start the trajectory returns
    to the generator's ground truth at the measurement-noise floor."""
    from scipy.spatial.transform import Rotation as Rot
    P = 1500
    fg = R.synth_helix3d(P=P)
    R.dead_reckon_init_pose3(fg, seed=7)
    gt = np.array([fg.ground_truth["x%d" % k] for k in range(P)])
    init = np.array([fg.getVal("x%d" % k).mean(axis=1) for k in range(P)])
    xp = R.solveGraphParametric(fg)
    X = np.array([xp["x%d" % k] for k in range(P)])
    e_init = np.sqrt(((init[:, :3] - gt[:, :3]) ** 2).sum(1).mean())
    e = np.sqrt(((X[:, :3] - gt[:, :3]) ** 2).sum(1).mean())
    ang = (Rot.from_rotvec(X[:, 3:]).inv() * Rot.from_rotvec(gt[:, 3:])).magnitude()
    # measured: 9.4 m dead-reckoned -> 0.9 m (the remaining error is the drift along the helix axis that odometry noise leaves
    # unobservable: closures only tie adjacent turns together)
    assert e < 1.5 and e < 0.2 * e_init and np.median(ang) < 0.08, (e_init, e, np.median(ang))


@pytest.mark.timeout(900)
def test_helix3d_parametric_gauss_newton_at_10k_poses():
    """BASELINE configs[4] AS WRITTEN: 10 000 Pose3 / ~24 000 Pose3Pose3 factors, parametric Gauss-Newton on the batched
    residual / Jacobian kernel.  Every LM iteration is one rome_linearize launch over all factors (60 000 unknowns) + the host
    sparse solve; the solution returns to the generator's ground truth at the noise floor, like the 1500-pose case above."""
    import time
    from scipy.spatial.transform import Rotation as Rot
    P = 10000
    fg = R.synth_helix3d(P=P)
    R.dead_reckon_init_pose3(fg, seed=7)
    gt = np.array([fg.ground_truth["x%d" % k] for k in range(P)])
    init = np.array([fg.getVal("x%d" % k).mean(axis=1) for k in range(P)])
    t = time.perf_counter()
    xp = R.solveGraphParametric(fg, max_iters=40)
    dt = time.perf_counter() - t
    X = np.array([xp["x%d" % k] for k in range(P)])
    e_init = np.sqrt(((init[:, :3] - gt[:, :3]) ** 2).sum(1).mean())
    e = np.sqrt(((X[:, :3] - gt[:, :3]) ** 2).sum(1).mean())
    ang = (Rot.from_rotvec(X[:, 3:]).inv() * Rot.from_rotvec(gt[:, 3:])).magnitude()
    print("helix 10k parametric: %.1f s, rms %.2f m (dead-reckoned %.1f m), median angle %.3f rad" % (dt, e, e_init, np.median(ang)))
    # 500 turns of helix: what odometry + adjacent-turn closures leave unobservable (drift along the axis, a slow twist) grows
    # with the length of the chain, so the distance to the generator's truth is larger than for 1500 poses (measured: 215 m
    # dead-reckoned -> 26 m, median angle 0.12 rad, 5.6 s of which the kernel part is ~40 ms); the solve must still remove
    # most of the dead-reckoning error
    assert e < 0.2 * e_init and np.median(ang) < 0.25, (e_init, e, np.median(ang))


def test_fixed_lag_example_end_to_end(tmp_path):
    """examples/manhattan_fixedlag.py (counterpart of examples/ManhattanDatasetFixedLag.jl): incremental parse -> approxConv init ->
    fifoFreeze -> window solve -> calcPPE / kde_bandwidth -> saveDFG + exportG2o; the archive reloads to the same graph and the
    estimates stay on the dead-reckoned Manhattan grid."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("manhattan_fixedlag", os.path.join(root, "examples", "manhattan_fixedlag.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    fg, arch, g2o = mod.run(n_instructions=60, qfl=12, stride=15, out_prefix=str(tmp_path / "fl"), verbose=False)
    assert len(fg.ls()) >= 50 and R.isMarginalized(fg, "x0") and not R.isMarginalized(fg, fg.ls()[-1])
    g = R.loadDFG(arch)
    assert g.ls() == fg.ls() and len(g.factors) == len(fg.factors)
    assert all(np.array_equal(g.getVal(l), fg.getVal(l)) for l in fg.ls())
    assert np.allclose(g.ppes["x5"]["default"]["suggested"], fg.ppes["x5"]["default"]["suggested"]) and (g.bws["x5"] > 0).all()
    lines = open(g2o).read().splitlines()
    assert sum(l.startswith("VERTEX_SE2") for l in lines) == len(fg.ls()) and sum(l.startswith("EDGE_SE2") for l in lines) == 60
    # unit-step Manhattan world: consecutive pose estimates are ~1 m apart
    P = np.array([fg.ppes["x%d" % k]["default"]["suggested"][:2] for k in range(len(fg.ls()))])
    d = np.linalg.norm(np.diff(P, axis=0), axis=1)
    assert 0.7 < np.median(d) < 1.3 and d.max() < 2.5
