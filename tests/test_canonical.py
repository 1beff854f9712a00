"""Canonical generators and the g2o exporter against the reference's pins (SURVEY.md §8(c) rows "g2o" and "Generators"):
test/testG2oParser.jl:27-50, test/testGenerateHelix.jl:18-26,59-64,85-116, test/testBeehiveGrow.jl:44-48."""
import os

import numpy as np
import pytest

import rome_jl_amd as R


def test_export_g2o_hexagonal_matches_reference_lines(tmp_path):
    # test/testG2oParser.jl:27-34: the eight literal lines exportG2o(generateGraph_Hexagonal()) must produce
    reflines = ["EDGE_SE2 0 1 10.0 0.0 1.0471975511965976 100.0 0.0 -0.0 100.0 -0.0 100.0",
                "LANDMARK 0 2 0.0 20.0 99.99999999999999 0.0 1.0",
                "EDGE_SE2 1 3 10.0 0.0 1.0471975511965976 100.0 0.0 -0.0 100.0 -0.0 100.0",
                "EDGE_SE2 3 4 10.0 0.0 1.0471975511965976 100.0 0.0 -0.0 100.0 -0.0 100.0",
                "EDGE_SE2 4 5 10.0 0.0 1.0471975511965976 100.0 0.0 -0.0 100.0 -0.0 100.0",
                "EDGE_SE2 5 6 10.0 0.0 1.0471975511965976 100.0 0.0 -0.0 100.0 -0.0 100.0",
                "EDGE_SE2 6 7 10.0 0.0 1.0471975511965976 100.0 0.0 -0.0 100.0 -0.0 100.0",
                "LANDMARK 7 2 0.0 20.0 99.99999999999999 0.0 1.0"]
    fg = R.generateGraph_Hexagonal()
    path = R.exportG2o(fg, filename=str(tmp_path / "hex.g2o"))
    with open(path) as fh:
        lines = fh.read().splitlines()
    assert lines == reflines


def test_export_then_import_round_trip(tmp_path):
    fg = R.synth_manhattan(P=300, loops=60)
    path = R.exportG2o(fg, filename=str(tmp_path / "m.g2o"))
    back = R.loadG2o(path)
    # the exporter renumbers variables in order of first appearance and walks the factors pose by pose: same factor set
    a = sorted((tuple(f.Z.mu), tuple(f.Z.cov.ravel())) for _, _, f in fg.factors if isinstance(f, R.Pose2Pose2))
    b = sorted((tuple(f.Z.mu), tuple(f.Z.cov.ravel())) for _, _, f in back.factors if isinstance(f, R.Pose2Pose2))
    assert len(a) == len(b) == 359
    for (ma, ca), (mb, cb) in zip(a, b):
        assert np.allclose(ma, mb, rtol=0, atol=1e-15)
        assert np.allclose(ca, cb, rtol=1e-9, atol=1e-18)
    assert len(back.ls()) == len(fg.ls())


def test_julia_float_strings():
    from rome_jl_amd.canonical import _jl
    assert [_jl(x) for x in (10.0, -0.0, 1.0471975511965976, 99.99999999999999, 1e-5, 1.5e-7, 123456.0, 1e6, 1.25e22, 0.0001)] == \
        ["10.0", "-0.0", "1.0471975511965976", "99.99999999999999", "1.0e-5", "1.5e-7", "123456.0", "1.0e6", "1.25e22", "0.0001"]


def test_invcov_matches_numpy():
    from rome_jl_amd.canonical import _invcov
    rng = np.random.default_rng(3)
    for n in (2, 3, 6):
        A = rng.standard_normal((n, n)); S = A @ A.T + n * np.eye(n)
        assert np.allclose(_invcov(S), np.linalg.inv(S), rtol=1e-12, atol=1e-14)


def test_boxes2d_corners():
    # test/testGenerateHelix.jl:15-26
    fg = R.generateGraph_Boxes2D(8)
    want = [[0, 0], [15, 0], [15, 15], [5, 15], [5, 0], [20, 0], [20, 15], [10, 15], [10, 0]]
    for k, w in enumerate(want):
        assert np.allclose(R.getPPE(fg, "x%d" % k), w, atol=1e-3)
    assert len(fg.ls()) == 9 and all(fg.variables[l] is R.Point2 for l in fg.ls())


def test_helix2d_slew_last_pose():
    # test/testGenerateHelix.jl:59-64
    seen = []
    fg = R.generateGraph_Helix2DSlew(46, slew_x=2 / 3, posesperturn=15, radius=10, Qd=np.diag(np.square([0.1, 0.1, 0.05])),
                                     postpose_cb=lambda g, l: seen.append(l))
    last = sorted((l for l in fg.ls()), key=lambda s: int(s[1:]))[-1]
    assert last == "x45" and seen[0] == "x0" and seen[-1] == "x45"
    assert np.allclose(R.getPPE(fg, last), [20, 0, 1.465088], atol=0.001)


def test_helix2d_spiral_runs():
    # test/testGenerateHelix.jl:69
    fg = R.generateGraph_Helix2DSpiral(200, rate_r=0.6, rate_a=6, radius=100)
    assert len(fg.ls()) == 200
    assert all(np.isfinite(R.getPPE(fg, l)).all() for l in fg.ls())
    assert all(np.isfinite(f.Z.mu).all() for _, _, f in fg.factors)


def test_helix2d_pins_and_extension():
    # test/testGenerateHelix.jl:79-116
    fg = R.generateGraph_Helix2D(5, posesperturn=15, radius=10)
    ppes = [[0.0, 0.0, 1.5707963267948966],
            [0.8645454235739924, 4.067366430758004, 1.151917276019672],
            [3.3086939364114176, 7.431448254773942, 0.7330382545911657],
            [6.909830056250526, 9.510565162951536, 0.31415923447063226],
            [11.045284632676536, 9.945218953682733, -0.10471978645923721]]
    labels = fg.ls()
    assert labels == ["x0", "x1", "x2", "x3", "x4"]
    for l, w in zip(labels, ppes):
        assert np.allclose(R.getPPE(fg, l), w, atol=1e-5)
    R.generateGraph_Helix2D(5, fg=fg, posesperturn=15, radius=10)
    assert len(fg.ls()) == 5 and not fg.exists("x5")
    R.generateGraph_Helix2D(6, fg=fg, posesperturn=15, radius=10)
    assert len(fg.ls()) == 6 and fg.exists("x5")
    assert np.allclose(R.getPPE(fg, "x5"), [15.0, 8.660254037844387, -0.5235988055902416], atol=1e-5)
    # the odometry factors are the exact relative poses of consecutive simulated poses
    from rome_jl_amd.graph import se2_compose
    for (_, ls_, f) in fg.factors:
        if isinstance(f, R.Pose2Pose2):
            q = se2_compose(R.getPPE(fg, ls_[0]), f.Z.mu)
            assert np.allclose(q[:2], R.getPPE(fg, ls_[1])[:2], atol=1e-9)
            assert abs(np.angle(np.exp(1j * (q[2] - R.getPPE(fg, ls_[1])[2])))) < 1e-9


def test_honeycomb_growth_and_landmark_association():
    # test/testBeehiveGrow.jl:20-48: grown 7 -> 14 -> 21 poses; centres of the windows the reference asserts after solving
    fg = R.generateGraph_Honeycomb(7)
    assert fg.exists("x7") and not fg.exists("x8")
    R.generateGraph_Honeycomb(14, fg=fg)
    R.generateGraph_Honeycomb(21, fg=fg)
    poses = [l for l in fg.ls() if l[0] == "x"]
    assert len(poses) == 22
    s3 = np.sin(np.pi / 3)
    assert np.allclose(R.getPPE(fg, "l11"), [5, 10 * s3], atol=1e-8)
    assert np.allclose(R.getPPE(fg, "l0"), [20, 0], atol=1e-8)
    assert np.allclose(R.getPPE(fg, "l7"), [20, -20 * s3], atol=1e-8)
    assert np.allclose(R.getPPE(fg, "x21")[:2], [10, -20 * s3], atol=1e-8)
    # landmark re-sightings of the reference's recipe table (GenerateHoneycomb.jl:4-12): x6 and x18 see l0 again, x10 sees l1
    sight = {ls_[0]: ls_[1] for _, ls_, f in fg.factors if isinstance(f, R.Pose2Point2BearingRange)}
    assert sight["x6"] == "l0" and sight["x18"] == "l0" and sight["x10"] == "l1" and sight["x12"] == "l5" and sight["x19"] == "l5"
    assert not fg.exists("l6") and not fg.exists("l18") and not fg.exists("l10")
    # growing in one go gives the same graph
    fg2 = R.generateGraph_Honeycomb(21)
    assert fg2.ls() == fg.ls() and [f[0] for f in fg2.factors] == [f[0] for f in fg.factors]


def test_honeycomb_full_recipe_is_reproduced_geometrically():
    # every landmark merge of the 84-pose comb comes out of the 1 m association rule; spot-check the late entries of the
    # reference table (GenerateHoneycomb.jl:40-50): l46 -> l37, l53 -> l44, l60 -> l51, l38 -> l29
    fg = R.generateGraph_Honeycomb(84)
    sight = {ls_[0]: ls_[1] for _, ls_, f in fg.factors if isinstance(f, R.Pose2Point2BearingRange)}
    assert sight["x46"] == "l37" and sight["x53"] == "l44" and sight["x60"] == "l51" and sight["x38"] == "l29"
    assert sight["x70"] == "l0" and sight["x69"] == "l1" and sight["x62"] == "l11"


def test_beehive_generator():
    # test/testBeehiveGrow.jl:64-66
    fg = R.generateGraph_Beehive(8, seed=4)
    assert np.allclose(R.getPPE(fg, "x0")[:2], [0.0, 0.0], atol=1e-8)
    assert fg.exists("x8") and not fg.exists("x9")
    # every pose sits on a honeycomb lattice vertex 10 m from its predecessor
    for k in range(8):
        a, b = R.getPPE(fg, "x%d" % k), R.getPPE(fg, "x%d" % (k + 1))
        assert abs(np.hypot(*(b[:2] - a[:2])) - 10.0) < 1e-9


def test_two_pose_odo_and_chain():
    fg = R.generateGraph_TwoPoseOdo()
    assert fg.ls() == ["x0", "x1", "l1"]
    assert np.allclose(R.getPPE(fg, "x1"), [10, 0, 0]) and np.allclose(R.getPPE(fg, "l1"), [30, 0])
    fg = R.buildGraphChain()
    assert fg.ls() == ["x0", "x1", "x2", "x3"] and np.allclose(R.getPPE(fg, "x3"), [30, 0, 0])


def test_point2point2_is_outside_the_device_path():
    fg = R.generateGraph_Boxes2D(4)
    with pytest.raises(TypeError):
        R.PackedGraph(fg)


def test_g2o_se3_export_import_round_trip(tmp_path):
    # test/testG2oExportSE3.jl:7-20 (export of an SE(3) graph with vertex estimates) + the EDGE_SE3:QUAT / VERTEX_SE3:QUAT
    # import path of src/services/g2oParser.jl:76-168
    from scipy.spatial.transform import Rotation as Rot
    fg = R.synth_helix3d(P=40)
    labels = [l for l in fg.ls()]
    est = {l: fg.ground_truth[l] for l in labels}
    path = R.exportG2o(fg, filename=str(tmp_path / "h.g2o"), estimates=est, varIntLabel={l: i for i, l in enumerate(labels)})
    ins = R.importG2o(path)
    assert sum(1 for i in ins if i[0] == "VERTEX_SE3:QUAT") == 40
    n_edges = sum(1 for _, _, f in fg.factors if isinstance(f, R.Pose3Pose3))
    assert sum(1 for i in ins if i[0] == "EDGE_SE3:QUAT") == n_edges and all(len(i) == 31 for i in ins if i[0] == "EDGE_SE3:QUAT")
    back = R.initfg()
    for i in ins:
        R.parseG2oInstruction(back, i)
    assert len(back.ls()) == 40
    a = [(l, f) for _, l, f in fg.factors if isinstance(f, R.Pose3Pose3)]
    b = [(l, f) for _, l, f in back.factors if isinstance(f, R.Pose3Pose3)]
    assert sorted(tuple(l) for l, _ in a) == sorted(tuple(l) for l, _ in b)   # written pose by pose, same factor set
    bd = {tuple(l): f for l, f in b}
    for la, fa in a:
        fb = bd[tuple(la)]
        assert np.allclose(fa.Z.mu[:3], fb.Z.mu[:3], atol=1e-14)
        dR = Rot.from_rotvec(fa.Z.mu[3:]).as_matrix().T @ Rot.from_rotvec(fb.Z.mu[3:]).as_matrix()
        assert np.linalg.norm(Rot.from_matrix(dR).as_rotvec()) < 1e-12
        assert np.allclose(fa.Z.cov, fb.Z.cov, rtol=1e-9, atol=1e-16)
    for l in labels:
        assert np.allclose(back.vertex_init[l][:3], est[l][:3], atol=1e-14)
        dR = Rot.from_rotvec(back.vertex_init[l][3:]).as_matrix().T @ Rot.from_rotvec(est[l][3:]).as_matrix()
        assert np.linalg.norm(Rot.from_matrix(dR).as_rotvec()) < 1e-12


def test_accumulate_factor_means():
    # test/testAccumulateFactors.jl:13-33
    fg = R.initfg()
    fg.addVariable("x0", R.Pose2); fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), 0.001 * np.eye(3))))
    fg.addVariable("x1", R.Pose2); fg.addFactor(["x0", "x1"], R.Pose2Pose2(R.MvNormal([10, 0, 0.0], 0.001 * np.eye(3))))
    assert np.allclose(R.accumulateFactorMeans(fg, ["x0f1", "x0x1f1"]), [10, 0, 0], atol=1e-3)
    # test/testParametricSimulated.jl:20-65: heading wraps to ≈ ±π
    fg = R.initfg()
    fg.addVariable("x0", R.Pose2); fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), 0.01 * np.eye(3))))
    fg.addVariable("x1", R.Pose2); fg.addFactor(["x0", "x1"], R.Pose2Pose2(R.MvNormal([0, 0, -np.pi + 0.01], 0.03 * np.eye(3))))
    v = R.accumulateFactorMeans(fg, ["x0f1", "x0x1f1"])
    assert np.allclose(v[:2], [0, 0], atol=5e-4) and 0.9 * np.pi < abs(v[2])
    # test/testParametricSimulated.jl:78-152
    fg = R.initfg()
    fg.addVariable("x2", R.Pose2)
    fg.addFactor(["x2"], R.PriorPose2(R.MvNormal([15.000000000016204, 8.660254037814505, 2.0943951023931953], 0.01 * np.eye(3))))
    fg.addVariable("x3", R.Pose2); fg.addFactor(["x2", "x3"], R.Pose2Pose2(R.MvNormal([10, 0, np.pi / 3], 0.01 * np.eye(3))))
    v = R.accumulateFactorMeans(fg, ["x2f1", "x2x3f1"])
    assert np.allclose(v[:2], [10, 17.32], atol=1e-2) and abs(abs(v[2]) - np.pi) < 1e-2


# ---------------------------------------------------------------------------------------------- saveDFG / loadDFG
def test_loaddfg_reads_the_reference_test_graph_and_exports_it():
    """test/testG2oExportSE3.jl:21-31: loadDFG! of test/testdata/g2otest.tar.gz (byte copy in tests/golden), then exportG2o with
    varIntLabel; the graph is the one the commented recipe of that test builds (:9-15)."""
    import os
    import tempfile
    fg = R.loadDFG(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g2otest.tar.gz"))
    assert fg.ls() == ["x0", "x1", "x2", "x3"] and all(fg.variables[l] is R.Pose3 for l in fg.ls())
    assert fg.N == 100 and fg.solverParams["inflateCycles"] == 3 and fg.solverParams["inflation"] == 5.0
    kinds = sorted((type(f).__name__, tuple(ls)) for _, ls, f in fg.factors)
    assert kinds == [("Pose3Pose3", ("x0", "x1")), ("Pose3Pose3", ("x1", "x2")), ("Pose3Pose3", ("x2", "x3")), ("PriorPose3", ("x0",))]
    for _, ls, f in fg.factors:
        assert np.allclose(f.Z.cov, 0.1 * np.eye(6))
        assert np.allclose(f.Z.mu, [1, 0, 0, 0, 0, 0] if len(ls) == 2 else np.zeros(6))
    assert not any(fg.isInitialized(l) for l in fg.ls())          # saved with graphinit=false
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "test.g2o")
        R.exportG2o(fg, filename=out, varIntLabel={l: i for i, l in enumerate(fg.ls())},
                    estimates={l: np.zeros(6) for l in fg.ls()})
        lines = open(out).read().splitlines()
    assert [l.split()[0] for l in lines] == ["VERTEX_SE3:QUAT"] * 4 + ["EDGE_SE3:QUAT"] * 3
    e = lines[4].split()
    assert e[1:3] == ["0", "1"] and [float(x) for x in e[3:10]] == [1, 0, 0, 0, 0, 0, 1] and float(e[10]) == pytest.approx(10.0)


def test_savedfg_loaddfg_round_trip_all_factor_types():
    import os
    import tempfile
    fg = R.generateGraph_Hexagonal(N=50)
    fg.addVariable("l2", R.Point2)
    fg.addFactor(["x2", "l1", "l2"], R.Pose2Point2BearingRange(R.Normal(0.1, 0.05), R.Normal(12.0, 0.4)), multihypo=[1.0, 0.3, 0.7])
    fg.addFactor(["l2"], R.PriorPoint2(R.MvNormal([1.0, 2.0], np.array([[0.5, 0.1], [0.1, 0.3]]))))
    fg.addFactor(["l1", "l2"], R.Point2Point2(R.MvNormal([3.0, -1.0], 0.2 * np.eye(2))))
    fg.addVariable("p0", R.Pose3); fg.addVariable("p1", R.Pose3)
    fg.addFactor(["p0"], R.PriorPose3(R.MvNormal(np.zeros(6), 0.01 * np.eye(6))))
    S = np.diag([0.1, 0.2, 0.3, 0.01, 0.02, 0.03]); S[0, 4] = S[4, 0] = 0.005
    fg.addFactor(["p0", "p1"], R.Pose3Pose3(R.MvNormal([1, 2, 3, 0.1, 0.2, 0.3], S)))
    R.dead_reckon_init(fg, seed=2)
    fg.bws = {"x1": np.array([0.1, 0.2, 0.03])}
    fg.ppes = {"x1": {"default": {"suggested": np.array([1.0, 2.0, 0.5]), "max": np.array([1.1, 2.1, 0.5]), "mean": np.array([1.0, 2.0, 0.4])}}}
    with tempfile.TemporaryDirectory() as td:
        path = R.saveDFG(fg, os.path.join(td, "fg.tar.gz"))
        g = R.loadDFG(path)
    assert g.ls() == fg.ls() and g.N == 50
    assert [(a, b, type(c).__name__) for a, b, c in g.factors] == [(a, b, type(c).__name__) for a, b, c in fg.factors]
    for (_, _, a), (_, _, b) in zip(fg.factors, g.factors):
        for fld in ("Z", "bearing", "range"):
            if hasattr(a, fld):
                x, y = getattr(a, fld), getattr(b, fld)
                assert type(x) is type(y)
                if isinstance(x, R.MvNormal):
                    assert np.array_equal(x.mu, y.mu) and np.array_equal(x.cov, y.cov)       # column-major cov round trip (asymmetric index check)
                else:
                    assert (x.mu, x.sigma) == (y.mu, y.sigma)
    assert g.multihypo == fg.multihypo
    for l in fg.ls():
        assert g.isInitialized(l) == fg.isInitialized(l)
        if fg.isInitialized(l):
            assert np.array_equal(g.getVal(l), fg.getVal(l))
    assert np.array_equal(g.bws["x1"], fg.bws["x1"]) and np.array_equal(g.ppes["x1"]["default"]["max"], [1.1, 2.1, 0.5])
    with pytest.raises(ValueError):
        R.unpackFactor("Pose2Pose2Bogus", {"Z": {}})
    assert isinstance(R.unpackFactor("PackedPose2Pose2", {"Z": R.packBelief(R.MvNormal(np.zeros(3), np.eye(3)))}), R.Pose2Pose2)
    z = R.unpackBelief({"_type": "IncrementalInference.PackedZeroMeanFullNormal", "cov": [1.0, 0.5, 0.5, 2.0]})
    assert np.array_equal(z.mu, [0, 0]) and np.array_equal(z.cov, [[1, 0.5], [0.5, 2]])
