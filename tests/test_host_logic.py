"""CPU-only tests: C-ABI surface, host-side mirror of the reference interface, graph packing,
g2o import (reference fixture test/octagon.g2o), generators, oracle self-consistency."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import oracle as ro
import rome_jl_amd as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _has_gpu():
    import torch
    return torch.cuda.is_available()


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rome_mi355.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rome_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = R._lib.load()
    for name in declared:
        assert hasattr(lib, name), "librome_mi355.so does not export %s" % name
    assert declared == set(R._lib.SIGNATURES), declared ^ set(R._lib.SIGNATURES)
    assert lib.rome_version() == 122


def test_struct_layouts_match_header():
    assert ctypes.sizeof(R._lib.Opts) == 72
    assert ctypes.sizeof(R._lib.ConvDev) == 8 + 11 * 8 + 24 + 8 + 16 + 8 + 8 + 8   # ... + rows4 + mirror_map
    o = R.make_opts(solver=R.SOLVER_NELDER_MEAD)
    assert (o.n_particles, o.max_iters, o.inflate_cycles, o.tol, o.inflation) == (100, 1000, 3, 1e-8, 5.0)
    o = R.make_opts(N=64)
    assert (o.solver, o.max_iters, o.tol, o.seed) == (R.SOLVER_NEWTON, 20, 1e-12, 0x524F4D45)


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly (never route through the oracle)."""
    if _has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(R.RomeError) as e:
        R.Context(0)
    assert e.value.code == R._lib.ERR_NO_DEVICE
    with pytest.raises(RuntimeError):
        R.DeviceGraph(R.generateGraph_Hexagonal())
    with pytest.raises(R.RomeError):
        R.residual_pose2pose2([[0, 0, 0]], [[0, 0, 0]], [[0, 0, 0]])
    bel = np.zeros((1, 3, 10)); bel[0, :, 1:] = 1.0
    for call in (lambda: R.kde_bandwidth(bel), lambda: R.kde_max(bel, np.ones((1, 3))), lambda: R.calcPPE(bel), lambda: R.belief_stats(bel),
                 lambda: R.initAll(R.generateGraph_Hexagonal()), lambda: R.solveGraph(R.generateGraph_Hexagonal()),
                 lambda: R.setPPE(R.dead_reckon_init(R.generateGraph_Hexagonal()))):
        with pytest.raises((R.RomeError, RuntimeError)):
            call()
    src = "".join(open(os.path.join(ROOT, "rome.jl_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "rome.jl_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_cholesky_host_helper_matches_oracle_and_numpy():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((5, 6, 6)); cov = A @ A.transpose(0, 2, 1) + np.eye(6)
    L = R.cholesky_lower(cov)
    for k in range(5):
        assert np.allclose(L[k], ro.cholesky_lower(cov[k]), atol=0)
        Lf = np.linalg.cholesky(cov[k])
        assert np.allclose(L[k], Lf[np.tril_indices(6)], atol=1e-13)
    with pytest.raises(R.RomeError) as e:
        R.cholesky_lower(-np.eye(3))
    assert e.value.code == R._lib.ERR_NOT_POSDEF


# ------------------------------------------------------------------ factors / plugin surface
def test_factor_constructors_and_defaults():
    f = R.Pose2Pose2()
    assert np.array_equal(f.Z.cov, np.eye(3))                      # Pose2D.jl:31
    assert np.array_equal(R.PriorPose2().Z.cov, np.diag([1, 1, 0.1]))   # PriorPose2.jl:14
    assert np.allclose(np.diag(R.Pose3Pose3().Z.cov), [0.01] * 3 + [0.0001] * 3)  # Pose3Pose3.jl:10
    with pytest.raises(ValueError):
        R.Pose2Pose2(R.MvNormal(np.zeros(2), np.eye(2)))
    with pytest.raises(TypeError):
        R.Pose2Point2BearingRange(R.MvNormal(np.zeros(3), np.eye(3)), R.Normal(1, 1))


def test_getMeasurementParametric_bearingrange():
    # src/factors/BearingRange2D.jl:30-37
    mu, iS = R.getMeasurementParametric(R.Pose2Point2BearingRange(R.Normal(0.3, 0.1), R.Normal(20.0, 2.0)))
    assert np.allclose(mu, [0.3, 20.0]) and np.allclose(iS, np.diag([100.0, 0.25]))


def test_packed_factor_roundtrip():
    # test/testpackingconverters.jl:51-132 (Packed* <-> factor for the hot-path factors)
    fs = [R.Pose2Pose2(R.MvNormal([1, 2, 0.3], np.diag([0.1, 0.2, 0.3]))), R.PriorPose2(R.MvNormal([0, 0, 0], 0.01 * np.eye(3))),
          R.Pose2Point2BearingRange(R.Normal(0.1, 0.05), R.Normal(20, 1)), R.Pose3Pose3(), R.PriorPose3()]
    for f in fs:
        g = R.unpack_factor(json.loads(json.dumps(R.pack_factor(f))))
        assert type(g) is type(f)
        if hasattr(f, "Z"):
            assert np.array_equal(g.Z.mu, f.Z.mu) and np.array_equal(g.Z.cov, f.Z.cov)
        else:
            assert (g.bearing.mu, g.bearing.sigma, g.range.mu, g.range.sigma) == (f.bearing.mu, f.bearing.sigma, f.range.mu, f.range.sigma)


def test_point_coordinate_layouts():
    c = np.array([1.0, -2.0, 0.7])
    p = R.getPoint(R.Pose2, c)
    assert np.allclose(p, ro.pose2_point(c)) and np.allclose(R.getCoordinates(R.Pose2, p), c)
    c6 = np.array([1.0, 2.0, 3.0, 0.3, -0.2, 0.5])
    p6 = R.getPoint(R.Pose3, c6)
    assert np.allclose(p6, ro.pose3_point(c6), atol=1e-15) and np.allclose(R.getCoordinates(R.Pose3, p6), c6)


# ------------------------------------------------------------------ g2o + graph packing
def test_g2o_import_octagon_fixture():
    # test/testG2oParser.jl:4-20 on the reference's own data file
    ins = R.importG2o(os.path.join(GOLDEN, "octagon.g2o"))
    assert ins[0][0] == "EDGE_SE2" and ins[6][0] == "EDGE_SE2"
    assert ins[5][11] == "6541.252776" and ins[2][6] == "1211.201664"
    assert len(ins) == 8 and len(ins[1]) == 12
    fg = R.initfg()
    for i in ins:
        R.parseG2oInstruction(fg, i)
    assert len(fg.variables) == 8 and len(fg.factors) == 8
    _, labels, f = fg.factors[0]
    assert labels == ["x0", "x1"] and np.allclose(f.Z.mu, [1.0, 0.0, 0.785])
    info = np.array([[3533.219465, 13825.498244, 0.0], [13825.498244, 54832.844537, 0.0], [0.0, 0.0, 6065.357771]])
    cov = np.linalg.inv(info); cov = (cov + cov.T) / 2           # src/services/g2oParser.jl:103-109
    assert np.allclose(f.Z.cov, cov, rtol=1e-12) and np.array_equal(f.Z.cov, f.Z.cov.T)


def test_hexagonal_generator_matches_reference_shape():
    # src/canonical/GenerateCircular.jl:56-90 ; BASELINE.json configs[0]
    fg = R.generateGraph_Hexagonal()
    assert [l for l in fg.variables] == ["x0", "x1", "x2", "x3", "x4", "x5", "x6", "l1"]
    kinds = [type(f).__name__ for _, _, f in fg.factors]
    assert kinds == ["PriorPose2"] + ["Pose2Pose2"] * 6 + ["Pose2Point2BearingRange"] * 2
    _, _, pp = fg.factors[1]
    assert np.allclose(pp.Z.mu, [10.0, 0.0, np.pi / 3]) and np.allclose(pp.Z.cov, np.diag([0.01] * 3))
    assert fg.factors[-1][1] == ["x6", "l1"]


def test_synth_manhattan_shape_and_tables():
    fg = R.synth_manhattan()
    assert len(fg.variables) == 3500 and len(fg.factors) == 5454
    pk = R.PackedGraph(fg)
    assert pk.p2p2["F"] == 5453 and pk.prior2["F"] == 1
    assert (pk.p2p2["var_to"][:3499] - pk.p2p2["var_from"][:3499] == 1).all()      # odometry chain
    assert ((pk.p2p2["var_to"][3499:] - pk.p2p2["var_from"][3499:]) >= 4).all()     # closures i<j
    factor, dr, fixed, target = R.PackedGraph.conv_table(pk.p2p2)
    assert len(factor) == 10906 and dr[:4].tolist() == [0, 1, 0, 1]
    assert fixed[0] == pk.p2p2["var_from"][0] and target[0] == pk.p2p2["var_to"][0]
    assert fixed[1] == pk.p2p2["var_to"][0] and target[1] == pk.p2p2["var_from"][0]
    for k in range(5453):                                              # every Σ positive definite
        np.linalg.cholesky(pk.p2p2["cov"][k])
    fg2 = R.synth_manhattan()
    assert np.array_equal(R.PackedGraph(fg2).p2p2["mu"], pk.p2p2["mu"])  # deterministic


def test_dead_reckon_init_follows_odometry():
    fg = R.generateGraph_Hexagonal(N=50)
    R.dead_reckon_init(fg, sigma=(0, 0, 0))
    assert np.allclose(fg.getVal("x1")[:, 0], [10, 0, np.pi / 3])
    assert np.allclose(fg.getVal("x3")[:2, 0], [10, 17.320508], atol=1e-5)   # test/testParametricSimulated.jl:152


# ------------------------------------------------------------------ oracle self-consistency (solvers agree)
def test_oracle_solvers_agree_pose2pose2():
    rng = np.random.default_rng(4)
    N, V = 100, 12
    bel = rng.standard_normal((V, 3, N)) * np.array([0.2, 0.2, 0.05])[None, :, None] + rng.standard_normal((V, 3, 1)) * [[5], [5], [2]]
    F = V - 1
    mu = rng.standard_normal((F, 3)); L = np.tile(ro.cholesky_lower(np.diag([0.02, 0.01, 0.003])), (F, 1))
    fv = np.repeat(np.arange(F), 2); tv = fv.copy(); dr = np.tile([0, 1], F)
    fv[0::2] = np.arange(F); tv[0::2] = np.arange(F) + 1; fv[1::2] = np.arange(F) + 1; tv[1::2] = np.arange(F)
    fa = np.repeat(np.arange(F), 2)
    outs = [ro.conv_pose2pose2(ro.make_opts(N=N, solver=s, seed=9), mu, L, bel, fv, tv, dr, factor=fa) for s in (0, 1, 2)]
    d01 = outs[0] - outs[1]; d01[:, 2] = np.arctan2(np.sin(d01[:, 2]), np.cos(d01[:, 2]))
    assert np.abs(d01).max() < 1e-11
    d02 = outs[0] - outs[2]; d02[:, 2] = np.arctan2(np.sin(d02[:, 2]), np.cos(d02[:, 2]))
    assert np.median(np.abs(d02).max(axis=1)) < 5e-4      # Optim NM accuracy ~1e-4


def test_oracle_bearingrange_constraint_and_ring():
    # test/TestPoseAndPoint2Constraints.jl:97-105 analogue: landmark proposals lie on the range ring
    N = 100
    rng = np.random.default_rng(2)
    pose = np.zeros((1, 3, N)); pose[0, :2] = 0.01 * rng.standard_normal((2, N))
    lm0 = np.zeros((1, 2, N))
    out = ro.conv_pose2point2br(ro.make_opts(N=N, solver=1), 0, [[0.0, 10.0]], [[3.0, 1.0]], pose, lm0, [0], [0])
    r = np.hypot(out[0, 0], out[0, 1])
    assert (r > 5).all() and (r < 15).all()


def test_oracle_nelder_mead_on_rosenbrock():
    # Optim.jl README example: NelderMead on Rosenbrock from (0,0) converges to (1,1)
    f = lambda x: (1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2
    x, rc, ne = ro.nelder_mead(f, [0.0, 0.0])
    assert rc == 0 and np.allclose(x, [1, 1], atol=1e-3) and 50 < ne < 400


def test_entropy_rng_definition():
    """Entropy uniforms as the oracle defines them: one Philox call per (particle, cycle) (two for d = 6), domain 2,
    counter word 3 = (2 << 16) | (2·cycle + block), one 32-bit word per coordinate, u = (w + 0.5) / 2^32."""
    u = ro.rng_entropy(7, 3, 5, 0, 3); v = ro.rng_entropy(7, 3, 5, 1, 3); w = ro.rng_entropy(7, 3, 5 + 64, 0, 3)
    assert ((u > 0) & (u < 1)).all() and not np.allclose(u, v) and not np.allclose(u, w)
    for part in (5, 5 + 64):
        for cyc in range(4):
            wds = ro.philox([part, 3, 0, (2 << 16) | (2 * cyc)], [7, 0])
            assert np.array_equal(ro.rng_entropy(7, 3, part, cyc, 3), [(x + 0.5) / 2 ** 32 for x in wds[:3]])
            assert np.array_equal(ro.rng_entropy(7, 3, part, cyc, 2), [(x + 0.5) / 2 ** 32 for x in wds[:2]])
            wd2 = ro.philox([part, 3, 0, (2 << 16) | (2 * cyc + 1)], [7, 0])
            assert np.array_equal(ro.rng_entropy(7, 3, part, cyc, 6), [(x + 0.5) / 2 ** 32 for x in wds[:3] + wd2[:3]])
    allu = np.array([ro.rng_entropy(1, 0, i, c, 3) for i in range(400) for c in range(6)])
    assert abs(allu.mean() - 0.5) < 0.02 and abs(allu.std() - 12 ** -0.5) < 0.02


def _bm_exact(wa, wb):
    """the transform ro_box_muller approximates in single precision, evaluated in double precision"""
    wa = np.asarray(wa, dtype=np.uint64); wb = np.asarray(wb, dtype=np.uint64)
    r = np.sqrt(-2 * np.log((wa.astype(np.float64) + 1) / 2 ** 32))
    a = (np.pi / 4) * ((wb << np.uint64(2)) & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32).astype(np.float64) / 2 ** 31
    sx = np.where(wb & np.uint64(0x80000000), -1.0, 1.0); sy = np.where(wb & np.uint64(0x40000000), -1.0, 1.0)
    return sx * r * np.cos(np.pi / 4 + a), sy * r * np.sin(np.pi / 4 + a)


def test_normal_generator_definition():
    """Neighbouring particles 2j, 2j+1 draw from the SAME Philox calls (counter = the even particle id, domain 1)."""
    f21 = lambda f: (f << 11) | 0x400                                    # a 21-bit field enters box_muller as the centre of its bin
    w = ro.philox([4, 3, 0, (1 << 16) | 0], [7, 0])
    # d = 3: ONE call per pair, six 21-bit fields
    a, b = ro.rng_normals(7, 3, 4, 3), ro.rng_normals(7, 3, 5, 3)
    top = [x >> 11 for x in w]
    f4 = ((w[0] & 0x7FF) << 10) | ((w[1] & 0x7FE) >> 1); f5 = ((w[2] & 0x7FF) << 10) | ((w[3] & 0x7FE) >> 1)
    assert np.array_equal(a[:2], ro.box_muller(f21(top[0]), f21(top[1]))) and np.array_equal(b[:2], ro.box_muller(f21(top[2]), f21(top[3])))
    sh = ro.box_muller(f21(f4), f21(f5))
    assert a[2] == sh[0] and b[2] == sh[1] and not np.allclose(a[:2], b[:2])
    e0, e1 = _bm_exact([f21(f4)], [f21(f5)])
    assert abs(sh[0] - e0[0]) < 1e-4 and abs(sh[1] - e1[0]) < 1e-4
    # d = 2: the even particle takes words (0, 1) of the call, the odd one words (2, 3)
    assert np.array_equal(ro.rng_normals(7, 3, 4, 2), ro.box_muller(w[0], w[1])) and np.array_equal(ro.rng_normals(7, 3, 5, 2), ro.box_muller(w[2], w[3]))
    # d = 6: three calls per pair; even = words 0..5, odd = words 6..11
    w1 = ro.philox([4, 3, 0, (1 << 16) | 1], [7, 0]); w2 = ro.philox([4, 3, 0, (1 << 16) | 2], [7, 0])
    assert np.array_equal(ro.rng_normals(7, 3, 4, 6), list(ro.box_muller(w[0], w[1])) + list(ro.box_muller(w[2], w[3])) + list(ro.box_muller(w1[0], w1[1])))
    assert np.array_equal(ro.rng_normals(7, 3, 5, 6), list(ro.box_muller(w1[2], w1[3])) + list(ro.box_muller(w2[0], w2[1])) + list(ro.box_muller(w2[2], w2[3])))
    # edge words: the largest radius (6.66 sigma; 5.52 for a 21-bit field), the zero radius, the four mirror quadrants
    big = ro.box_muller(0, 0x20000000)
    assert abs(np.hypot(*big) - np.sqrt(2 * 32 * np.log(2))) < 1e-5 and ro.box_muller(0xFFFFFFFF, 123) == (0.0, 0.0)
    assert abs(np.hypot(*ro.box_muller(f21(0), 0x20000000)) - np.sqrt(2 * 22 * np.log(2))) < 1e-3
    sg = [tuple(np.sign(ro.box_muller(12345, q << 30 | 0x1234567))) for q in range(4)]
    assert sg == [(1, 1), (1, -1), (-1, 1), (-1, -1)]
    for d in (2, 3, 6):
        nd = np.array([ro.rng_normals(11, 2, i, d) for i in range(4096)])
        assert np.abs(nd.mean(0)).max() < 0.06 and np.abs(nd.std(0) - 1).max() < 0.05
        cc = np.corrcoef(np.concatenate([nd[0::2], nd[1::2]], 1).T)                # the 2d draws of a pair of particles are uncorrelated
        assert np.abs(cc - np.eye(2 * d)).max() < 0.08


def test_normal_generator_law_21bit_fields():
    """Pose2 measurement normals (21-bit radius and angle fields): Kolmogorov distance to N(0,1) below 2e-6 by construction (the bins
    of the radius and angle are 2^-21 / 2^-19 wide); checked here on 6e5 draws against N(0,1) (KS) with exact moments."""
    from scipy import stats
    n3 = np.array([ro.rng_normals(5, s, i, 3) for s in range(100) for i in range(2000)])
    assert stats.kstest(n3.ravel(), "norm").pvalue > 1e-3
    assert np.abs(n3.mean(0)).max() < 0.01 and np.abs(n3.std(0) - 1).max() < 0.01
    assert abs(stats.kurtosis(n3.ravel())) < 0.03 and np.abs(n3).max() < 5.6


def test_normal_generator_law():
    """The single-precision Box-Muller evaluation against the exact transform of the same words (<= 5e-5 absolute, only near
    the zero radius; <= 5e-6 relative on radii above 0.5) and against N(0,1) (KS on 4e5 draws)."""
    from scipy import stats
    rng = np.random.default_rng(5)
    W = rng.integers(0, 2 ** 32, size=(200000, 2), dtype=np.uint64)
    out = np.array([ro.box_muller(int(a), int(b)) for a, b in W])
    e0, e1 = _bm_exact(W[:, 0], W[:, 1])
    ex = np.stack([e0, e1], 1)
    assert np.abs(out - ex).max() < 5e-5
    rad, rex = np.hypot(out[:, 0], out[:, 1]), np.hypot(e0, e1)
    assert (np.abs(rad - rex) / rex)[rex > 0.5].max() < 5e-6
    assert stats.kstest(out.ravel(), "norm").pvalue > 1e-3
    assert abs(np.corrcoef(out.T)[0, 1]) < 0.01 and np.abs(out.mean(0)).max() < 0.01 and np.abs(out.std(0) - 1).max() < 0.01


def test_real_datasets_load():
    """The pose-graph datasets the reference ships as examples/manhattan.g2o (M3500) and examples/MIT.g2o, kept as data
    fixtures under tests/golden/: counts and the record semantics of src/services/g2oParser.jl:98-121."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fg = R.loadG2o(os.path.join(gold, "manhattan.g2o"))
    assert sum(1 for t in fg.variables.values() if t is R.Pose2) == 3500
    p2 = [(l, f) for _, l, f in fg.factors if isinstance(f, R.Pose2Pose2)]
    assert len(p2) == 5453 and sum(1 for _, _, f in fg.factors if isinstance(f, R.PriorPose2)) == 1
    l, f = p2[0]
    assert l == ["x0", "x1"] and np.allclose(f.Z.mu, [1.030390, 0.011350, -0.012958])
    info = np.array([[44.635358, -7.962220, 0.0], [-7.962220, 376.516380, 0.0], [0.0, 0.0, 9745.791650]])
    assert np.allclose(np.linalg.inv(f.Z.cov), info, rtol=1e-9)
    assert sum(1 for l, _ in p2 if int(l[1][1:]) - int(l[0][1:]) != 1) == 5453 - 3499
    pk = R.PackedGraph(fg)
    assert pk.p2p2["F"] == 5453 and len(pk.labels[R.Pose2]) == 3500
    mit = R.loadG2o(os.path.join(gold, "MIT.g2o"))
    assert sum(1 for t in mit.variables.values() if t is R.Pose2) == 808
    assert sum(1 for _, _, f in mit.factors if isinstance(f, R.Pose2Pose2)) == 827


def test_g2o_parser_covariances_match_the_reference_serialisation():
    """The reference's solved-graph artefact (tests/golden/manhattan500_reference_solve.npz) stores, for 500 Manhattan edges, the
    MvNormal(μ, Σ) its own parser built from the g2o records (Σ = inv(Λ), src/services/g2oParser.jl:98-121): parsing the same
    records of the dataset with this package must give the same μ and Σ."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = np.load(os.path.join(gold, "manhattan500_reference_solve.npz"))
    fg = R.loadG2o(os.path.join(gold, "manhattan.g2o"))
    mine = {(int(l[0][1:]), int(l[1][1:])): f for _, l, f in fg.factors if isinstance(f, R.Pose2Pose2)}
    for (i, j), mu, cov in zip(d["edges"], d["mu"], d["cov"]):
        f = mine[(int(i), int(j))]
        assert np.array_equal(f.Z.mu, mu)
        assert np.allclose(f.Z.cov, cov, rtol=1e-10, atol=1e-16)


def test_factor_labels_stay_unique_after_delete():
    """DFG-style labels <vars>f<k>: deleting a factor must not let a later addFactor reuse a label that is still in the graph."""
    fg = R.initfg(8)
    fg.addVariable("x0", R.Pose2); fg.addVariable("x1", R.Pose2)
    z = R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], np.eye(3) * 0.01))
    a = fg.addFactor(["x0", "x1"], z); b = fg.addFactor(["x0", "x1"], z)
    assert (a, b) == ("x0x1f1", "x0x1f2")
    fg.deleteFactor(a)
    c = fg.addFactor(["x0", "x1"], z)
    assert c == "x0x1f1" and len({f[0] for f in fg.factors}) == 2     # the free suffix, not a duplicate of f2
    d = fg.addFactor(["x0", "x1"], z)
    assert d == "x0x1f3"


def test_loadg2o_max_edges_counts_edges_only(tmp_path):
    p = tmp_path / "g.g2o"
    p.write_text("VERTEX_SE2 0 0 0 0\nVERTEX_SE2 1 1 0 0\nVERTEX_SE2 2 2 0 0\n"
                 "EDGE_SE2 0 1 1 0 0 10 0 0 10 0 10\nEDGE_SE2 1 2 1 0 0 10 0 0 10 0 10\nEDGE_SE2 0 2 2 0 0 10 0 0 10 0 10\n")
    fg = R.loadG2o(str(p), N=4, max_edges=2)
    assert sum(1 for _, _, f in fg.factors if isinstance(f, R.Pose2Pose2)) == 2


def test_multihypo_is_accepted_on_pose2pose2_and_carried_by_the_graph_tables():
    """IIF's `multihypo=` keyword on a relative pose factor: labels [a, b1, b2]."""
    fg = R.initfg(10)
    for l in ("a", "b1", "b2"):
        fg.addVariable(l, R.Pose2)
    fg.addVariable("l0", R.Point2)
    f = R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], np.diag([0.01, 0.01, 0.0025])))
    fl = fg.addFactor(["a", "b1", "b2"], f, multihypo=[1.0, 0.6, 0.4])
    assert fg.multihypo[fl] == (0.6, 0.4) and fg.getFactor(fl)[1] == ["a", "b1", "b2"]
    with pytest.raises(TypeError):
        fg.addFactor(["a", "b1", "l0"], f, multihypo=[1.0, 0.5, 0.5])       # the alternative must be a Pose2 too
    with pytest.raises(ValueError):
        fg.addFactor(["a", "b1", "b2"], f, multihypo=[1.0, 0.6, 0.6])
    pk = R.PackedGraph(fg)                                                     # the tables carry the hypotheses
    alt, w, ex = R.PackedGraph.conv_hypotheses(pk.p2p2)
    assert list(alt) == [pk.index["b2"]] * 2 and list(w) == [0.6, 0.6]
    assert list(ex["target"]) == [pk.index["b2"]] and list(ex["alt"]) == [pk.index["b1"]] and list(ex["w"]) == [0.4] and list(ex["dir"]) == [0]
    fg.deleteFactor(fl)
    assert fl not in fg.multihypo and R.PackedGraph(fg).p2p2["F"] == 0 and R.PackedGraph.conv_hypotheses(R.PackedGraph(fg).p2p2) is None


# ------------------------------------------------------------------ ordered up-solve schedules (rome_jl_amd.schedule)
def test_schedule_orderings_and_init_pass_over_oracle_standins():
    """init_rounds / greedy_colouring and the OrderedSolve driver, with the oracle-backed stand-ins of tests/dist_standin.py in place of
    the device plans: every variable is initialised from ALREADY initialised neighbours only, groups are independent sets, a sweep
    visits every variable."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_standin import OracleStore, OraclePlan
    from rome_jl_amd.schedule import OrderedSolve, init_rounds, greedy_colouring, adjacency
    N = 24
    fg = R.generateGraph_Hexagonal(N=N)
    lv, left = init_rounds(fg)
    assert not left
    assert lv[0] == ["x0"] and set(lv[1]) == {"x1", "l1"} and sum(len(x) for x in lv) == len(fg.variables)
    nb = adjacency(fg)
    for cls in greedy_colouring(fg):
        assert all(not (nb[a] & set(cls)) for a in cls)
    store = OracleStore(R, fg)                      # no variable has a belief: the init pass must not need any
    assert not fg.vals
    osv = OrderedSolve(store, kind="colour", plan_cls=OraclePlan)
    assert [g for g in osv.init_groups][0] == ["x0"]
    seen = set()
    for grp, plan in zip(osv.init_groups, osv.init_plans):
        for fl, dest in plan.fp["pairs"]:            # every pair convolves from variables initialised BEFORE this group
            assert all(l in seen for l in fg.getFactor(fl)[1] if l != dest), (fl, dest)
        assert plan.fp["pairs"] or grp == [], grp
        seen |= set(grp)
    assert seen == set(fg.variables)
    osv.init(R.make_opts(N=N, seed=3))
    pts = {l: store.get(l) for l in fg.variables}
    assert all(np.isfinite(p).all() for p in pts.values())
    # the hexagon's geometry comes out of the init pass alone (prior at the origin, 10 m legs, 60 degree turns)
    assert np.abs(pts["x1"][:2].mean(1) - [10.0, 0.0]).max() < 1.5 and np.abs(pts["x3"][:2].mean(1) - [10.0, 17.32]).max() < 4.0
    before = {l: p.copy() for l, p in pts.items()}
    osv.sweep(R.make_opts(N=N, seed=4))
    assert all(not np.array_equal(store.get(l), before[l]) for l in fg.variables)
    assert sorted(l for g in osv.sweep_groups for l in g) == sorted(fg.variables)
    lvl = OrderedSolve(OracleStore(R, fg), kind="levels", plan_cls=OraclePlan)
    assert lvl.sweep_groups[0] == ["x0"] and lvl.sweep_groups[-1] == ["x0"] and len(lvl.sweep_groups) == 2 * len(lvl.init_groups) - 1


def test_bench_dry_run_builds_and_checks_every_rank_of_an_8_gpu_run():
    """`bench.py --gpus 8 --dry-run`: every rank's tables / arena layout / exchange plan on CPU tensors, checked for consistency"""
    import subprocess, sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["n_gpus"] == 8 and out["checks"] > 100 and not out["failed"]
    assert sum(out["strong"]["rows_per_rank"]) == 10907 and out["frontier"]["cliques"] > 900


def test_init_rounds_multihypo_candidates_need_the_certain_variable_only():
    """two candidate landmarks reachable only through [x, l1, l2] multihypo sightings, no priors of their own, must not block
    each other (IIF initialises a fractional variable from the certain one)."""
    from rome_jl_amd.schedule import init_rounds
    fg = R.initfg(16)
    fg.addVariable("x0", R.Pose2); fg.addVariable("l1", R.Point2); fg.addVariable("l2", R.Point2)
    fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), 0.01 * np.eye(3))))
    fg.addFactor(["x0", "l1", "l2"], R.Pose2Point2BearingRange(R.Normal(0.0, 0.03), R.Normal(20.0, 0.5)), multihypo=[1.0, 0.5, 0.5])
    rounds, left = init_rounds(fg)
    assert not left and rounds[0] == ["x0"] and set(rounds[1]) == {"l1", "l2"}
    from rome_jl_amd.clique import frontier_pairs
    for cand in ("l1", "l2"):     # (the two candidates share the factor: they are updated in different groups of the round)
        pairs = frontier_pairs(fg, [[cand]], [cand], usable={"x0"}.__contains__)
        assert [p[1] for p in pairs] == [cand]


# ------------------------------------------------------------------ Bayes tree (rome_jl_amd.tree)
def test_bayes_tree_structure_on_manhattan_3500():
    """elimination order -> cliques -> levels: every variable is frontal exactly once, every factor lives in a clique that holds all its
    variables, the separators of a clique are members of its parent (running intersection), children sit on lower levels; a chain in
    natural order gives the textbook cliques"""
    from rome_jl_amd.tree import BayesTree
    fg = R.loadG2o(os.path.join(ROOT, "tests", "golden", "manhattan.g2o"), N=8)
    t = BayesTree.build(list(fg.variables), [(fl, tuple(ls)) for fl, ls, _ in fg.factors])
    fr = [v for c in t.cliques for v in c.frontals]
    assert sorted(fr) == sorted(fg.variables) and len(set(fr)) == len(fr)
    members = [set(c.frontals) | set(c.separators) for c in t.cliques]
    findex = {fl: ls for fl, ls, _ in fg.factors}
    seen = 0
    for c in t.cliques:
        for fl in c.factors:
            assert set(findex[fl]) <= members[c.id], (c, fl)
            seen += 1
        if c.parent >= 0:
            assert set(c.separators) <= members[c.parent] and c.separators and t.cliques[c.parent].level > c.level
            assert c.id in t.cliques[c.parent].children
        else:
            assert not c.separators
    assert seen == len(fg.factors)
    assert sum(len(l) for l in t.levels) == len(t.cliques) and all(t.cliques[c].level == h for h, l in enumerate(t.levels) for c in l)
    assert len(t.levels) < 80 and max(len(l) for l in t.levels) > 500          # 40 levels, 1140 leaves: the frontier-width profile of DESIGN §12
    # variables constrained to the end of the order land in a root
    t2 = BayesTree.build(list(fg.variables), [(fl, tuple(ls)) for fl, ls, _ in fg.factors], last=("x0",))
    assert t2.cliques[t2.clique_of["x0"]].parent == -1
    # a chain in natural order: (x0 | x1), (x1 | x2), root (x2, x3)
    ch = BayesTree.build(["x0", "x1", "x2", "x3"], [("p", ("x0",)), ("a", ("x0", "x1")), ("b", ("x1", "x2")), ("c", ("x2", "x3"))], order="natural")
    got = sorted((tuple(c.frontals), tuple(c.separators)) for c in ch.cliques)
    assert got == [(("x0",), ("x1",)), (("x1",), ("x2",)), (("x2", "x3"), ())], got
    assert [len(l) for l in ch.levels] == [1, 1, 1] and ch.cliques[ch.clique_of["x0"]].factors == ["p", "a"]


def test_tree_solver_edge_cases_over_the_oracle_backend():
    """a forest (two disconnected components, each with its prior: two roots), a graph without any prior (the relative form refuses: no
    gauge; IIF's form runs from the current values), a single variable"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_standin import OracleTreeBackend
    from rome_jl_amd.tree import TreeSolver
    N = 24
    fg = R.initfg(N)
    for c in "ab":
        for k in range(4):
            fg.addVariable("%s%d" % (c, k), R.Pose2)
        fg.addFactor(["%s0" % c], R.PriorPose2(R.MvNormal(np.array([0.0 if c == "a" else 50.0, 0, 0]), 0.01 * np.eye(3))))
        for k in range(3):
            fg.addFactor(["%s%d" % (c, k), "%s%d" % (c, k + 1)], R.Pose2Pose2(R.MvNormal([10.0, 0, 0.1], 0.01 * np.eye(3))))
        fg.addFactor(["%s0" % c, "%s3" % c], R.Pose2Pose2(R.MvNormal([29.5, 3.0, 0.3], 0.04 * np.eye(3))))
    R.dead_reckon_init(fg, seed=1)
    for msg in ("relative", "marginal"):
        ts = TreeSolver(fg, messages=msg, backend=OracleTreeBackend(R)); ts.upload(); ts.solve(R.make_opts(N=N, seed=2))
        assert sum(1 for c in ts.tree.cliques if c.parent < 0) == 2 and ts.stats()["unreached"] == 0
        assert abs(ts.store.get("a3")[0].mean() - 29.5) < 1.5 and abs(ts.store.get("b3")[0].mean() - 79.5) < 1.5, msg
    fg2 = R.initfg(N)
    for k in range(3):
        fg2.addVariable("x%d" % k, R.Pose2)
    for k in range(2):
        fg2.addFactor(["x%d" % k, "x%d" % (k + 1)], R.Pose2Pose2(R.MvNormal([1.0, 0, 0], 0.01 * np.eye(3))))
    R.dead_reckon_init(fg2, seed=1)
    with pytest.raises(ValueError, match="no gauge"):
        TreeSolver(fg2, messages="relative", backend=OracleTreeBackend(R))
    ts = TreeSolver(fg2, messages="marginal", backend=OracleTreeBackend(R)); ts.upload(); ts.solve(R.make_opts(N=N, seed=2))
    fg3 = R.initfg(N); fg3.addVariable("x0", R.Pose2); fg3.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), 0.01 * np.eye(3))))
    for msg in ("relative", "marginal"):
        ts = TreeSolver(fg3, messages=msg, backend=OracleTreeBackend(R)); ts.upload(); ts.solve(R.make_opts(N=N, seed=2))
        assert 0.05 < ts.store.get("x0")[0].std() < 0.2


def test_entropy_uniforms_law_independent_of_the_oracle_definition():
    """device == oracle on the jitter draws is agreement of two implementations of ONE definition; this is the
    counterweight -- the LAW of the entropy uniforms themselves: U(0, 1) marginally (KS on 1.2e5 draws per coordinate), no correlation
    between the coordinates of a draw, between consecutive inflation cycles of a particle (lag 1 in the cycle index), between
    neighbouring particles, or between neighbouring streams (rows); d = 6 uses a second Philox block: same checks across the blocks."""
    from scipy import stats
    P, Cy, S = 500, 6, 40
    U = np.array([[[ro.rng_entropy(0x524F4D45, s, i, c, 6) for c in range(Cy)] for i in range(P)] for s in range(S)])   # [stream, particle, cycle, coord]
    assert ((U > 0) & (U < 1)).all()
    for d in range(6):
        x = U[..., d].ravel()
        assert stats.kstest(x, "uniform").pvalue > 1e-3, d
        assert abs(x.mean() - 0.5) < 3e-3 and abs(x.var() - 1 / 12) < 1.5e-3
    flat = U.reshape(-1, 6)
    cc = np.corrcoef(flat.T)
    assert np.abs(cc - np.eye(6)).max() < 0.012                                  # coordinates of one draw (incl. across the two blocks)
    lag_cycle = np.corrcoef(U[:, :, :-1, :].ravel(), U[:, :, 1:, :].ravel())[0, 1]
    lag_particle = np.corrcoef(U[:, :-1].ravel(), U[:, 1:].ravel())[0, 1]
    lag_stream = np.corrcoef(U[:-1].ravel(), U[1:].ravel())[0, 1]
    assert max(abs(lag_cycle), abs(lag_particle), abs(lag_stream)) < 0.006, (lag_cycle, lag_particle, lag_stream)
    # the jitter a particle receives over the three default cycles: the sum of three uniforms per coordinate (Irwin-Hall): variance 3/12
    s3 = (U[:, :, :3, :3] - 0.5).sum(axis=2).ravel()
    assert abs(s3.var() - 0.25) < 4e-3 and abs(stats.skew(s3)) < 0.02
