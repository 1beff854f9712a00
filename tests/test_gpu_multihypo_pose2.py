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


def test_approxconv_on_a_graph_and_graph_tables_refuse():
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
    with pytest.raises(NotImplementedError):
        R.DeviceGraph(fg)                                       # whole-graph tables carry bearing-range hypotheses only
    with pytest.raises(R.RomeError):
        R.conv_pose2pose2(R.make_opts(N=N), mu, cov, b1[None], a[None], dirs=[1], alt=b2[None], hypo_w=[1.5])
