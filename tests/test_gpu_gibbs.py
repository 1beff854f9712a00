"""manifoldProduct on the device (rome_product_gibbs_dev): the multiscale Gibbs product of kernel density estimates that the
reference runs through ⚠AMP / ⚠KDE.jl (Ihler et al., NIPS 2003), against its step-by-step restatement in the oracle
(ro_product_msgibbs) sample by sample, and against what a product of Gaussians must give.  Unseeded upstream: statistical pins only
(the reference-side windows are exercised through DeviceGraph.solve in tests/test_gpu_solve.py)."""
import ctypes as C

import numpy as np
import pytest

import oracle as ro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _pkg():
    global R, torch
    import torch as _t
    import rome_jl_amd
    R, torch = rome_jl_amd, _t
    R.default_context()
    yield


def _device_product(dim, N, ptr, rows, prop, bw, bel_in, circ, iters=1, seed=11, stream_offset=5):
    from rome_jl_amd import _lib
    ctx = R.default_context()
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    tp, tb, ti = t(prop, torch.float64), t(bw, torch.float64), t(bel_in, torch.float64)
    tptr, trows = t(ptr, torch.int32), t(rows, torch.int32)
    out = torch.empty_like(ti)
    o = R.make_opts(N=N, seed=seed, stream_offset=stream_offset)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.load().rome_product_gibbs_dev(ctx.handle, C.byref(o), dim, len(ptr) - 1, tptr.data_ptr(), trows.data_ptr(), tp.data_ptr(),
                                                  tb.data_ptr(), len(prop), ti.data_ptr(), out.data_ptr(), circ, iters, int(max(1, np.diff(ptr).max()))),
               ctx.handle)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _problem(dim, N, Ks, rng, circ):
    props = []
    ptr = [0]
    for K in Ks:
        centre = rng.normal(0, 3, dim)
        for _ in range(K):
            mu = centre + rng.normal(0, 0.3, dim)
            sd = rng.uniform(0.05, 0.8, dim)
            P = mu[:, None] + sd[:, None] * rng.standard_normal((dim, N))
            if rng.random() < 0.3:   # a bimodal proposal
                P[0, rng.random(N) < 0.4] += 2.5
            if circ:
                P[dim - 1] = np.arctan2(np.sin(P[dim - 1] + 3.0), np.cos(P[dim - 1] + 3.0))   # headings straddling ±π
            props.append(P)
        ptr.append(ptr[-1] + K)
    prop = np.stack(props)
    rows = rng.permutation(len(props)).astype(np.int32)          # proposals scattered over the table
    prop = prop[np.argsort(rows)]                                # row rows[k] holds the k-th proposal of the CSR order
    return np.array(ptr, dtype=np.int32), rows, prop


@pytest.mark.parametrize("dim,N,iters", [(3, 100, 1), (3, 100, 2), (2, 100, 1), (3, 64, 1), (3, 37, 1), (3, 128, 1), (2, 5, 1), (3, 2, 1),
                                         # 128 < N <= 256: the 256-slot instantiation (the reference runs N = 150 / 200 in places:
                                         # test/testPoint2Point2Init.jl:12, test/testPartialRangeCrossCorrelations.jl:17)
                                         (3, 129, 1), (3, 150, 1), (2, 200, 1), (3, 200, 2), (3, 255, 1), (3, 256, 1), (2, 256, 1)])
def test_device_equals_oracle_sample_by_sample(dim, N, iters):
    rng = np.random.default_rng(100 * dim + N + iters)
    circ = 0b100 if dim == 3 else 0
    Ks = [2, 3, 0, 1, 5, 11, 2, 4, 7, 3]
    ptr, rows, prop = _problem(dim, N, Ks, rng, bool(circ))
    bw = ro.kde_bandwidths(prop, circ) if N > 2 else np.full((len(prop), dim), 0.3)
    bel_in = rng.standard_normal((len(Ks), dim, N))
    got = _device_product(dim, N, ptr, rows, prop, bw, bel_in, circ, iters)
    o = ro.make_opts(N=N, seed=11, stream_offset=5)
    ref = ro.product_msgibbs(o, dim, ptr, rows, prop, bw, bel_in, circ, iters)
    assert np.isfinite(got).all()
    d = got - ref
    if circ:
        d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    same = np.abs(d).max(axis=1) < 1e-9                       # (variable, sample)
    # a categorical draw can differ only when a uniform lands within rounding of a cumulative boundary
    assert same.mean() > 0.995, same.mean()
    assert np.array_equal(got[2], bel_in[2])                   # K = 0: the belief is kept
    assert np.array_equal(got[3], prop[rows[ptr[3]]])          # K = 1: the proposal itself (AMP returns it unchanged)


@pytest.mark.parametrize("dim,N,circ", [(3, 99, 0b011), (3, 33, 0), (2, 101, 0b10), (2, 1, 0), (3, 1, 0b100), (3, 3, 0b100)])
def test_unusual_masks_odd_counts_and_single_particles(dim, N, circ):
    """The kernels are instantiated for the circular masks of the variable types (none; Pose2's heading) and once per dimension for
    any other mask read at run time; odd N leaves the last candidate pair of the leaf level half empty; N = 1 has no tree levels."""
    rng = np.random.default_rng(1000 * dim + 10 * N + circ)
    Ks = [2, 3, 1, 0, 6, 2]
    ptr, rows, prop = _problem(dim, N, Ks, rng, False)
    for d in range(dim):
        if (circ >> d) & 1:
            prop[:, d] = np.arctan2(np.sin(prop[:, d] + 3.0), np.cos(prop[:, d] + 3.0))
    bw = rng.uniform(0.05, 0.4, (len(prop), dim))
    bel_in = rng.standard_normal((len(Ks), dim, N))
    got = _device_product(dim, N, ptr, rows, prop, bw, bel_in, circ, 1)
    ref = ro.product_msgibbs(ro.make_opts(N=N, seed=11, stream_offset=5), dim, ptr, rows, prop, bw, bel_in, circ, 1)
    assert np.isfinite(got).all()
    d = got - ref
    for k in range(dim):
        if (circ >> k) & 1:
            d[:, k] = np.arctan2(np.sin(d[:, k]), np.cos(d[:, k]))
    assert (np.abs(d).max(axis=1) < 1e-9).mean() > 0.99


def test_many_proposals_per_variable_dynamic_lds_beyond_48k():
    """A variable with 24 proposals: the LDS of its block (one level image per proposal) passes 48 kB -- the launch raises the
    kernel's dynamic-LDS limit -- and the counting sort of the dispatch order sees a count far from the others."""
    rng = np.random.default_rng(77)
    N, Ks = 100, [24, 2, 3]
    ptr, rows, prop = _problem(3, N, Ks, rng, True)
    bw = rng.uniform(0.1, 0.4, (len(prop), 3))
    bel_in = np.zeros((len(Ks), 3, N))
    got = _device_product(3, N, ptr, rows, prop, bw, bel_in, 0b100, 1)
    ref = ro.product_msgibbs(ro.make_opts(N=N, seed=11, stream_offset=5), 3, ptr, rows, prop, bw, bel_in, 0b100, 1)
    d = got - ref
    d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    assert np.isfinite(got).all() and (np.abs(d).max(axis=1) < 1e-9).mean() > 0.99


def test_proposals_far_apart_stay_finite_and_equal_the_oracle():
    """Proposals hundreds of bandwidths apart: every candidate weight of a draw underflows against the running maximum (the
    clamped exponential); the draw must still follow the oracle and stay finite -- the product sits between the proposals."""
    rng = np.random.default_rng(5)
    N, V = 100, 40
    prop = np.empty((2 * V, 3, N))
    prop[0::2] = np.array([0.0, 0.0, 0.0])[None, :, None] + 0.05 * rng.standard_normal((V, 3, N))
    prop[1::2] = np.array([40.0, -25.0, 1.0])[None, :, None] + 0.05 * rng.standard_normal((V, 3, N))
    bw = np.full((2 * V, 3), 0.03)
    ptr = np.arange(0, 2 * V + 1, 2, dtype=np.int32); rows = np.arange(2 * V, dtype=np.int32)
    got = _device_product(3, N, ptr, rows, prop, bw, np.zeros((V, 3, N)), 0b100)
    ref = ro.product_msgibbs(ro.make_opts(N=N, seed=11, stream_offset=5), 3, ptr, rows, prop, bw, np.zeros((V, 3, N)), 0b100, 1)
    assert np.isfinite(got).all()
    assert (np.abs(got - ref).max(axis=1) < 1e-9).mean() > 0.99
    assert np.abs(got[:, 0].mean() - 20.0) < 1.0 and np.abs(got[:, 1].mean() + 12.5) < 1.0


def test_product_of_gaussian_densities_has_the_gaussian_product_moments():
    rng = np.random.default_rng(7)
    N, V = 100, 300
    m1, s1 = np.array([0.0, 0.0, 0.1]), np.array([1.0, 0.5, 0.1])
    m2, s2 = np.array([1.0, 0.5, 0.3]), np.array([0.5, 1.0, 0.2])
    prop = np.empty((2 * V, 3, N))
    prop[0::2] = m1[None, :, None] + s1[None, :, None] * rng.standard_normal((V, 3, N))
    prop[1::2] = m2[None, :, None] + s2[None, :, None] * rng.standard_normal((V, 3, N))
    bw = R.kde_bandwidth(prop)
    ptr = np.arange(0, 2 * V + 1, 2, dtype=np.int32); rows = np.arange(2 * V, dtype=np.int32)
    out = _device_product(3, N, ptr, rows, prop, bw, np.zeros((V, 3, N)), 0b100)
    # KDE smoothing widens each factor to s² + h²; the product of the smoothed Gaussians is the target
    h1, h2 = bw[0::2].mean(0), bw[1::2].mean(0)
    v1, v2 = s1 ** 2 + h1 ** 2, s2 ** 2 + h2 ** 2
    P = 1 / v1 + 1 / v2
    mean, sd = (m1 / v1 + m2 / v2) / P, 1 / np.sqrt(P)
    assert np.abs(out.mean((0, 2)) - mean).max() < 0.03, (out.mean((0, 2)), mean)
    pooled = np.sqrt(((out - out.mean(2, keepdims=True)) ** 2).mean((0, 2)) * N / (N - 1))
    assert np.abs(pooled / sd - 1).max() < 0.08, (pooled, sd)


def _exact_product_samples(P, H, n, rng):
    """n exact draws from the product of two kernel density estimates (Euclidean coordinates): the product of two mixtures of N
    Gaussians is a mixture of N² Gaussians with closed-form weights, means and variances.  P: (2, D, N), H: (2, D)."""
    D, N = P.shape[1], P.shape[2]
    logw = np.zeros((N, N))
    for d in range(D):
        logw += -0.5 * (P[0, d][:, None] - P[1, d][None, :]) ** 2 / (H[0, d] ** 2 + H[1, d] ** 2)
    w = np.exp(logw - logw.max()).ravel(); w /= w.sum()
    idx = rng.choice(N * N, size=n, p=w); i, j = idx // N, idx % N
    out = np.empty((D, n))
    for d in range(D):
        p1, p2 = 1 / H[0, d] ** 2, 1 / H[1, d] ** 2
        out[d] = (P[0, d][i] * p1 + P[1, d][j] * p2) / (p1 + p2) + rng.standard_normal(n) / np.sqrt(p1 + p2)
    return out


def test_against_exact_sampling_of_the_product_mixture():
    """A pin that does not go through the oracle: 30 000 draws of the device's multiscale Gibbs product against 30 000 EXACT draws
    of the same product density (two-sample Kolmogorov-Smirnov per coordinate).  Overlapping unimodal proposals: one Gibbs sweep
    (the reference's Niter = 1) already matches.  Proposals with two modes each: the relative weight of the surviving modes needs a
    few sweeps -- with one sweep the labels inherited from the coarse levels are still felt (a property of the published algorithm,
    which the reference shares at its default); with three the sampler matches the exact product."""
    from scipy import stats
    rng = np.random.default_rng(1)
    N, V = 100, 300
    ptr = np.arange(0, 2 * V + 1, 2, dtype=np.int32); rows = np.arange(2 * V, dtype=np.int32)

    def device_draws(P, H, iters):
        got = _device_product(2, N, ptr, rows, np.tile(P, (V, 1, 1)), np.tile(H, (V, 1)), np.zeros((V, 2, N)), 0, iters)
        return got.transpose(1, 0, 2).reshape(2, -1)          # every variable has the same two proposals and its own Philox stream

    A = np.stack([rng.normal(0, 1, N), rng.normal(0, 0.5, N)]); B = np.stack([rng.normal(0.8, 0.7, N), rng.normal(-0.3, 0.8, N)])
    P = np.stack([A, B]); H = R.kde_bandwidth(P, 0)
    ex = _exact_product_samples(P, H, V * N, rng)
    g = device_draws(P, H, 1)
    assert max(stats.ks_2samp(g[d], ex[d]).statistic for d in range(2)) < 0.025
    A = np.stack([np.concatenate([rng.normal(-2, 0.3, N // 2), rng.normal(2, 0.3, N - N // 2)]), rng.normal(0, 0.5, N)])
    B = np.stack([np.concatenate([rng.normal(2.2, 0.3, N // 3), rng.normal(-1.7, 0.3, N - N // 3)]), rng.normal(0.2, 0.5, N)])
    P = np.stack([A, B]); H = R.kde_bandwidth(P, 0)
    ex = _exact_product_samples(P, H, V * N, rng)
    g3 = device_draws(P, H, 3)
    assert max(stats.ks_2samp(g3[d], ex[d]).statistic for d in range(2)) < 0.025
    assert abs((g3[0] > 0).mean() - (ex[0] > 0).mean()) < 0.02
    g1 = device_draws(P, H, 1)                                  # one sweep: both modes are there, widths right, weights not yet mixed
    assert 0.2 < (g1[0] > 0).mean() < 0.8 and stats.ks_2samp(g1[1], ex[1]).statistic < 0.04


def test_manifold_product_wrapper_and_multimodal_selection():
    """Two bimodal densities that agree on ONE mode only: the product keeps that mode (what an importance product on one
    density's points also does, but here from the Gibbs labels), and `manifoldProduct` of a single density returns its points."""
    rng = np.random.default_rng(3)
    N = 100
    a = np.concatenate([rng.normal(-3, 0.2, N // 2), rng.normal(2, 0.2, N - N // 2)])
    b = np.concatenate([rng.normal(2.1, 0.2, N // 2), rng.normal(6, 0.2, N - N // 2)])
    P = np.stack([np.stack([a, rng.normal(0, 0.2, N)]), np.stack([b, rng.normal(0, 0.2, N)])])
    x = R.manifoldProduct(P)
    assert x.shape == (2, N) and (np.abs(x[0] - 2.05) < 1.0).mean() > 0.97
    assert np.array_equal(R.manifoldProduct(P[:1]), P[0])


def test_solve_loop_with_the_reference_product_hexagonal():
    """DeviceGraph.solve(product="gibbs") -- convolutions, `manikde!` bandwidths, multiscale Gibbs product, all on the device --
    against the oracle's restatement of the same loop (pose means within north_star's 1e-3; nearly all particles identical),
    and against the reference's own acceptance windows for this graph (test/testHexagonal2D_CliqByCliq.jl:37-79)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from solve_ref import solve_ref
    N, S = 100, 4
    fg = R.generateGraph_Hexagonal(N=N)
    R.dead_reckon_init(fg, seed=5)
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    dg.solve(R.make_opts(N=N, solver=1, seed=77), n_sweeps=S, product="gibbs")
    b2, bl = solve_ref(R, fg, S, N, seed=77, product="gibbs")
    got2 = dg.bel[R.Pose2].cpu().numpy()
    d = got2 - b2; d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    m2, _ = dg.belief_stats(R.Pose2)
    for v in range(b2.shape[0]):
        mo, _ = ro.belief_spread(b2[v])
        dm = m2[v].cpu().numpy() - mo; dm[2] = np.arctan2(np.sin(dm[2]), np.cos(dm[2]))
        assert np.abs(dm).max() < 1e-3, (v, dm)
    assert np.mean(np.abs(d) < 1e-6) > 0.9
    dg.solve(R.make_opts(N=N, solver=1, seed=2026), n_sweeps=8, product="gibbs")
    b = dg.bel[R.Pose2].cpu().numpy(); l = dg.bel[R.Point2].cpu().numpy()
    truth = [(0, 0, 0), (10, 0, np.pi / 3), (15, 8.66, 2 * np.pi / 3), (10, 17.32, np.pi), (0, 17.32, -2 * np.pi / 3),
             (-5, 8.66, -np.pi / 3), (0, 0, 0)]
    for k, (x, y, th) in enumerate(truth):
        dth = np.arctan2(np.sin(b[k, 2] - th), np.cos(b[k, 2] - th))
        inbox = (np.abs(b[k, 0] - x) < 3) & (np.abs(b[k, 1] - y) < 3) & (np.abs(dth) < 0.3)
        assert inbox.sum() > 35, (k, inbox.sum(), b[k].mean(axis=1))
    assert ((np.abs(l[0, 0] - 20) < 3) & (np.abs(l[0, 1]) < 3)).sum() > 35


def test_reference_solved_graph_product_consistency():
    """The reference's own solved Manhattan-500 graph (examples/fg-after-solve.tar.gz): for every variable, convolve the
    reference's posteriors of its neighbours through the factors (device), take the multiscale Gibbs product of the proposals
    (device) and compare with the reference's posterior of that variable -- the fixed point its solveTree! had reached.  The
    product must sit on the reference posterior (normalised mean offset) and have a comparable width; the round-1 importance
    product is held to the same windows, so the two products can be compared on reference data."""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manhattan500_reference_solve.npz"))
    bel = np.ascontiguousarray(fx["particles"].astype(np.float64).transpose(0, 2, 1))     # [361, 3, 100]
    N = bel.shape[2]
    fg = R.initfg(N)
    labels = ["x%d" % k for k in range(bel.shape[0])]
    for l in labels:
        fg.addVariable(l, R.Pose2)
    idx = {l: k for k, l in enumerate(labels)}
    for (a, b), mu, cov in zip(fx["edges"], fx["mu"], fx["cov"]):
        fg.addFactor([labels[int(a)], labels[int(b)]], R.Pose2Pose2(R.MvNormal(mu, cov)))
    fg.addFactor([labels[0]], R.PriorPose2(R.MvNormal(fx["prior_mu"], fx["prior_cov"])))
    for k, l in enumerate(labels):
        fg.initVariable(l, bel[k])
    res = {}
    for product in ("gibbs", "importance"):
        dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
        o = R.make_opts(N=N, solver=1, seed=5)
        dg.conv_step(o, 0); dg.product_step(o, 0, "lcv", product)
        new = dg.bel[R.Pose2].cpu().numpy()
        sd_ref = bel[:, :2].std(axis=2)
        off = np.abs(new[:, :2].mean(axis=2) - bel[:, :2].mean(axis=2)) / np.maximum(sd_ref, 1e-3)
        ratio = new[:, :2].std(axis=2) / np.maximum(sd_ref, 1e-3)
        res[product] = (np.median(off), np.percentile(off, 95), np.median(ratio))
    g = res["gibbs"]
    print("reference-solve consistency (median offset/sd, 95 %% offset/sd, median sd ratio):", res)
    assert g[0] < 1.0 and g[1] < 4.0, res            # the product of the neighbours' messages sits on the reference posterior
    assert 0.4 < g[2] < 1.6, res                      # ... with a comparable width


@pytest.mark.parametrize("N", [100, 200])
def test_pose3_product_device_equals_oracle_and_gaussian_moments(N):
    """SE(3): the rotation coordinates of every proposal live in the chart at the rotation of its point 0; trees and candidate
    evaluations are Euclidean there, the product Gaussians of selected nodes change charts by Exp / Log.  Device (quaternions) ==
    oracle (rotation matrices) sample by sample; the product of two Gaussian densities on SE(3) has the Gaussian-product moments."""
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(2)
    V = 60

    def make(mu, sd):
        t = np.asarray(mu[:3])[:, None] + np.asarray(sd[:3])[:, None] * rng.standard_normal((3, N))
        w = (Rot.from_rotvec(mu[3:]) * Rot.from_rotvec((np.asarray(sd[3:])[:, None] * rng.standard_normal((3, N))).T)).as_rotvec().T
        return np.concatenate([t, w])
    m1, s1 = np.array([0, 0, 0, 0.2, 2.9, -0.1]), np.array([1.0, 0.5, 0.2, 0.05, 0.1, 0.02])
    m2, s2 = np.array([1.0, 0.5, 0.1, 0.25, 2.95, -0.05]), np.array([0.5, 1.0, 0.3, 0.1, 0.05, 0.04])
    Ks = [2] * (V - 3) + [3, 1, 0]
    props = []
    for K in Ks:
        for k in range(K):
            props.append(make(m1, s1) if k % 2 == 0 else make(m2, s2))
    prop = np.stack(props)
    ptr = np.concatenate([[0], np.cumsum(Ks)]).astype(np.int32); rows = np.arange(len(prop), dtype=np.int32)
    bw = R.kde_bandwidth(prop, 0b111000)
    bel_in = rng.standard_normal((V, 6, N))
    got = _device_product(6, N, ptr, rows, prop, bw, bel_in, 0)
    ref = ro.product_msgibbs(ro.make_opts(N=N, seed=11, stream_offset=5), 6, ptr, rows, prop, bw, bel_in, 0, 1)
    assert np.isfinite(got).all()
    dR = (Rot.from_rotvec(got.transpose(0, 2, 1).reshape(-1, 6)[:, 3:]).inv() * Rot.from_rotvec(ref.transpose(0, 2, 1).reshape(-1, 6)[:, 3:])).magnitude()
    same = (np.abs(got[:, :3] - ref[:, :3]).max(axis=1).reshape(-1) < 1e-9) & (dR < 1e-9)
    assert same.mean() > 0.995, same.mean()
    assert np.array_equal(got[V - 1], bel_in[V - 1]) and np.array_equal(got[V - 2], prop[ptr[V - 2]])
    two = got[:V - 3]
    h = bw[:2]
    v1, v2 = s1 ** 2 + h[0] ** 2, s2 ** 2 + h[1] ** 2
    P = 1 / v1 + 1 / v2
    assert np.abs(two[:, :3].mean((0, 2)) - ((m1 / v1 + m2 / v2) / P)[:3]).max() < 0.05
    R1 = Rot.from_rotvec(m1[3:])
    dev = (R1.inv() * Rot.from_rotvec(two.transpose(0, 2, 1).reshape(-1, 6)[:, 3:])).as_rotvec()
    exp_dev = ((R1.inv() * Rot.from_rotvec(m2[3:])).as_rotvec() / v2[3:]) / P[3:]
    assert np.abs(dev.mean(0) - exp_dev).max() < 0.01 and np.abs(dev.std(0) * np.sqrt(P[3:]) - 1).max() < 0.12
    x = R.manifoldProduct(prop[:2])
    assert x.shape == (6, N) and np.isfinite(x).all()
