"""Randomised option sweep: the HIP path against the oracle over random particle counts (every particles-per-lane
instantiation), inflation-cycle counts, inflation factors, seeds / stream offsets, in-kernel vs pre-sampled noise, layouts
and nullhypo fractions, for the three relative-factor kernels (closed form and Newton; Nelder-Mead is compared in
test_gpu_parity.py at its own tolerance)."""
import numpy as np
import pytest

import oracle as ro
from test_gpu_parity import _p2_inputs, _p3_inputs, _br_inputs, wrapdiff

pytestmark = pytest.mark.gpu
R = None


@pytest.fixture(scope="module", autouse=True)
def _pkg():
    global R
    import rome_jl_amd
    R = rome_jl_amd
    R.default_context()
    yield


def _opts(rng, N, solver):
    kw = dict(N=N, solver=solver, inflate_cycles=int(rng.integers(0, 6)), inflation=float(rng.choice([0.0, 0.5, 5.0, 50.0])),
              seed=int(rng.integers(1, 2 ** 62)), stream_offset=int(rng.integers(0, 2 ** 40)))
    if rng.uniform() < 0.3:
        kw["nullhypo"] = float(rng.uniform(0.05, 0.9))
    return kw


@pytest.mark.parametrize("trial", range(int(__import__("os").environ.get("ROME_FUZZ_TRIALS", "24"))))
def test_random_configurations(trial):
    rng = np.random.default_rng(9000 + trial)
    N = int(rng.choice([1, 2, 3, 31, 64, 65, 100, 127, 128, 129, 200, 256, 257, 333, 512]))
    C_ = int(rng.integers(1, 12))
    solver = int(rng.integers(0, 2))
    kind = ["p2p2", "br0", "br1", "p3p3"][trial % 4]
    kw = _opts(rng, N, solver)
    use_noise = rng.uniform() < 0.5
    if kind == "p2p2":
        mu, cov, fixed, target, dirs, noise = _p2_inputs(C_, N, 1 + trial)
        nz = noise if use_noise else None
        o = R.make_opts(**kw)
        aos = rng.uniform() < 0.5
        if aos:
            o.layout = R.LAYOUT_AOS
            tr = lambda a: None if a is None else np.ascontiguousarray(a.transpose(0, 2, 1))
            out = R.conv_pose2pose2(o, mu, cov, tr(fixed), tr(target), dirs=dirs, noise=tr(nz)).transpose(0, 2, 1)
        else:
            out = R.conv_pose2pose2(o, mu, cov, fixed, target, dirs=dirs, noise=nz)
        L = np.array([ro.cholesky_lower(c) for c in cov])
        ref = ro.conv_pose2pose2(ro.make_opts(**kw), mu, L, np.concatenate([fixed, target], 0), np.arange(C_), C_ + np.arange(C_), dirs, noise=nz)
        assert np.abs(wrapdiff(out, ref, [2])).max() < 1e-8, (kind, N, kw)
    elif kind in ("br0", "br1"):
        d = int(kind[-1])
        mu, sigma, fixed, target, noise = _br_inputs(C_, N, d, 50 + trial)
        nz = noise if use_noise else None
        o = R.make_opts(**kw)
        if rng.uniform() < 0.5:      # AoS blocks [C][N][dim]
            o.layout = R.LAYOUT_AOS
            tr = lambda a: None if a is None else np.ascontiguousarray(a.transpose(0, 2, 1))
            out = R.conv_pose2point2br(o, d, mu, sigma, tr(fixed), tr(target), noise=tr(nz)).transpose(0, 2, 1)
        else:
            out = R.conv_pose2point2br(o, d, mu, sigma, fixed, target, noise=nz)
        ref = ro.conv_pose2point2br(ro.make_opts(**kw), d, mu, sigma, fixed, target, np.arange(C_), np.arange(C_), noise=nz)
        assert np.abs(wrapdiff(out, ref, [2] if d == 1 else [])).max() < 1e-7, (kind, N, kw)
    else:
        from scipy.spatial.transform import Rotation as Rot
        mu, cov, fixed, target, dirs, noise = _p3_inputs(C_, N, 300 + trial)
        nz = noise if use_noise else None
        o = R.make_opts(**kw)
        mode = rng.integers(0, 3)
        tr = lambda a: None if a is None else np.ascontiguousarray(a.transpose(0, 2, 1))
        if mode == 1:                # AoS coordinates
            o.layout = R.LAYOUT_AOS
            out = R.conv_pose3pose3(o, mu, cov, tr(fixed), tr(target), dirs=dirs, noise=tr(nz)).transpose(0, 2, 1)
        elif mode == 2:              # the reference's native points [t(3), R column-major(9)] per particle
            o.layout = R.LAYOUT_AOS_POINTS
            pts = lambda a: R.coords_to_points(6, tr(a).reshape(-1, 6)).reshape(C_, N, 12)
            res = R.conv_pose3pose3(o, mu, cov, pts(fixed), pts(target), dirs=dirs, noise=tr(nz))
            out = R.points_to_coords(6, res.reshape(-1, 12)).reshape(C_, N, 6).transpose(0, 2, 1)
        else:
            out = R.conv_pose3pose3(o, mu, cov, fixed, target, dirs=dirs, noise=nz)
        L = np.array([ro.cholesky_lower(c) for c in cov])
        ref = ro.conv_pose3pose3(ro.make_opts(**kw), mu, L, np.concatenate([fixed, target], 0), np.arange(C_), C_ + np.arange(C_), dirs, noise=nz)
        ang = (Rot.from_rotvec(out[:, 3:].transpose(0, 2, 1).reshape(-1, 3)).inv() *
               Rot.from_rotvec(ref[:, 3:].transpose(0, 2, 1).reshape(-1, 3))).magnitude()
        # particles whose rotation (result or start point) is within 1e-2 of |ω| = π are compared at the conditioning of the
        # matrix Log the oracle restates from Manifolds (the kernel's quaternion Log is the better conditioned one); with
        # nullhypo a start rotation that close to π also moves the translation entropy R·e_t by that rotation error
        near = (np.linalg.norm(ref[:, 3:], axis=1).reshape(-1) > np.pi - 1e-2) | (np.linalg.norm(target[:, 3:], axis=1).reshape(-1) > np.pi - 1e-2)
        # the spread statistic uses Log(R_0ᵀ R_i): a start belief holding a rotation within 1e-3 of π FROM PARTICLE 0 sits at the
        # "snap to θ = π" threshold of Manifolds' log (cos θ + 1 <= √eps), where a rounding difference flips the branch and moves
        # the scalar spread -- and with it every entropy jitter of that convolution -- by ~1e-7
        rel0 = (Rot.from_rotvec(np.repeat(target[:, 3:, :1], N, axis=2).transpose(0, 2, 1).reshape(-1, 3)).inv() *
                Rot.from_rotvec(target[:, 3:].transpose(0, 2, 1).reshape(-1, 3))).magnitude().reshape(C_, N)
        near = near | np.repeat(rel0.max(axis=1) > np.pi - 1e-3, N)
        dt = np.abs(out[:, :3] - ref[:, :3]).max(axis=1).reshape(-1)
        assert dt[~near].max(initial=0.0) < 1e-8 and dt[near].max(initial=0.0) < 1e-4, (kind, N, kw)
        assert ang[~near].max(initial=0.0) < 1e-8 and ang[near].max(initial=0.0) < 2e-4, (kind, N, kw)


def test_device_measurement_normals_law_independent_of_the_oracle():
    """The in-kernel normal generator judged against N(0, 1) ITSELF -- scipy, no oracle -- on ~1e6 device draws per layout: Pose2
    measurements (d = 3: ONE Philox call per particle pair cut into six 21-bit Box-Muller fields; particles 2j, 2j+1 share the radius
    of the third pair), Point2 (d = 2) and Pose3 (d = 6) with full 32-bit words.  Moments, Kolmogorov-Smirnov, tail mass, and the
    correlation -- of the values AND of their squares (a shared radius shows up there) -- between the same coordinate of neighbouring
    particles.  The draws are read off prior samples with unit covariance (rome_sample_prior*: proposal = mu + L xi)."""
    from scipy import stats
    N = 100
    for d, sample, C_ in ((3, R.sample_priorpose2, 3400), (2, R.sample_priorpoint2, 5000), (6, R.sample_priorpose3, 1700)):
        scale = np.ones(d)
        if d == 3:
            scale[2] = 0.25          # the heading is wrapped to (-pi, pi]: sigma = 0.25 keeps 12 sigma inside
        if d == 6:
            scale[3:] = 0.05         # rotation vector: small angles, Log(Exp(w)) = w
        cov = np.diag(scale ** 2)
        out = sample(R.make_opts(N=N, seed=0x5EED + d), np.zeros((C_, d)), np.tile(cov, (C_, 1, 1)))     # (C, d, N)
        xi = out / scale[None, :, None]
        n = xi.size
        assert n >= 1e6 - 1
        flat = xi.reshape(-1)
        assert abs(flat.mean()) < 4.0 / np.sqrt(n) and abs(flat.var() - 1.0) < 6.0 * np.sqrt(2.0 / n)
        assert abs(stats.kurtosis(flat)) < 0.03 and abs(stats.skew(flat)) < 0.015
        for k in range(d):
            col = xi[:, k, :].reshape(-1)
            ks = stats.kstest(col, "norm").statistic
            assert ks < 2.2 / np.sqrt(col.size), (d, k, ks)                 # (the 1 % critical value is 1.63 / sqrt(n))
            tail = np.mean(np.abs(col) > 3.0)
            assert abs(tail - 2.0 * stats.norm.sf(3.0)) < 6.0 * np.sqrt(2.7e-3 / col.size), (d, k, tail)
            a, b = xi[:, k, 0::2].reshape(-1), xi[:, k, 1::2].reshape(-1)   # the same coordinate of particles 2j and 2j+1
            lim = 5.0 / np.sqrt(a.size)
            assert abs(np.corrcoef(a, b)[0, 1]) < lim, (d, k)
            assert abs(np.corrcoef(a * a, b * b)[0, 1]) < lim, (d, k, np.corrcoef(a * a, b * b)[0, 1])
        # different coordinates of one particle are uncorrelated too (values and squares)
        cc = np.corrcoef(xi.transpose(1, 0, 2).reshape(d, -1)); c2 = np.corrcoef((xi ** 2).transpose(1, 0, 2).reshape(d, -1))
        off = ~np.eye(d, dtype=bool)
        assert np.abs(cc[off]).max() < 5.0 / np.sqrt(C_ * N) and np.abs(c2[off]).max() < 5.0 / np.sqrt(C_ * N)
