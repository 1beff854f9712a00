"""numpy-facing wrappers of the host-pointer C-ABI entry points (include/rome_mi355.h).

`calcFactorResidualTemporary` mirrors the IIF helper the reference's known-answer tests use
(test/testBearingRange2D.jl:64, test/testParametricSimulated.jl:40, test/testPartialPose3.jl:430).
"""
import ctypes as C

import numpy as np

from . import _lib
from .factors import (Pose2, Point2, Pose3, Pose2Pose2, PriorPose2, Pose2Point2BearingRange, Pose3Pose3,
                      PriorPose3, getCoordinates)

_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)

_default_ctx = None


def default_context():
    """Process-wide Context on device 0 (raises without a GPU: no CPU fallback)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = _lib.Context(0)
    return _default_ctx


def _d(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("expected array of shape %s, got %s" % (tuple(shape), a.shape))
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(_PD)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_PI)


# ------------------------------------------------------------------ residuals
def _rows(fn, ctx, ins, widths, out_w):
    ctx = ctx or default_context()
    arrs = [np.atleast_2d(_d(a)) for a in ins]
    n = arrs[0].shape[0]
    for a, w in zip(arrs, widths):
        if a.shape != (n, w):
            raise ValueError("expected (%d,%d) rows, got %s" % (n, w, a.shape))
    out = np.empty((n, out_w))
    _lib.check(fn(ctx.handle, n, *[_p(a) for a in arrs], _p(out)), ctx.handle)
    return out


def residual_pose2pose2(z, p, q, ctx=None):
    return _rows(_lib.load().rome_residual_pose2pose2, ctx, (z, p, q), (3, 3, 3), 3)


def residual_priorpose2(m, p, ctx=None):
    return _rows(_lib.load().rome_residual_priorpose2, ctx, (m, p), (3, 3), 3)


def residual_pose2point2br(z, p, l, ctx=None):
    return _rows(_lib.load().rome_residual_pose2point2br, ctx, (z, p, l), (2, 3, 2), 2)


def residual_pose2point2br_pt(z, p_pt, l, ctx=None):
    return _rows(_lib.load().rome_residual_pose2point2br_pt, ctx, (z, p_pt, l), (2, 6, 2), 2)


def residual_pose3pose3(z, p, q, ctx=None):
    return _rows(_lib.load().rome_residual_pose3pose3, ctx, (z, p, q), (6, 6, 6), 6)


def residual_pose3pose3_pt(z, p_pt, q_pt, ctx=None):
    return _rows(_lib.load().rome_residual_pose3pose3_pt, ctx, (z, p_pt, q_pt), (6, 12, 12), 6)


def residual_priorpose3(m, p, ctx=None):
    return _rows(_lib.load().rome_residual_priorpose3, ctx, (m, p), (6, 6), 6)


def _meas_coords(factor, meas):
    """Accepts the reference's tangent containers (hat form) or plain coordinates."""
    m = np.asarray(meas, dtype=np.float64).ravel()
    if isinstance(factor, (Pose2Pose2, PriorPose2)):
        if m.size == 3:
            return m
        if m.size == 6:  # ((x,y), [0 -θ; θ 0]) column-major
            return np.array([m[0], m[1], m[3]])
    elif isinstance(factor, Pose2Point2BearingRange):
        if m.size == 2:
            return m
        if m.size == 5:  # ([0 -b; b 0], [ρ])
            return np.array([m[1], m[4]])
    elif isinstance(factor, (Pose3Pose3, PriorPose3)):
        if m.size == 6:
            return m
        if m.size == 12:  # (t, skew) -> (X32, X13, X21)
            return np.array([m[0], m[1], m[2], m[3 + 5], m[3 + 6], m[3 + 1]])
    raise ValueError("measurement of unexpected length %d for %s" % (m.size, type(factor).__name__))


def calcFactorResidualTemporary(factor, vartypes, meas, points, ctx=None):
    """Residual of `factor` for one measurement and one point per variable (native point layouts,
    or coordinate vectors of the variable's dimension)."""
    z = _meas_coords(factor, meas)
    pts = [np.asarray(p, dtype=np.float64).ravel() for p in points]
    if isinstance(factor, Pose2Pose2):
        c = [p if p.size == 3 else getCoordinates(Pose2, p) for p in pts]
        return residual_pose2pose2([z], [c[0]], [c[1]], ctx)[0]
    if isinstance(factor, PriorPose2):
        # prior residual is evaluated between the sampled measurement POINT and the variable
        c = pts[0] if pts[0].size == 3 else getCoordinates(Pose2, pts[0])
        return residual_priorpose2([z], [c], ctx)[0]
    if isinstance(factor, Pose2Point2BearingRange):
        if pts[0].size == 6:
            return residual_pose2point2br_pt([z], [pts[0]], [pts[1]], ctx)[0]
        return residual_pose2point2br([z], [pts[0]], [pts[1]], ctx)[0]
    if isinstance(factor, Pose3Pose3):
        if pts[0].size == 12:
            return residual_pose3pose3_pt([z], [pts[0]], [pts[1]], ctx)[0]
        return residual_pose3pose3([z], [pts[0]], [pts[1]], ctx)[0]
    if isinstance(factor, PriorPose3):
        c = pts[0] if pts[0].size == 6 else getCoordinates(Pose3, pts[0])
        return residual_priorpose3([z], [c], ctx)[0]
    raise TypeError("unsupported factor type %s" % type(factor).__name__)


# ------------------------------------------------------------------ parametric linearisation
_LIN_DIMS = {_lib.FACTOR_PRIORPOSE2: (3, 3, 3, 0), _lib.FACTOR_POSE2POSE2: (3, 3, 3, 3), _lib.FACTOR_POSE2POINT2BR: (2, 2, 3, 2),
             _lib.FACTOR_PRIORPOINT2: (2, 2, 2, 0), _lib.FACTOR_POSE3POSE3: (6, 6, 6, 6), _lib.FACTOR_PRIORPOSE3: (6, 6, 6, 0)}


def linearize(kind, mu, W, xa, xb=None, ctx=None):
    """Whitened residuals and Jacobians of F factors of one kind (rome_linearize):
    -> r (F,dr), Ja (F,dr,da), Jb (F,dr,db) or None."""
    ctx = ctx or default_context()
    dz, dr, da, db = _LIN_DIMS[kind]
    mu = np.atleast_2d(_d(mu)); F = mu.shape[0]
    mu = _d(mu, (F, dz)); W = _d(W, (F, dr, dr)); xa = _d(xa, (F, da))
    r = np.empty((F, dr)); Ja = np.empty((F, dr, da))
    if db:
        xb = _d(xb, (F, db)); Jb = np.empty((F, dr, db))
    else:
        xb = None; Jb = None
    _lib.check(_lib.load().rome_linearize(ctx.handle, int(kind), F, _p(mu), _p(W), _p(xa), _p(xb), _p(r), _p(Ja), _p(Jb)), ctx.handle)
    return r, Ja, Jb


def belief_stats(bel, ctx=None):
    """bel (V, dim, N) host array -> (mean (V,dim), std (V,dim)) via rome_belief_stats."""
    ctx = ctx or default_context()
    bel = _d(bel)
    V, d, N = bel.shape
    mean = np.empty((V, d)); sd = np.empty((V, d))
    _lib.check(_lib.load().rome_belief_stats(ctx.handle, d, V, N, _p(bel), _p(mean), _p(sd)), ctx.handle)
    return mean, sd


def kde_bandwidth(bel, circular_mask=None, tol_euclid=0.0, tol_circular=0.0, ctx=None):
    """bel (V, dim, N) host array -> (V, dim) bandwidths as `manikde!` selects them (leave-one-out likelihood
    cross-validation per coordinate), via rome_kde_bandwidth.  circular_mask: bit k set = coordinate k is an angle
    (default: 0b100 for dim 3 -- Pose2 --, 0 otherwise); tol_* = 0 selects the reference's stopping rules."""
    ctx = ctx or default_context()
    bel = _d(bel)
    V, d, N = bel.shape
    if circular_mask is None:
        circular_mask = 0b100 if d == 3 else 0
    bw = np.empty((V, d))
    _lib.check(_lib.load().rome_kde_bandwidth(ctx.handle, d, V, N, _p(bel), int(circular_mask), float(tol_euclid),
                                              float(tol_circular), _p(bw)), ctx.handle)
    return bw


def manifoldProduct(proposals, circular_mask=None, bandwidths=None, Niter=1, opts=None, ctx=None):
    """⚠AMP `manifoldProduct(ff, manifold; Niter=1)`: N samples from the product of K kernel density estimates by multiscale Gibbs
    sampling (rome_product_gibbs_dev).  proposals (K, dim, N) host array (dim 2: Point2, 3: Pose2, 6: Pose3 coordinates (t, ω));
    bandwidths (K, dim) or None = the `manikde!` rule (kde_bandwidth; Pose3: rotation-vector coordinates as circular).
    -> (dim, N).  K = 1 returns the density's own points, as AMP does."""
    import torch
    ctx = ctx or default_context()
    P = _d(proposals)
    K, d, N = P.shape
    if circular_mask is None:
        circular_mask = 0b100 if d == 3 else 0
    bw = kde_bandwidth(P, 0b111000 if d == 6 else circular_mask, ctx=ctx) if bandwidths is None else _d(bandwidths, (K, d))
    if d == 6:
        circular_mask = 0     # rotations live in the chart of each density: no wrapped coordinate
    o = opts if opts is not None else make_opts(N=N)
    dev = torch.device("cuda", ctx.device if hasattr(ctx, "device") else 0)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    tp, tb = t(P, torch.float64), t(bw, torch.float64)
    ptr, rows = t([0, K], torch.int32), t(np.arange(K), torch.int32)
    bin_, out = torch.zeros((1, d, N), dtype=torch.float64, device=dev), torch.empty((1, d, N), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.load().rome_product_gibbs_dev(ctx.handle, C.byref(o), d, 1, ptr.data_ptr(), rows.data_ptr(), tp.data_ptr(), tb.data_ptr(), K,
                                                  bin_.data_ptr(), out.data_ptr(), int(circular_mask), int(Niter), max(1, K)), ctx.handle)
    torch.cuda.synchronize(dev)
    return out[0].cpu().numpy()


def kde_max(bel, bw=None, grid_points=0, ctx=None):
    """bel (V, dim, N) host array -> (V, dim) max-density coordinates as IIF's getKDEMax computes them (PPE `max`), via rome_kde_max.
    bw (V, dim): kernel bandwidths; None selects them with kde_bandwidth (what `manikde!` would have stored)."""
    ctx = ctx or default_context()
    bel = _d(bel)
    V, d, N = bel.shape
    bw = kde_bandwidth(bel, ctx=ctx) if bw is None else _d(bw, (V, d))
    out = np.empty((V, d))
    _lib.check(_lib.load().rome_kde_max(ctx.handle, d, V, N, _p(bel), _p(bw), int(grid_points), _p(out)), ctx.handle)
    return out


def calcPPE(bel, bw=None, ctx=None):
    """Point estimates of V beliefs as the reference's calcVariablePPE stores them: dict(mean, max, suggested), each (V, dim).
    `suggested` follows the solved graph the reference ships (examples/fg-after-solve.tar.gz): mean translation, max-density heading for
    Pose2; the mean otherwise."""
    bel = _d(bel)
    mean, _ = belief_stats(bel, ctx)
    mx = kde_max(bel, bw, ctx=ctx)
    sug = mean.copy()
    if bel.shape[1] == 3:
        sug[:, 2] = mx[:, 2]
    return {"mean": mean, "max": mx, "suggested": sug}


# ------------------------------------------------------------------ helpers
def cholesky_lower(cov):
    """n covariances (n,d,d) or one (d,d) -> packed lower factors (n, d(d+1)/2)."""
    cov = _d(cov)
    single = cov.ndim == 2
    if single:
        cov = cov[None]
    n, d, _ = cov.shape
    L = np.empty((n, d * (d + 1) // 2))
    _lib.check(_lib.load().rome_cholesky_lower(d, n, _p(cov), _p(L)))
    return L[0] if single else L


def make_opts(N=100, solver=_lib.SOLVER_NEWTON, max_iters=None, inflate_cycles=None, tol=None, inflation=None,
              seed=None, stream_offset=None, layout=None, spread_nh=None, nullhypo=None, presampled=None):
    """presampled=1 (NOISE_MEASUREMENTS): the `noise` argument of the conv_* calls holds the measurement samples themselves
    (any SamplableBelief sampled by the caller) instead of standard normals."""
    return _lib.default_opts(solver, n_particles=N, max_iters=max_iters, inflate_cycles=inflate_cycles, tol=tol,
                             inflation=inflation, seed=seed, stream_offset=stream_offset, layout=layout,
                             spread_nh=spread_nh, nullhypo=nullhypo, presampled=presampled)


def _blocks(a, C_, N, d, layout, points_ok=True):
    if layout == _lib.LAYOUT_AOS_POINTS and points_ok:
        d = {3: 6, 6: 12}.get(d, d)
    shape = (C_, d, N) if layout == _lib.LAYOUT_SOA else (C_, N, d)
    return _d(a, shape)


def points_to_coords(dim, pts, ctx=None):
    ctx = ctx or default_context()
    pts = np.atleast_2d(_d(pts)); n = pts.shape[0]
    out = np.empty((n, dim))
    _lib.check(_lib.load().rome_points_to_coords(ctx.handle, dim, n, _p(pts), _p(out)), ctx.handle)
    return out


def coords_to_points(dim, coords, ctx=None):
    ctx = ctx or default_context()
    coords = np.atleast_2d(_d(coords)); n = coords.shape[0]
    out = np.empty((n, {3: 6, 6: 12}.get(dim, dim)))
    _lib.check(_lib.load().rome_coords_to_points(ctx.handle, dim, n, _p(coords), _p(out)), ctx.handle)
    return out


# ------------------------------------------------------------------ host-pointer convolutions
def conv_pose2pose2(opts, mu, cov, fixed, target, dirs=None, noise=None, want_status=False, ctx=None, alt=None, hypo_w=None):
    """alt / hypo_w: multihypo over two candidates for the factor's second pose (blocks of the other candidate, P(primary));
    `dirs` must then be one direction (0 or 1) for the whole call."""
    ctx = ctx or default_context()
    mu = np.atleast_2d(_d(mu)); C_ = mu.shape[0]; N = opts.n_particles
    cov = _d(cov, (C_, 3, 3))
    fixed = _blocks(fixed, C_, N, 3, opts.layout)
    out = _blocks(target, C_, N, 3, opts.layout).copy()
    noise = None if noise is None else _blocks(noise, C_, N, 3, opts.layout, points_ok=False)
    st = np.zeros((C_, N), dtype=np.int32) if want_status else None
    if alt is not None:
        d = np.unique(np.asarray(0 if dirs is None else dirs, dtype=np.int32))
        if d.size != 1 or int(d[0]) not in (0, 1):
            raise ValueError("conv_pose2pose2 with multihypo: one direction (0 or 1) per call")
        alt = _blocks(alt, C_, N, 3, opts.layout); hypo_w = _d(hypo_w, (C_,))
        _lib.check(_lib.load().rome_conv_pose2pose2_mh(ctx.handle, C.byref(opts), C_, int(d[0]), _p(mu), _p(cov), _p(fixed), _p(alt),
                                                      _p(hypo_w), _p(noise), _p(out), _pi(st)), ctx.handle)
        return (out, st) if want_status else out
    dirs = None if dirs is None else np.ascontiguousarray(dirs, dtype=np.int32)
    _lib.check(_lib.load().rome_conv_pose2pose2(ctx.handle, C.byref(opts), C_, _pi(dirs), _p(mu), _p(cov), _p(fixed),
                                               _p(noise), _p(out), _pi(st)), ctx.handle)
    return (out, st) if want_status else out


def conv_pose2point2br(opts, direction, mu, sigma, fixed, target, noise=None, want_status=False, ctx=None,
                       alt=None, hypo_w=None):
    """alt / hypo_w: multihypo over two landmark candidates (blocks of the other landmark, P(primary))."""
    ctx = ctx or default_context()
    mu = np.atleast_2d(_d(mu)); C_ = mu.shape[0]; N = opts.n_particles
    sigma = _d(sigma, (C_, 2))
    df, dt = (3, 2) if direction == 0 else (2, 3)
    fixed = _blocks(fixed, C_, N, df, opts.layout)
    out = _blocks(target, C_, N, dt, opts.layout).copy()
    noise = None if noise is None else _blocks(noise, C_, N, 2, opts.layout, points_ok=False)
    st = np.zeros((C_, N), dtype=np.int32) if want_status else None
    if alt is None:
        _lib.check(_lib.load().rome_conv_pose2point2br(ctx.handle, C.byref(opts), C_, int(direction), _p(mu), _p(sigma),
                                                      _p(fixed), _p(noise), _p(out), _pi(st)), ctx.handle)
    else:
        alt = _blocks(alt, C_, N, 2, opts.layout); hypo_w = _d(hypo_w, (C_,))
        _lib.check(_lib.load().rome_conv_pose2point2br_mh(ctx.handle, C.byref(opts), C_, int(direction), _p(mu), _p(sigma),
                                                         _p(fixed), _p(alt), _p(hypo_w), _p(noise), _p(out), _pi(st)), ctx.handle)
    return (out, st) if want_status else out


def conv_pose3pose3(opts, mu, cov, fixed, target, dirs=None, noise=None, want_status=False, ctx=None):
    ctx = ctx or default_context()
    mu = np.atleast_2d(_d(mu)); C_ = mu.shape[0]; N = opts.n_particles
    cov = _d(cov, (C_, 6, 6))
    fixed = _blocks(fixed, C_, N, 6, opts.layout)
    out = _blocks(target, C_, N, 6, opts.layout).copy()
    noise = None if noise is None else _blocks(noise, C_, N, 6, opts.layout, points_ok=False)
    dirs = None if dirs is None else np.ascontiguousarray(dirs, dtype=np.int32)
    st = np.zeros((C_, N), dtype=np.int32) if want_status else None
    _lib.check(_lib.load().rome_conv_pose3pose3(ctx.handle, C.byref(opts), C_, _pi(dirs), _p(mu), _p(cov), _p(fixed),
                                               _p(noise), _p(out), _pi(st)), ctx.handle)
    return (out, st) if want_status else out


def _sample_prior(fn, d, opts, mu, cov, noise, ctx):
    ctx = ctx or default_context()
    mu = np.atleast_2d(_d(mu)); C_ = mu.shape[0]; N = opts.n_particles
    cov = _d(cov, (C_, d, d))
    noise = None if noise is None else _blocks(noise, C_, N, d, opts.layout, points_ok=False)
    pl = {2: 2, 3: 6, 6: 12}[d] if opts.layout == _lib.LAYOUT_AOS_POINTS else d
    out = np.empty((C_, d, N) if opts.layout == _lib.LAYOUT_SOA else (C_, N, pl))
    _lib.check(fn(ctx.handle, C.byref(opts), C_, _p(mu), _p(cov), _p(noise), _p(out)), ctx.handle)
    return out


def sample_priorpose2(opts, mu, cov, noise=None, ctx=None):
    return _sample_prior(_lib.load().rome_sample_priorpose2, 3, opts, mu, cov, noise, ctx)


def sample_priorpose3(opts, mu, cov, noise=None, ctx=None):
    return _sample_prior(_lib.load().rome_sample_priorpose3, 6, opts, mu, cov, noise, ctx)


def sample_priorpoint2(opts, mu, cov, noise=None, ctx=None):
    """N samples of `PriorPoint2(MvNormal(mu, cov))` (src/factors/Point2D.jl:8-18): the proposal a landmark prior contributes"""
    return _sample_prior(_lib.load().rome_sample_priorpoint2, 2, opts, mu, cov, noise, ctx)
