"""ctypes binding of oracle/librome_oracle.so (TEST INFRASTRUCTURE ONLY).

Arrays are float64 C-contiguous.  Belief/proposal blocks are SoA: [block][dim][N].
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librome_oracle.so")

SOLVER_CLOSED_FORM, SOLVER_NEWTON, SOLVER_NELDER_MEAD = 0, 1, 2
SOLVER_GAUSS_NEWTON = 1   # the oracle's Newton mode IS the Gauss-Newton iteration on the residual functor (p2p2_newton, br_newton, p3p3_newton_pt)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("rome_oracle.c", "rome_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librome_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class Opts(C.Structure):
    _fields_ = [("n_particles", C.c_int32), ("solver", C.c_int32), ("max_iters", C.c_int32),
                ("inflate_cycles", C.c_int32), ("tol", C.c_double), ("inflation", C.c_double),
                ("seed", C.c_uint64), ("stream_offset", C.c_uint64), ("nullhypo", C.c_double), ("spread_nh", C.c_double)]


def make_opts(N=100, solver=SOLVER_NEWTON, max_iters=None, inflate_cycles=3, tol=None, inflation=5.0,
              seed=0x524F4D45, stream_offset=0, nullhypo=0.0, spread_nh=3.0):
    if solver == 3:   # the device library's ROME_SOLVER_GAUSS_NEWTON: the oracle's Newton mode IS that iteration on the functor
        solver = SOLVER_NEWTON
    if solver not in (SOLVER_CLOSED_FORM, SOLVER_NEWTON, SOLVER_NELDER_MEAD):
        raise ValueError("oracle: unknown solver %r" % (solver,))
    if max_iters is None:
        max_iters = 1000 if solver == SOLVER_NELDER_MEAD else 20
    if tol is None:
        tol = 1e-8 if solver == SOLVER_NELDER_MEAD else 1e-12
    return Opts(N, solver, max_iters, inflate_cycles, tol, inflation, seed, stream_offset, nullhypo, spread_nh)


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("ROME_ORACLE_SO")   # cpu_bench.py: the -O3 -march=native build made on the timing box
        if not so:
            build()
            so = _SO
        _lib = C.CDLL(so)
        _lib.ro_sym_rem.restype = C.c_double
        _lib.ro_sym_rem.argtypes = [C.c_double]
        _lib.ro_num_threads.restype = C.c_int
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().ro_philox4x32_10(c, k, o)
    return list(o)


def rng_normals(seed, stream, particle, d):
    out = np.zeros(d + 1)
    lib().ro_rng_normals(C.c_uint64(seed), C.c_uint64(stream), C.c_uint32(particle), d, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:d]


def box_muller(wa, wb):
    a, b = C.c_double(), C.c_double()
    lib().ro_box_muller(C.c_uint32(wa), C.c_uint32(wb), C.byref(a), C.byref(b))
    return a.value, b.value


def rng_entropy(seed, stream, particle, cycle, d):
    out = np.zeros(d + 3)
    lib().ro_rng_entropy(C.c_uint64(seed), C.c_uint64(stream), C.c_uint32(particle), cycle, d, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:d]


def sym_rem(x):
    return lib().ro_sym_rem(float(x))


def _conv1(fn, a, nout):
    a, pa = _d(a); out = np.zeros(nout); fn(pa, out.ctypes.data_as(C.POINTER(C.c_double))); return out


def pose2_point(c): return _conv1(lib().ro_pose2_point_from_coords, c, 6)
def pose2_coords(p): return _conv1(lib().ro_pose2_coords_from_point, p, 3)
def pose3_point(c): return _conv1(lib().ro_pose3_point_from_coords, c, 12)
def pose3_coords(p): return _conv1(lib().ro_pose3_coords_from_point, p, 6)
def so3_exp(w): return _conv1(lib().ro_so3_exp, w, 9)
def so3_log(R): return _conv1(lib().ro_so3_log, R, 3)


def rotxyz(r, p, y):
    out = np.zeros(9)
    lib().ro_rotxyz(C.c_double(r), C.c_double(p), C.c_double(y), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def _res_pt(fn, nout, *args):
    ptrs = []; keep = []
    for a in args:
        a, p = _d(a); keep.append(a); ptrs.append(p)
    out = np.zeros(nout)
    fn(*ptrs, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def residual_pose2pose2_pt(X, p, q): return _res_pt(lib().ro_residual_pose2pose2_pt, 3, X, p, q)
def residual_priorpose2_pt(m, p): return _res_pt(lib().ro_residual_priorpose2_pt, 3, m, p)
def residual_pose2point2br_pt(meas, p, l): return _res_pt(lib().ro_residual_pose2point2br_pt, 2, meas, p, l)
def residual_pose3pose3_pt(X, p, q): return _res_pt(lib().ro_residual_pose3pose3_pt, 6, X, p, q)
def residual_priorpose3_pt(m, p): return _res_pt(lib().ro_residual_priorpose3_pt, 6, m, p)


def _res_batch(fn, dout, *args):
    keep = [_d(np.atleast_2d(a)) for a in args]
    n = keep[0][0].shape[0]
    out = np.zeros((n, dout))
    fn(n, *[k[1] for k in keep], out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def residual_pose2pose2(z, p, q): return _res_batch(lib().ro_residual_pose2pose2, 3, z, p, q)
def residual_priorpose2(m, p): return _res_batch(lib().ro_residual_priorpose2, 3, m, p)
def residual_pose2point2br(z, p, l): return _res_batch(lib().ro_residual_pose2point2br, 2, z, p, l)
def residual_pose3pose3(z, p, q): return _res_batch(lib().ro_residual_pose3pose3, 6, z, p, q)
def residual_priorpose3(m, p): return _res_batch(lib().ro_residual_priorpose3, 6, m, p)


def cholesky_lower(cov):
    cov = np.asarray(cov, dtype=np.float64)
    d = cov.shape[0]
    a, pa = _d(cov); out = np.zeros(d * (d + 1) // 2)
    rc = lib().ro_cholesky_lower(d, pa, out.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise ValueError("covariance not positive definite")
    return out


def belief_spread(blk):
    """blk: [d][N] with d in {2 (Point2), 3 (Pose2), 6 (Pose3)} -> (mean[d], std[d])"""
    blk, p = _d(blk)
    d, N = blk.shape
    mean = np.zeros(d); std = np.zeros(d)
    pm = mean.ctypes.data_as(C.POINTER(C.c_double)); ps = std.ctypes.data_as(C.POINTER(C.c_double))
    if d == 3:
        r = [blk[k].ctypes.data_as(C.POINTER(C.c_double)) for k in range(3)]
        lib().ro_belief_spread_se2(N, r[0], r[1], r[2], pm, ps)
    elif d == 2:
        r = [blk[k].ctypes.data_as(C.POINTER(C.c_double)) for k in range(2)]
        lib().ro_belief_spread_r2(N, r[0], r[1], pm, ps)
    elif d == 6:
        lib().ro_belief_spread_se3(N, p, pm, ps)
    else:
        raise ValueError(d)
    return mean, std


def nelder_mead(f, x0, max_iters=1000, g_tol=1e-8):
    x = np.array(x0, dtype=np.float64)
    n = x.size
    CB = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_void_p)
    cb = CB(lambda px, _ctx: float(f(np.ctypeslib.as_array(px, shape=(n,)).copy())))
    ne = C.c_int(0)
    rc = lib().ro_nelder_mead(n, cb, None, x.ctypes.data_as(C.POINTER(C.c_double)), max_iters, C.c_double(g_tol), C.byref(ne))
    return x, rc, ne.value


def conv_pose2pose2(opts, mu, L, bel, fixed_var, target_var, dirs, factor=None, noise=None, want_status=False,
                    alt_var=None, hypo_w=None, spread_nh=3.0):
    mu, pmu = _d(mu); L, pL = _d(L); bel, pb = _d(bel)
    fv, pfv = _i(fixed_var); tv, ptv = _i(target_var); dr, pdr = _i(dirs); fa, pfa = _i(factor)
    Cn = len(fv); N = opts.n_particles
    nz, pn = (None, None) if noise is None else _d(noise)
    out = np.zeros((Cn, 3, N)); st = np.zeros((Cn, N), dtype=np.int32)
    av, pav = _i(alt_var)
    hw, phw = (None, None) if hypo_w is None else _d(hypo_w)
    rc = lib().ro_conv_pose2pose2_mh(C.byref(opts), Cn, pfa, pdr, pfv, ptv, pmu, pL, pb, pn,
                                     out.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_int32)),
                                     pav, phw, C.c_double(spread_nh))
    assert rc == 0
    return (out, st) if want_status else out


def conv_pose2point2br(opts, direction, mu, sigma, bel_fixed, bel_target, fixed_var, target_var, factor=None,
                       noise=None, want_status=False, alt_var=None, hypo_w=None, spread_nh=3.0):
    mu, pmu = _d(mu); sg, psg = _d(sigma); bf, pbf = _d(bel_fixed); bt, pbt = _d(bel_target)
    fv, pfv = _i(fixed_var); tv, ptv = _i(target_var); fa, pfa = _i(factor)
    Cn = len(fv); N = opts.n_particles; dt = 2 if direction == 0 else 3
    nz, pn = (None, None) if noise is None else _d(noise)
    out = np.zeros((Cn, dt, N)); st = np.zeros((Cn, N), dtype=np.int32)
    av, pav = _i(alt_var)
    hw, phw = (None, None) if hypo_w is None else _d(hypo_w)
    rc = lib().ro_conv_pose2point2br_mh(C.byref(opts), Cn, pfa, int(direction), pfv, ptv, pmu, psg, pbf, pbt, pn,
                                        out.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_int32)),
                                        pav, phw, C.c_double(spread_nh))
    assert rc == 0
    return (out, st) if want_status else out


def sample_priorpose2(opts, mu, L, factor=None, noise=None, C_=None):
    mu, pmu = _d(np.atleast_2d(mu)); L, pL = _d(np.atleast_2d(L)); fa, pfa = _i(factor)
    Cn = C_ if C_ is not None else (len(fa) if fa is not None else mu.shape[0])
    nz, pn = (None, None) if noise is None else _d(noise)
    out = np.zeros((Cn, 3, opts.n_particles))
    rc = lib().ro_sample_priorpose2(C.byref(opts), Cn, pfa, pmu, pL, pn, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


def sample_priorpoint2(opts, mu, L, factor=None, noise=None, C_=None):
    mu, pmu = _d(np.atleast_2d(mu)); L, pL = _d(np.atleast_2d(L)); fa, pfa = _i(factor)
    Cn = C_ if C_ is not None else (len(fa) if fa is not None else mu.shape[0])
    nz, pn = (None, None) if noise is None else _d(noise)
    out = np.zeros((Cn, 2, opts.n_particles))
    rc = lib().ro_sample_priorpoint2(C.byref(opts), Cn, pfa, pmu, pL, pn, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


def conv_pose3pose3(opts, mu, L, bel, fixed_var, target_var, dirs, factor=None, noise=None, want_status=False):
    mu, pmu = _d(mu); L, pL = _d(L); bel, pb = _d(bel)
    fv, pfv = _i(fixed_var); tv, ptv = _i(target_var); dr, pdr = _i(dirs); fa, pfa = _i(factor)
    Cn = len(fv); N = opts.n_particles
    nz, pn = (None, None) if noise is None else _d(noise)
    out = np.zeros((Cn, 6, N)); st = np.zeros((Cn, N), dtype=np.int32)
    rc = lib().ro_conv_pose3pose3(C.byref(opts), Cn, pfa, pdr, pfv, ptv, pmu, pL, pb, pn,
                                  out.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0
    return (out, st) if want_status else out


def sample_priorpose3(opts, mu, L, factor=None, noise=None, C_=None):
    mu, pmu = _d(np.atleast_2d(mu)); L, pL = _d(np.atleast_2d(L)); fa, pfa = _i(factor)
    Cn = C_ if C_ is not None else (len(fa) if fa is not None else mu.shape[0])
    nz, pn = (None, None) if noise is None else _d(noise)
    out = np.zeros((Cn, 6, opts.n_particles))
    rc = lib().ro_sample_priorpose3(C.byref(opts), Cn, pfa, pmu, pL, pn, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


def product(opts, dim, prop_ptr, prop_rows, prop, bel_in, prop_bw=None):
    pp, ppp = _i(prop_ptr); pr, ppr = _i(prop_rows); P, pP = _d(prop); B, pB = _d(bel_in)
    bw, pbw = (None, None) if prop_bw is None else _d(prop_bw)
    V = len(pp) - 1
    out = np.zeros_like(B)
    rc = lib().ro_product_bw(C.byref(opts), int(dim), V, ppp, ppr, pP, pbw, pB, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


def product_msgibbs(opts, dim, prop_ptr, prop_rows, prop, prop_bw, bel_in, circular_mask=0, gibbs_iters=1):
    """⚠AMP manifoldProduct restated (multiscale Gibbs product of the proposal KDEs); prop_bw (rows, dim) required."""
    pp, ppp = _i(prop_ptr); pr, ppr = _i(prop_rows); P, pP = _d(prop); B, pB = _d(bel_in); bw, pbw = _d(prop_bw)
    V = len(pp) - 1
    out = np.zeros_like(B)
    fn = lib().ro_product_msgibbs
    fn.argtypes = [C.POINTER(Opts), C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double),
                   C.POINTER(C.c_double), C.c_uint32, C.c_int, C.POINTER(C.c_double)]
    rc = fn(C.byref(opts), int(dim), V, ppp, ppr, pP, pbw, pB, int(circular_mask), int(gibbs_iters), out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


def kde_bandwidths(bel, circular_mask, tol_euclid=1e-2, tol_circular=1e-6):
    """bel (V, dim, N) -> (V, dim) leave-one-out likelihood bandwidths (ro_kde_bandwidths)."""
    B, pB = _d(bel)
    V, dim, N = B.shape
    out = np.zeros((V, dim))
    lib().ro_kde_bandwidths.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_uint32, C.c_double, C.c_double,
                                        C.POINTER(C.c_double)]
    rc = lib().ro_kde_bandwidths(dim, V, N, pB, int(circular_mask), float(tol_euclid), float(tol_circular),
                                 out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


def kde_max(bel, bw, G=200, extend=0.1):
    """bel (V, dim, N), bw (V, dim) -> (V, dim) max-density coordinates (ro_kde_max)."""
    B, pB = _d(bel); H, pH = _d(bw)
    V, dim, N = B.shape
    out = np.zeros((V, dim))
    lib().ro_kde_max.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_double,
                                 C.POINTER(C.c_double)]
    assert lib().ro_kde_max(dim, V, N, pB, pH, int(G), float(extend), out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    return out


def num_threads():
    return lib().ro_num_threads()


def set_num_threads(n):
    lib().ro_set_num_threads(int(n))
