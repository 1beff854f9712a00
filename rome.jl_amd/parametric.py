"""`solveGraphParametric!` mirror (IIF API as used by the reference: test/testParametric.jl:41,
test/testPose3.jl:46, src/services/AdditionalUtils.jl:22): Levenberg-Marquardt on
    Σ_f ‖ Σ_f^{-1/2} r_f(μ_f ; x) ‖²
over all variables.  The batched residuals + Jacobians come from the device (`rome_linearize`); the
sparse normal-equation assembly/solve and the manifold retraction are host logic (scipy.sparse)."""
import numpy as np

from . import _lib, api
from .factors import (Pose2, Point2, Pose3, Pose2Pose2, PriorPose2, Pose2Point2BearingRange, Pose3Pose3, PriorPose3,
                      PriorPoint2, getMeasurementParametric)

_KIND = {PriorPose2: _lib.FACTOR_PRIORPOSE2, Pose2Pose2: _lib.FACTOR_POSE2POSE2, Pose2Point2BearingRange: _lib.FACTOR_POSE2POINT2BR,
         PriorPoint2: _lib.FACTOR_PRIORPOINT2, Pose3Pose3: _lib.FACTOR_POSE3POSE3, PriorPose3: _lib.FACTOR_PRIORPOSE3}


def _whitening(info):
    """W with WᵀW = Σ⁻¹ (upper Cholesky factor of the information matrix)."""
    return np.linalg.cholesky(info).T


def _retract(vt, x, d):
    if vt is Pose3:
        from scipy.spatial.transform import Rotation as Rot
        R = Rot.from_rotvec(x[3:]) * Rot.from_rotvec(d[3:])
        return np.concatenate([x[:3] + d[:3], R.as_rotvec()])
    y = x + d
    if vt is Pose2:
        y[2] = np.arctan2(np.sin(y[2]), np.cos(y[2]))
    return y


def _se2_compose(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([a[0] + c * b[0] - s * b[1], a[1] + s * b[0] + c * b[1], a[2] + b[2]])


def _se3_compose(a, b):
    from scipy.spatial.transform import Rotation as Rot
    Ra = Rot.from_rotvec(a[3:])
    return np.concatenate([a[:3] + Ra.apply(b[:3]), (Ra * Rot.from_rotvec(b[3:])).as_rotvec()])


def initParametric(fg, refine=True):
    """Deterministic initial values: prior means, then measurement means composed along relative factors
    (stand-in for initAll! + initParametricFrom!, test/testParametric.jl:32-33); pure Pose2 pose graphs are then refined by two
    linear solves (`_pose2_two_stage_init`) unless refine=False."""
    x = {}
    for _, labels, f in fg.factors:
        if isinstance(f, (PriorPose2, PriorPose3, PriorPoint2)):
            x[labels[0]] = np.array(f.Z.mu, dtype=float)
    progress = True
    while progress:
        progress = False
        for _, labels, f in fg.factors:
            if isinstance(f, Pose2Pose2) and labels[0] in x and labels[1] not in x:
                x[labels[1]] = _se2_compose(x[labels[0]], f.Z.mu); progress = True
            elif isinstance(f, Pose3Pose3) and labels[0] in x and labels[1] not in x:
                x[labels[1]] = _se3_compose(x[labels[0]], f.Z.mu); progress = True
            elif isinstance(f, Pose2Point2BearingRange):
                p, l = labels
                if p in x and l not in x:
                    a = x[p][2] + f.bearing.mu
                    x[l] = x[p][:2] + f.range.mu * np.array([np.cos(a), np.sin(a)]); progress = True
    # poses only seen through bearing-range sightings of known landmarks: multi-start over the heading,
    # position from the first sighting, candidate cost from the device residual entry point
    for lbl, vt in fg.variables.items():
        if lbl in x or vt is not Pose2:
            continue
        sight = [(f, ls[1]) for _, ls, f in fg.factors if isinstance(f, Pose2Point2BearingRange) and ls[0] == lbl and ls[1] in x]
        if not sight:
            continue
        f0, l0 = sight[0]
        cands = []
        for th in np.linspace(-np.pi, np.pi, 16, endpoint=False):
            a = th + f0.bearing.mu
            cands.append(np.array([x[l0][0] - f0.range.mu * np.cos(a), x[l0][1] - f0.range.mu * np.sin(a), th]))
        cands = np.array(cands)
        cost = np.zeros(len(cands))
        for f, l in sight:
            z = np.tile([f.bearing.mu, f.range.mu], (len(cands), 1))
            r = api.residual_pose2point2br(z, cands, np.tile(x[l], (len(cands), 1)))
            cost += (r[:, 0] / f.bearing.sigma) ** 2 + (r[:, 1] / f.range.sigma) ** 2
        x[lbl] = cands[int(np.argmin(cost))]
    for l, vt in fg.variables.items():
        if l not in x:
            x[l] = np.zeros(vt.dim)
    return _pose2_two_stage_init(fg, x) if refine else x


def _pose2_two_stage_init(fg, x):
    """Refine dead-reckoned Pose2 values of a pure Pose2Pose2 pose graph with two LINEAR least-squares solves: all headings from
    the relative-heading measurements (the 2π ambiguity of every edge is fixed by rounding against the dead-reckoned headings),
    then all translations with those headings held fixed (the residual of src/factors/Pose2D.jl:51-67 is linear in the
    translations once R(θ_p) is known).  On Manhattan-3500 this brings Levenberg-Marquardt from ~17 iterations to a handful.
    Returns x unchanged for graphs it does not apply to (other variable / factor types, no prior)."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    if any(vt is not Pose2 for vt in fg.variables.values()):
        return x
    rel = [(ls, f) for _, ls, f in fg.factors if isinstance(f, Pose2Pose2)]
    pri = [(ls, f) for _, ls, f in fg.factors if isinstance(f, PriorPose2)]
    if len(rel) + len(pri) != len(fg.factors) or not pri or len(rel) < 2:
        return x
    labels = list(fg.variables); idx = {l: k for k, l in enumerate(labels)}
    n, F = len(labels), len(rel)
    i = np.array([idx[ls[0]] for ls, _ in rel]); j = np.array([idx[ls[1]] for ls, _ in rel])
    mu = np.array([f.Z.mu for _, f in rel]); info = np.linalg.inv(np.array([f.Z.cov for _, f in rel]))   # (one batched call)
    th0 = np.array([x[l][2] for l in labels])
    # ---- headings: θ_j − θ_i = z_θ + 2π k_ij
    k = np.round((th0[j] - th0[i] - mu[:, 2]) / (2 * np.pi))
    w = np.sqrt(info[:, 2, 2])
    rows = np.r_[np.arange(F), np.arange(F)]; cols = np.r_[j, i]; vals = np.r_[w, -w]
    rhs = list(w * (mu[:, 2] + 2 * np.pi * k))
    for r_, (ls, f) in enumerate(pri):
        wp = 1.0 / np.sqrt(f.Z.cov[2, 2])
        rows = np.r_[rows, F + r_]; cols = np.r_[cols, idx[ls[0]]]; vals = np.r_[vals, wp]
        rhs.append(wp * (f.Z.mu[2] + 2 * np.pi * np.round((th0[idx[ls[0]]] - f.Z.mu[2]) / (2 * np.pi))))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(F + len(pri), n))
    with _blas_single_thread():
        th = spsolve((A.T @ A).tocsc(), A.T @ np.array(rhs))
    # ---- translations with R(θ_i) fixed: t_j − t_i = R(θ_i) z_t, whitened with the translation block of the information
    c, s_ = np.cos(th[i]), np.sin(th[i])
    d = np.stack([c * mu[:, 0] - s_ * mu[:, 1], s_ * mu[:, 0] + c * mu[:, 1]], 1)
    Wt = np.linalg.cholesky(info[:, :2, :2]).transpose(0, 2, 1)        # Wᵀ W = Λ_tt
    rows, cols, vals = [], [], []
    for a_ in range(2):
        for b_ in range(2):
            rows += [2 * np.arange(F) + a_, 2 * np.arange(F) + a_]
            cols += [2 * j + b_, 2 * i + b_]
            vals += [Wt[:, a_, b_], -Wt[:, a_, b_]]
    rhs = list(np.einsum("fab,fb->fa", Wt, d).ravel())
    m = 2 * F
    for ls, f in pri:
        Wp = np.linalg.cholesky(np.linalg.inv(f.Z.cov[:2, :2])).T
        for a_ in range(2):
            for b_ in range(2):
                rows.append(np.array([m + a_])); cols.append(np.array([2 * idx[ls[0]] + b_])); vals.append(np.array([Wp[a_, b_]]))
        rhs += list(Wp @ f.Z.mu[:2]); m += 2
    B = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, 2 * n))
    with _blas_single_thread():
        t = spsolve((B.T @ B).tocsc(), B.T @ np.array(rhs)).reshape(n, 2)
    return {l: np.array([t[k_, 0], t[k_, 1], np.arctan2(np.sin(th[k_]), np.cos(th[k_]))]) for k_, l in enumerate(labels)}


class _NoLimit:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_BLAS_CONTROLLER = []


def _blas_single_thread():
    """context manager: the BLAS pools of this process limited to one thread (threadpoolctl when importable, else a no-op).  The
    controller is built ONCE: constructing it walks the loaded shared objects (tens of ms), limiting through it is microseconds."""
    if not _BLAS_CONTROLLER:
        try:
            from threadpoolctl import ThreadpoolController
            _BLAS_CONTROLLER.append(ThreadpoolController())
        except Exception:   # noqa: BLE001
            _BLAS_CONTROLLER.append(None)
    c = _BLAS_CONTROLLER[0]
    return _NoLimit() if c is None else c.limit(limits=1, user_api="blas")


class _Problem:
    """Flat state vector + precomputed gather indices and sparsity pattern (all per-iteration work is
    vectorised numpy around one `rome_linearize` call per factor kind)."""

    def __init__(self, fg):
        self.fg = fg
        self.labels = list(fg.variables)
        self.vt = [fg.variables[l] for l in self.labels]
        self.off = np.concatenate([[0], np.cumsum([t.dim for t in self.vt])]).astype(np.int64)
        self.n = int(self.off[-1])
        self.index = {l: i for i, l in enumerate(self.labels)}
        groups = {}
        for _, labels, f in fg.factors:
            k = _KIND.get(type(f))
            if k is None:
                raise TypeError("factor %s is outside the hot path" % type(f).__name__)
            g = groups.setdefault(k, dict(mu=[], W=[], a=[], b=[], cov=[]))
            if hasattr(f, "Z"):      # MvNormal factors: μ now, the whitening of the whole group in one batched call below
                g["mu"].append(f.Z.mu); g["cov"].append(f.Z.cov)
            else:
                mu, info = getMeasurementParametric(f)
                g["mu"].append(mu); g["W"].append(_whitening(info))
            g["a"].append(self.index[labels[0]])
            g["b"].append(self.index[labels[1]] if len(labels) > 1 else -1)
        for g in groups.values():
            if g["cov"]:             # W with WᵀW = Σ⁻¹ for every factor of the group (same as getMeasurementParametric + _whitening)
                g["W"] = list(np.linalg.cholesky(np.linalg.inv(np.asarray(g["cov"]))).transpose(0, 2, 1))
            del g["cov"]
        self.groups = {}
        rows, cols = [], []
        m = 0
        for k, g in groups.items():
            g = {n_: np.asarray(v) for n_, v in g.items()}
            dz, dr, da, db = api._LIN_DIMS[k]
            F = len(g["a"])
            g["ia"] = self.off[g["a"]][:, None] + np.arange(da)[None, :]
            g["ib"] = self.off[g["b"]][:, None] + np.arange(db)[None, :] if db else None
            ridx = m + np.arange(F * dr).reshape(F, dr)
            g["rslice"] = slice(m, m + F * dr)
            for idx, dv in ((g["ia"], da), (g["ib"], db)):
                if idx is None:
                    continue
                rows.append(np.repeat(ridx[:, :, None], dv, axis=2).ravel())
                cols.append(np.repeat(idx[:, None, :], dr, axis=1).ravel())
            m += F * dr
            self.groups[k] = g
        self.m = m
        self.rows = np.concatenate(rows); self.cols = np.concatenate(cols)
        self.perm = self._fill_reducing_order(groups)          # elimination order of the scalar unknowns
        self.pos = np.empty(self.n, dtype=np.int64); self.pos[self.perm] = np.arange(self.n)
        self.cols_p = self.pos[self.cols]                      # J is assembled directly in that order
        # The sparsity pattern of J is fixed over the iterations: its CSR structure (row pointers, sorted column indices) and the
        # position of every CSR slot in the concatenated value blocks are computed ONCE; a linearisation then only permutes values
        # (scipy's COO -> CSR conversion of the 8.7e5 entries of the 10k helix cost ~5 ms per linearisation: most of linearize_s).
        import scipy.sparse as sp
        nnz = len(self.rows)
        S = sp.csr_matrix((np.arange(1, nnz + 1, dtype=np.float64), (self.rows, self.cols_p)), shape=(self.m, self.n))
        S.sort_indices()
        if S.nnz != nnz:
            raise AssertionError("duplicate (row, column) entries in the Jacobian pattern")
        self.csr_perm = (S.data - 1.0).astype(np.int64)
        self.csr_indices, self.csr_indptr = S.indices.copy(), S.indptr.copy()
        # retraction index sets
        self.pose2_th = np.array([self.off[i] + 2 for i, t in enumerate(self.vt) if t is Pose2], dtype=np.int64)
        self.pose3_w = np.array([self.off[i] + 3 for i, t in enumerate(self.vt) if t is Pose3], dtype=np.int64)

    def _fill_reducing_order(self, groups):
        """Minimum-degree ordering of the VARIABLE graph (one node per variable, not per scalar), expanded to the scalar
        unknowns: the pattern of JᵀJ is fixed over the LM iterations, so it is computed once.  On the Manhattan-shaped
        graph the factor of the permuted system has half the non-zeros of SuperLU's default COLAMD ordering."""
        import scipy.sparse as sp
        from scipy.sparse.linalg import splu
        V = len(self.labels)
        ii, jj = [], []
        for g in groups.values():
            a, b = np.asarray(g["a"]), np.asarray(g["b"])
            k = b >= 0
            ii.append(a[k]); jj.append(b[k])
        ii = np.concatenate(ii) if ii else np.zeros(0, np.int64); jj = np.concatenate(jj) if jj else np.zeros(0, np.int64)
        A = sp.csc_matrix((np.ones(2 * len(ii)), (np.r_[ii, jj], np.r_[jj, ii])), shape=(V, V)) + (V + 1.0) * sp.identity(V, format="csc")
        with _blas_single_thread():
            pc = splu(A.tocsc(), permc_spec="MMD_AT_PLUS_A", options=dict(SymmetricMode=True, DiagPivotThresh=0.0)).perm_c
        order = np.argsort(pc)            # perm_c[i] = position of column i  ->  variables in elimination order
        return np.concatenate([np.arange(self.off[v], self.off[v + 1]) for v in order]).astype(np.int64)

    def solve_spd(self, Hp, rhs_p):
        """x_p with Hp x_p = rhs_p for the symmetric positive definite damped normal matrix; Hp is already in the
        precomputed elimination order (columns of J are assembled in it), so SuperLU runs with NATURAL ordering."""
        from scipy.sparse.linalg import splu
        # SuperLU hands its supernodes to BLAS in calls of a few hundred flops each; a multi-threaded OpenBLAS (64 threads by default on the
        # GPU box, behind a cgroup quota of 16 CPUs) turns every one of them into a thread-pool round trip: the factorisation of the 10k
        # helix takes 0.134 s with the default pool and 0.05 s on ONE BLAS thread (bench.py parametric_helix10k: 6.1 -> 2.35 s).
        with _blas_single_thread():
            # (panel_size = 1, relax = 4: the supernodes of a pose graph's factor are 3 - 6 columns wide; SuperLU's default panels of 10 / 20
            #  columns and relaxed supernodes of 10 cost 49 ms per factorisation of the 10k helix against 35 ms -- scripts/splu_options.py)
            return splu(Hp.tocsc(), permc_spec="NATURAL", panel_size=1, relax=4, options=dict(SymmetricMode=True, DiagPivotThresh=0.0)).solve(rhs_p)

    def pack(self, xdict):
        X = np.zeros(self.n)
        for i, l in enumerate(self.labels):
            X[self.off[i]:self.off[i + 1]] = xdict[l]
        return X

    def unpack(self, X):
        return {l: X[self.off[i]:self.off[i + 1]].copy() for i, l in enumerate(self.labels)}

    def retract(self, X, d):
        Y = X + d
        if len(self.pose2_th):
            Y[self.pose2_th] = np.arctan2(np.sin(Y[self.pose2_th]), np.cos(Y[self.pose2_th]))
        if len(self.pose3_w):
            from scipy.spatial.transform import Rotation as Rot
            idx = self.pose3_w[:, None] + np.arange(3)[None, :]
            Y[idx] = (Rot.from_rotvec(X[idx]) * Rot.from_rotvec(d[idx])).as_rotvec()
        return Y

    def linearize(self, X, ctx=None, shard=None):
        """-> (r (m,), J scipy.sparse.csr (m, n)).  `shard` = rome_jl_amd.distributed.LinearizeShard: every rank evaluates a
        contiguous slice of the rows of each factor kind on its own GPU and one all-gather per kind rebuilds the full blocks on
        every rank (factor rows are independent; BASELINE configs[4] "batched Jacobians on 8 GPUs")."""
        import scipy.sparse as sp
        r = np.empty(self.m); vals = []; blocks = []
        indexed = shard is not None and hasattr(shard, "linearize_indexed")
        if indexed:
            shard.begin(X)          # X crosses PCIe once; the gathers X[ia] / X[ib] of every factor kind run on the device
        for k, g in self.groups.items():
            if indexed:
                rk, Ja, Jb = shard.linearize_indexed(k, g["mu"], g["W"], X, g["ia"], g["ib"], ctx)
            else:
                xa = X[g["ia"]]
                xb = X[g["ib"]] if g["ib"] is not None else None
                if shard is None:
                    rk, Ja, Jb = api.linearize(k, g["mu"], g["W"], xa, xb, ctx=ctx)
                else:
                    rk, Ja, Jb = shard.linearize(k, g["mu"], g["W"], xa, xb, ctx)
            r[g["rslice"]] = rk.ravel()
            vals.append(Ja.ravel())
            if Jb is not None:
                vals.append(Jb.ravel())
            blocks.append((Ja, Jb))
        J = sp.csr_matrix((np.concatenate(vals)[self.csr_perm], self.csr_indices, self.csr_indptr), shape=(self.m, self.n))   # columns in elimination order
        J.has_sorted_indices = True
        self.blocks = blocks          # the whitened Jacobian blocks of THIS linearisation (normal_matrix)
        return r, J

    # ---- the normal matrix H = J^T J from the factor blocks (round 6).  scipy's generic sparse product J.T @ J was 2/3 of the host time of
    # an LM iteration on the 10k helix (0.11 of 0.17 s); H is the sum over the factors of [Ja Jb]^T [Ja Jb] -- a batched (da + db)^2 product
    # per factor, scattered into the FIXED pattern of H through slot indices computed once.
    def _normal_setup(self):
        import scipy.sparse as sp
        n = self.n
        rows, cols = [], []
        for k, g in self.groups.items():
            idx = self.pos[g["ia"]] if g["ib"] is None else np.concatenate([self.pos[g["ia"]], self.pos[g["ib"]]], axis=1)   # (F, d)
            d = idx.shape[1]
            rows.append(np.repeat(idx[:, :, None], d, axis=2).ravel()); cols.append(np.repeat(idx[:, None, :], d, axis=1).ravel())
        rows, cols = np.concatenate(rows).astype(np.int64), np.concatenate(cols).astype(np.int64)
        S = sp.csc_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
        S.sum_duplicates(); S.sort_indices()
        keys = np.repeat(np.arange(n, dtype=np.int64), np.diff(S.indptr)) * n + S.indices       # ascending: column-major, rows sorted
        self.h_slot = np.searchsorted(keys, cols * n + rows)
        self.h_diag = np.searchsorted(keys, np.arange(n, dtype=np.int64) * (n + 1))
        if not (np.array_equal(keys[self.h_diag], np.arange(n, dtype=np.int64) * (n + 1))):
            raise AssertionError("an unknown without a diagonal entry in the normal matrix")
        self.h_indices, self.h_indptr, self.h_nnz = S.indices.copy(), S.indptr.copy(), int(S.nnz)

    def normal_matrix(self, blocks):
        """H = J^T J (scipy csc, in the elimination order) from the whitened blocks [(Ja (F, dr, da), Jb (F, dr, db) | None)] of linearize()"""
        import scipy.sparse as sp
        if not hasattr(self, "h_slot"):
            self._normal_setup()
        vals = []
        for Ja, Jb in blocks:
            A = Ja if Jb is None else np.concatenate([Ja, Jb], axis=2)
            vals.append(np.matmul(A.transpose(0, 2, 1), A).ravel())
        data = np.bincount(self.h_slot, weights=np.concatenate(vals), minlength=self.h_nnz)
        H = sp.csc_matrix((data, self.h_indices, self.h_indptr), shape=(self.n, self.n))
        H.has_sorted_indices = True
        return H

    def damped(self, H, lam):
        """H + lam * diag(diag(H) + 1e-12) on the fixed pattern (no sparse addition)"""
        Hl = H.copy()
        Hl.data[self.h_diag] += lam * (H.data[self.h_diag] + 1e-12)
        return Hl


def solveGraphParametric(fg, init=None, max_iters=100, tol=1e-4, ctx=None, return_cov=False, verbose=False, shard=None, stats=None, polish=6):
    """-> {label: coordinates} (and, if return_cov, {label: marginal covariance block} from (JᵀJ)⁻¹).
    Stops when the relative cost decrease of an accepted step falls below `tol` (small graphs converge
    quadratically; the low-frequency modes of a weakly anchored 3500-pose graph creep at 1e-4/iteration
    long after the pose error has reached the measurement-noise floor) or the step is below 1e-8; then up to `polish` undamped
    Gauss-Newton steps take the iterate onto the optimum along the directions the cost barely sees (see below).
    stats: a dict that receives where the time went -- {setup_s, linearize_s (rome_linearize calls: the batched residual / Jacobian
    kernels, their transfers and, when sharded, the exchange), host_solve_s (normal equations + sparse Cholesky / LU on the host),
    iterations, linearizations}: on a 10 000-pose helix the host solve is > 95 % of the wall-clock.
    The host arithmetic of the whole solve runs with the BLAS pools limited to one thread: every BLAS call
    of this loop is small -- SuperLU supernodes, 12 x 12 block products -- and the worker threads of a 64-thread OpenBLAS pool on a
    16-CPU quota spin after each of them while the sequential parts of the factorisation want the cores: 10k helix 6.1 -> 2.4 s"""
    with _blas_single_thread():
        return _solve_graph_parametric(fg, init, max_iters, tol, ctx, return_cov, verbose, shard, stats, polish)


def _solve_graph_parametric(fg, init=None, max_iters=100, tol=1e-4, ctx=None, return_cov=False, verbose=False, shard=None, stats=None, polish=6):
    """the body of solveGraphParametric (run under the BLAS thread limit)"""
    import time
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    T = dict(setup_s=0.0, linearize_s=0.0, host_solve_s=0.0, iterations=0, linearizations=0)
    t_ = time.perf_counter()
    P = _Problem(fg)
    X = P.pack(initParametric(fg) if init is None else init)
    lam = 1e-6
    T["setup_s"] = time.perf_counter() - t_

    def lin(Xv):
        t0 = time.perf_counter()
        out = P.linearize(Xv, ctx, shard)
        T["linearize_s"] += time.perf_counter() - t0; T["linearizations"] += 1
        return out
    r, J = lin(X)
    B = P.blocks
    cost = float(r @ r)
    for it in range(max_iters):
        if verbose:
            print('LM iter %d cost %.6g lambda %.1e' % (it, cost, lam))
        t0 = time.perf_counter()
        H = P.normal_matrix(B); g = J.T @ r
        T["host_solve_s"] += time.perf_counter() - t0
        T["iterations"] += 1
        while True:
            t0 = time.perf_counter()
            d = np.empty(P.n)
            d[P.perm] = P.solve_spd(P.damped(H, lam), -g)
            Xn = P.retract(X, d)
            T["host_solve_s"] += time.perf_counter() - t0
            rn, Jn = lin(Xn)
            Bn = P.blocks
            cn = float(rn @ rn)
            if cn <= cost or lam > 1e12:
                break
            lam *= 10.0
        if cn > cost:   # damping exhausted without a descent step: keep the last accepted iterate (never commit an uphill one)
            break
        done = (cost - cn) <= tol * max(1.0, cost) or np.abs(d).max() < 1e-8
        X, r, J, B, cost = Xn, rn, Jn, Bn, cn
        lam = max(lam / 10.0, 1e-12)
        if done:
            break
    # ---- polish: UNDAMPED Gauss-Newton steps.  The damped iteration above stops on the cost decrease, and a weakly anchored graph has
    # directions the cost barely sees: on Manhattan-3500 the rotation of the whole map about the prior pose costs ~1 unit of 3533 per
    # prior sigma, and an LM run that stops at a relative decrease of 1e-4 ... 1e-9 sits 0.3 ... 1 m RMS from the optimum
    # (scripts/tree_linear_surrogate.py: three starts, three "solutions" 0.7 - 1.5 m apart, ONE optimum after two undamped steps).
    # lam * diag(H) is what slows exactly those directions; a full step resolves them at once.
    for _ in range(int(polish)):
        t0 = time.perf_counter()
        H = P.normal_matrix(B); g = J.T @ r
        d = np.empty(P.n)
        d[P.perm] = P.solve_spd(P.damped(H, 1e-12), -g)
        Xn = P.retract(X, d)
        T["host_solve_s"] += time.perf_counter() - t0
        T["iterations"] += 1
        rn, Jn = lin(Xn)
        Bn = P.blocks
        cn = float(rn @ rn)
        if not cn <= cost * (1.0 + 1e-12):
            break                                       # a full step that goes uphill: the damped iterate stands
        small = np.abs(d).max() < 1e-7
        X, r, J, B, cost = Xn, rn, Jn, Bn, cn
        if small:
            break
    out = P.unpack(X)
    if stats is not None:
        stats.update(T)
        if shard is not None and hasattr(shard, "stats"):
            stats["shard"] = dict(shard.stats)
    if return_cov:
        C = np.empty((P.n, P.n))
        C[np.ix_(P.perm, P.perm)] = np.linalg.inv((J.T @ J).toarray())
        cov = {l: C[P.off[i]:P.off[i + 1], P.off[i]:P.off[i + 1]] for i, l in enumerate(P.labels)}
        return out, cov, dict(cost=cost)
    return out
