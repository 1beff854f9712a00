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


def initParametric(fg):
    """Deterministic initial values: prior means, then measurement means composed along relative factors
    (stand-in for initAll! + initParametricFrom!, test/testParametric.jl:32-33)."""
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
    return x


class _Problem:
    def __init__(self, fg):
        self.fg = fg
        self.labels = list(fg.variables)
        self.vt = [fg.variables[l] for l in self.labels]
        self.off = np.concatenate([[0], np.cumsum([t.dim for t in self.vt])])
        self.index = {l: i for i, l in enumerate(self.labels)}
        groups = {}
        for _, labels, f in fg.factors:
            k = _KIND.get(type(f))
            if k is None:
                raise TypeError("factor %s is outside the hot path" % type(f).__name__)
            mu, info = getMeasurementParametric(f)
            g = groups.setdefault(k, dict(mu=[], W=[], a=[], b=[]))
            g["mu"].append(mu); g["W"].append(_whitening(info)); g["a"].append(self.index[labels[0]])
            g["b"].append(self.index[labels[1]] if len(labels) > 1 else -1)
        self.groups = {k: {n: np.asarray(v) for n, v in g.items()} for k, g in groups.items()}

    def linearize(self, x, ctx=None):
        """-> (r (m,), J scipy.sparse.csr (m, n))"""
        import scipy.sparse as sp
        rows, cols, vals, rs = [], [], [], []
        m = 0
        for k, g in self.groups.items():
            xa = np.stack([x[i] for i in g["a"]])
            xb = np.stack([x[i] for i in g["b"]]) if g["b"][0] >= 0 else None
            r, Ja, Jb = api.linearize(k, g["mu"], g["W"], xa, xb, ctx=ctx)
            F, dr = r.shape
            ridx = m + np.arange(F * dr).reshape(F, dr)
            rs.append(r.ravel())
            for J, vidx in ((Ja, g["a"]), (Jb, g["b"])):
                if J is None:
                    continue
                dv = J.shape[2]
                cidx = self.off[vidx][:, None] + np.arange(dv)[None, :]
                rows.append(np.repeat(ridx[:, :, None], dv, axis=2).ravel())
                cols.append(np.repeat(cidx[:, None, :], dr, axis=1).ravel())
                vals.append(J.ravel())
            m += F * dr
        J = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, self.off[-1]))
        return np.concatenate(rs), J


def solveGraphParametric(fg, init=None, max_iters=100, tol=1e-12, ctx=None, return_cov=False):
    """-> {label: coordinates} (and, if return_cov, {label: marginal covariance block} from (JᵀJ)⁻¹)."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    P = _Problem(fg)
    x0 = initParametric(fg) if init is None else init
    x = [np.array(x0[l], dtype=float) for l in P.labels]
    lam = 1e-6
    r, J = P.linearize(x, ctx)
    cost = float(r @ r)
    for _ in range(max_iters):
        H = (J.T @ J).tocsc(); g = J.T @ r
        D = sp.diags(H.diagonal() + 1e-12)
        while True:
            d = spsolve((H + lam * D).tocsc(), -g)
            xn = [_retract(P.vt[i], x[i], d[P.off[i]:P.off[i + 1]]) for i in range(len(x))]
            rn, Jn = P.linearize(xn, ctx)
            cn = float(rn @ rn)
            if cn <= cost or lam > 1e12:
                break
            lam *= 10.0
        done = (cost - cn) <= tol * max(1.0, cost) and np.abs(d).max() < 1e-9
        x, r, J, cost = xn, rn, Jn, cn
        lam = max(lam / 10.0, 1e-12)
        if done:
            break
    out = {l: x[i] for i, l in enumerate(P.labels)}
    out_info = dict(cost=cost)
    if return_cov:
        Hd = (J.T @ J).toarray()
        C = np.linalg.inv(Hd)
        cov = {l: C[P.off[i]:P.off[i + 1], P.off[i]:P.off[i + 1]] for i, l in enumerate(P.labels)}
        return out, cov, out_info
    return out
