"""A GAUSSIAN backend for rome_jl_amd.tree.TreeSolver (design tool, CPU only; not on the product path).

It executes EXACTLY the schedule the device executes -- the LevelSpecs of tree.TreeSolver: lifted labels, update groups, (factor,
destination) rows, store-resident messages, anchor / relative / copy block operations, two-stage products -- with every belief a
Gaussian (mean, covariance) on Pose2 coordinates: a convolution is an unscented transform through the factor, a product is
information fusion (headings wrapped), an anchor block is the mean with ~zero spread, a relative block the Gaussian of anchor^-1 * s.
Deterministic: what it returns is the BIAS of a message structure / solve order under the real schedule's semantics (outward solves, not
exact marginals), free of the sampling noise of N = 100 particles.

    python scripts/tree_gaussian_backend.py [--edges N] [--passes 4] [--forms star,hop] [--kw relIters=1]
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scripts.tree_surrogate import load, wrap, parametric, oplus, ominus, product   # noqa: E402
from scripts.tree_linear_surrogate import true_optimum, init_pass   # noqa: E402

_SP = None
TEMPER = 1.0      # experiment: the centre edge's spread scaled by this before a composition


def _sigma(n):
    global _SP
    if _SP is None or _SP[0] != n:
        k = 3.0 - n if n < 3 else 0.0
        lam = k
        W = np.full(2 * n + 1, 1.0 / (2 * (n + lam))); W[0] = lam / (n + lam)
        _SP = (n, lam, W)
    return _SP


def ut_conv(bel, zm, Cz, d):
    """unscented transform of (belief, measurement) through q = p (+) z (d = 0) or p = q (-) z (d = 1)"""
    m, S = bel
    n = 6
    _, lam, W = _sigma(n)
    P = np.zeros((6, 6)); P[:3, :3] = S; P[3:, 3:] = Cz
    L = np.linalg.cholesky((n + lam) * (P + 1e-15 * np.eye(6)))
    x0 = np.concatenate([m, zm])
    pts = [x0] + [x0 + L[:, k] for k in range(n)] + [x0 - L[:, k] for k in range(n)]
    f = oplus if d == 0 else ominus
    Y = np.array([f(p[:3], p[3:]) for p in pts])
    y0 = Y[0].copy()
    D = Y - y0; D[:, 2] = wrap(D[:, 2])
    dm = W @ D
    E = D - dm
    C = (E * W[:, None]).T @ E
    out = y0 + dm; out[2] = wrap(out[2])
    return out, 0.5 * (C + C.T) + 1e-15 * np.eye(3)


class GStore:
    def __init__(self, R, universe):
        self.R, self.fg, self.N = R, universe, universe.N
        self.vals, self.index = {}, {l: k for k, l in enumerate(universe.variables)}
        self.touched = set()

    def upload(self, fg, labels=None):
        pass                                                      # (the driver sets Gaussian beliefs directly)

    def download(self, fg, labels=None):
        pass

    def put(self, label, pts):
        pts = np.asarray(pts, dtype=float)
        self.vals[label] = (pts.mean(axis=1), np.cov(pts) + 1e-12 * np.eye(pts.shape[0]))

    def get(self, label):
        return self.vals[label]


class GPlan:
    def __init__(self, store, spec, share=None, mirror=None):
        self.store, self.spec = store, spec
        L = spec.fg
        self.by_dest = {}
        for fl, dst in spec.pairs:
            self.by_dest.setdefault(dst, []).append(L.getFactor(fl))
        self.msg = {}
        for src, dst in spec.smsgs:
            self.msg.setdefault(dst, []).append(src)

    def run(self, opts, **_):
        from rome_jl_amd.clique import SampledPose2Pose2
        v, sp = self.store.vals, self.spec
        for it in range(sp.gibbs_iters):
            for g in sorted(set(sp.groups)):
                new = {}
                for l, gl in zip(sp.order, sp.groups):
                    if gl != g:
                        continue
                    pr = []
                    for fl, labels, f in self.by_dest.get(l, ()):
                        if len(labels) == 1:
                            pr.append((np.array(f.Z.mu, dtype=float), np.array(f.Z.cov, dtype=float)))
                            continue
                        if isinstance(f, SampledPose2Pose2):
                            zm, Cz = v[f.meas]
                        else:
                            zm, Cz = f.Z.mu, f.Z.cov
                        a, b = labels
                        pr.append(ut_conv(v[a], zm, Cz, 0) if l == b else ut_conv(v[b], zm, Cz, 1))
                    pr += [v[s] for s in self.msg.get(l, ())]
                    if pr:
                        new[l] = product(pr)
                v.update(new)


class GBlockOp:
    def __init__(self, store, op, entries):
        self.store, self.op, self.entries = store, op, list(entries)

    def run(self):
        v = self.store.vals
        for e in self.entries:
            if self.op == "copy":
                v[e[1]] = v[e[0]]
            elif self.op == "anchor":
                v[e[1]] = (v[e[0]][0].copy(), 1e-12 * np.eye(3))
            elif self.op == "mix":       # (Gaussian stand-in: the running average of the means)
                (mp, Sp), (mn, Sn), p_ = v[e[0]], v[e[1]], float(e[2])
                d = mn - mp; d[2] = wrap(d[2])
                m = mp + d / p_; m[2] = wrap(m[2])
                v[e[1]] = (m, Sn)
            elif self.op == "compose":
                def inv(z):
                    c, s = np.cos(z[2]), np.sin(z[2])
                    return np.array([-(c * z[0] + s * z[1]), -(-s * z[0] + c * z[1]), -z[2]])
                (ma, Sa), (mb, Sb) = v[e[0]], v[e[1]]
                n = 6
                _, lam, W = _sigma(n)
                P = np.zeros((6, 6)); P[:3, :3] = Sa * TEMPER; P[3:, 3:] = Sb
                Lc = np.linalg.cholesky((n + lam) * (P + 1e-15 * np.eye(6)))
                x0 = np.concatenate([ma, mb])
                pts = [x0] + [x0 + Lc[:, k] for k in range(n)] + [x0 - Lc[:, k] for k in range(n)]
                Y = np.array([oplus(inv(p[:3]) if e[3] else p[:3], inv(p[3:]) if e[4] else p[3:]) for p in pts])
                y0 = Y[0].copy()
                D = Y - y0; D[:, 2] = wrap(D[:, 2])
                dm = W @ D
                Ed = D - dm
                Cc = (Ed * W[:, None]).T @ Ed
                out = y0 + dm; out[2] = wrap(out[2])
                Cc = 0.5 * (Cc + Cc.T)
                if len(e) > 5:      # star-mesh inflation of the composed spread (translation, heading)
                    g = np.diag([e[5], e[5], e[6]])
                    Cc = g @ Cc @ g
                v[e[2]] = (out, Cc + 1e-15 * np.eye(3))
            else:
                ref = v[e[0]][0]
                m, S = v[e[1]]
                c, s = np.cos(ref[2]), np.sin(ref[2])
                A = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1.0]])
                d = m - ref; d[2] = wrap(d[2])
                v[e[2]] = (A @ d, A @ S @ A.T)


class GaussianBackend:
    def __init__(self, R):
        self.R = R

    def Store(self, universe):
        return GStore(self.R, universe)

    def Plan(self, store, spec, share=None, mirror=None):
        return GPlan(store, spec)

    def BlockOp(self, store, op, entries):
        return GBlockOp(store, op, entries)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--g2o", default="tests/golden/manhattan.g2o")
    ap.add_argument("--edges", type=int, default=None)
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--forms", default="star,hop")
    ap.add_argument("--kw", default="")
    a = ap.parse_args()
    import rome_jl_amd as R
    from rome_jl_amd.tree import TreeSolver
    E = load(a.g2o, a.edges)
    n = 1 + max(max(i, j) for i, j, _, _ in E)
    prior = (np.zeros(3), np.diag([0.01, 0.01, 0.0025]))
    Xw, _ = parametric(E, n, prior)
    Xp = true_optimum(E, n, prior, Xw)
    fg = R.loadG2o(a.g2o, N=100, max_edges=a.edges)
    labels = ["x%d" % k for k in range(n)]
    rms = lambda M: np.sqrt(np.mean(np.sum((M[:, :2] - Xp[:, :2]) ** 2, axis=1)))   # noqa: E731

    def aligned(M):
        A, B = M[:, :2] - M[:, :2].mean(0), Xp[:, :2] - Xp[:, :2].mean(0)
        U, _, Vt = np.linalg.svd(A.T @ B)
        Rm = U @ np.diag([1, np.sign(np.linalg.det(U @ Vt))]) @ Vt
        return np.sqrt(np.mean(np.sum((A @ Rm - B) ** 2, axis=1)))
    # init pass (Gaussian restatement of initAll!)
    B0 = {}
    adj = {v: [] for v in range(n)}
    for k, (i, j, mu, C) in enumerate(E):
        adj[i].append((k, j, 1)); adj[j].append((k, i, 0))
    B0[0] = prior
    while len(B0) < n:
        new = {}
        for v in range(n):
            if v not in B0:
                pr = [ut_conv(B0[o], E[k][2], E[k][3], d) for k, o, d in adj[v] if o in B0]
                if pr:
                    new[v] = product(pr)
        B0.update(new)
    M0 = np.array([B0[v][0] for v in range(n)])
    print("init pass: RMS %.3f m raw, %.3f m aligned" % (rms(M0), aligned(M0)))
    kw = {k: (v if v.isalpha() else float(v) if "." in v or "e" in v else int(v)) for k, v in (kv.split("=") for kv in a.kw.split(",") if kv)}
    for form in a.forms.split(","):
        t0 = time.time()
        if form == "elimination":
            from rome_jl_amd.elimination import RelativeEliminationSolver
            ts = RelativeEliminationSolver(fg, backend=GaussianBackend(R), **kw)
            print("  ", ts.stats())
        else:
            ts = TreeSolver(fg, messages="marginal" if form == "marginal" else "relative", backend=GaussianBackend(R),
                            **({} if form == "marginal" else {"message_tree": form}), **kw)
            for v in range(n):
                ts.store.vals[labels[v]] = B0[v]
        tb = time.time() - t0
        out = []
        for ps in range(a.passes):
            t0 = time.time()
            ts.solve(R.make_opts(N=100, seed=ps))
            M = np.array([ts.store.vals[l][0] for l in labels])
            out.append((rms(M), aligned(M)))
            print("  %-8s pass %d: RMS %.3f m raw, %.3f m aligned   (%.1f s; build %.1f s)" % (form, ps, out[-1][0], out[-1][1], time.time() - t0, tb), flush=True)


if __name__ == "__main__":
    main()
