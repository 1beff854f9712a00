"""Gaussian surrogate of the Bayes-tree schedule (design tool, CPU only; nothing here is on the product path).

Every belief is a Gaussian (mean, covariance) on Pose2 coordinates; a Pose2Pose2 convolution pushes it through the factor with
first-order Jacobians, a product is information-weighted fusion (heading differences wrapped).  The SCHEDULE is the one of
rome_jl_amd/tree.py: elimination order -> cliques (frontals | separators) -> levels; up pass with clique-local separator copies and
per-variable separator marginals as messages; down pass with the separators fixed at their posteriors.  Used to answer, before any
kernel runs: does the schedule remove the loop error that the init pass leaves on Manhattan-3500, and what do ordering / Gibbs
iterations / repeated solves buy?

    python scripts/tree_surrogate.py [--edges N] [--gibbs 3] [--passes 2]
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def load(path, max_edges=None):
    E = []
    for ln in open(path):
        t = ln.split()
        if t and t[0] == "EDGE_SE2":
            i, j = int(t[1]), int(t[2])
            mu = np.array([float(x) for x in t[3:6]])
            u = [float(x) for x in t[6:12]]
            L = np.array([[u[0], u[1], u[2]], [u[1], u[3], u[4]], [u[2], u[4], u[5]]])
            C = np.linalg.inv(L); C = 0.5 * (C + C.T)
            E.append((i, j, mu, C))
            if max_edges and len(E) >= max_edges:
                break
    return E


def oplus(p, z):
    c, s = np.cos(p[2]), np.sin(p[2])
    return np.array([p[0] + c * z[0] - s * z[1], p[1] + s * z[0] + c * z[1], wrap(p[2] + z[2])])


def ominus(q, z):   # p with p (+) z = q
    th = q[2] - z[2]
    c, s = np.cos(th), np.sin(th)
    return np.array([q[0] - c * z[0] + s * z[1], q[1] - s * z[0] - c * z[1], wrap(th)])


def conv(bel, z, Cz, d):
    """d = 0: belief of the first variable -> second; 1: second -> first"""
    m, S = bel
    if d == 0:
        c, s = np.cos(m[2]), np.sin(m[2])
        A = np.array([[1, 0, -s * z[0] - c * z[1]], [0, 1, c * z[0] - s * z[1]], [0, 0, 1.0]])
        B = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        return oplus(m, z), A @ S @ A.T + B @ Cz @ B.T
    th = m[2] - z[2]
    c, s = np.cos(th), np.sin(th)
    dt = np.array([s * z[0] + c * z[1], -c * z[0] + s * z[1]])     # d p.t / d th
    A = np.array([[1, 0, dt[0]], [0, 1, dt[1]], [0, 0, 1.0]])
    B = np.array([[-c, s, -dt[0]], [-s, -c, -dt[1]], [0, 0, -1.0]])
    return ominus(m, z), A @ S @ A.T + B @ Cz @ B.T


def product(bels):
    if len(bels) == 1:
        return bels[0]
    m0 = bels[0][0]
    Li = np.zeros((3, 3)); h = np.zeros(3)
    for m, S in bels:
        I = np.linalg.inv(S)
        d = m - m0; d[2] = wrap(d[2])
        Li += I; h += I @ d
    S = np.linalg.inv(Li)
    m = m0 + S @ h; m[2] = wrap(m[2])
    return m, S


def parametric(E, n, prior, iters=30):
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    X = np.zeros((n, 3))
    done = {0}
    for i, j, mu, C in E:   # odometry chain first
        if j == i + 1:
            X[j] = oplus(X[i], mu)
    W = [np.linalg.cholesky(np.linalg.inv(C)).T for _, _, _, C in E]
    Wp = np.linalg.cholesky(np.linalg.inv(prior[1])).T
    lam = 1e-4
    def cost_J(X):
        rows, cols, vals, r = [], [], [], []
        k = 0
        for (i, j, mu, C), w in zip(E, W):
            p, q = X[i], X[j]
            c, s = np.cos(p[2]), np.sin(p[2])
            e = np.array([p[0] + c * mu[0] - s * mu[1] - q[0], p[1] + s * mu[0] + c * mu[1] - q[1], wrap(p[2] + mu[2] - q[2])])
            Jp = np.array([[1, 0, -s * mu[0] - c * mu[1]], [0, 1, c * mu[0] - s * mu[1]], [0, 0, 1.0]])
            r.append(w @ e)
            for (v, J) in ((i, w @ Jp), (j, -w)):
                for a in range(3):
                    for b in range(3):
                        rows.append(k + a); cols.append(3 * v + b); vals.append(J[a, b])
            k += 3
        e = X[0] - prior[0]; e[2] = wrap(e[2])
        r.append(Wp @ e)
        for a in range(3):
            for b in range(3):
                rows.append(k + a); cols.append(b); vals.append(Wp[a, b])
        k += 3
        return np.concatenate(r), sp.csc_matrix((vals, (rows, cols)), shape=(k, 3 * n))
    r, J = cost_J(X); c0 = r @ r
    for it in range(iters):
        H = (J.T @ J).tocsc(); g = J.T @ r
        while True:
            d = spsolve(H + lam * sp.diags(H.diagonal()), -g).reshape(n, 3)
            Xn = X + d; Xn[:, 2] = wrap(Xn[:, 2])
            rn, Jn = cost_J(Xn); cn = rn @ rn
            if cn <= c0 or lam > 1e8:
                break
            lam *= 10
        rel = (c0 - cn) / max(c0, 1)
        X, r, J, c0 = Xn, rn, Jn, cn
        lam = max(lam / 10, 1e-9)
        if rel < 1e-6:
            break
    return X, c0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--g2o", default="tests/golden/manhattan.g2o")
    ap.add_argument("--edges", type=int, default=None)
    ap.add_argument("--gibbs", type=int, default=3)
    ap.add_argument("--down", type=int, default=1)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--order", default="mmd")
    ap.add_argument("--up-seps", type=int, default=1, help="update separator copies in the up pass (IIF) or only read them")
    a = ap.parse_args()
    from rome_jl_amd import tree as T
    E = load(a.g2o, a.edges)
    n = 1 + max(max(i, j) for i, j, _, _ in E)
    prior = (np.zeros(3), np.diag([0.01, 0.01, 0.0025]))
    t0 = time.time()
    Xp, cost = parametric(E, n, prior)
    print("parametric: cost %.1f  (%.1fs)" % (cost, time.time() - t0))
    labels = list(range(n))
    factors = [(k, (i, j)) for k, (i, j, _, _) in enumerate(E)] + [(len(E), (0,))]
    bt = T.BayesTree.build(labels, factors, order=a.order)
    print(bt.summary())

    def rms(B):
        M = np.array([B[v][0] for v in range(n)])
        return np.sqrt(np.mean(np.sum((M[:, :2] - Xp[:, :2]) ** 2, axis=1)))

    # ---- init pass (IIF initAll!): rounds outward from the prior, product of the factors whose other end is initialised
    B = {0: prior}
    adj = {v: [] for v in range(n)}
    for k, (i, j, mu, C) in enumerate(E):
        adj[i].append((k, j, 1)); adj[j].append((k, i, 0))      # (factor, other, direction that targets this variable)
    while len(B) < n:
        new = {}
        for v in range(n):
            if v in B:
                continue
            pr = [conv(B[o], E[k][2], E[k][3], d) for k, o, d in adj[v] if o in B]
            if pr:
                new[v] = product(pr)
        B.update(new)
    print("init pass: RMS %.3f m" % rms(B))

    for ps in range(a.passes):
        # ---- up pass
        loc = {}                                   # (clique, variable) -> belief of the clique-local copy
        msgs = {c: [] for c in range(len(bt.cliques))}   # clique -> [(variable, belief)] from children
        for lvl in bt.levels:
            for c in lvl:
                F, S = bt.cliques[c].frontals, bt.cliques[c].separators
                cur = {v: B[v] for v in F}
                cur.update({v: B[v] for v in S})
                fs = bt.cliques[c].factors
                upd = list(F) + (list(S) if a.up_seps else [])
                for it in range(a.gibbs):
                    for v in upd:
                        pr = []
                        for k in fs:
                            if k == len(E):
                                if v == 0:
                                    pr.append(prior)
                                continue
                            i, j, mu, C = E[k]
                            if v == j:
                                pr.append(conv(cur[i], mu, C, 0))
                            elif v == i:
                                pr.append(conv(cur[j], mu, C, 1))
                        pr += [b for (mv, b) in msgs[c] if mv == v]
                        if pr:
                            cur[v] = product(pr)
                for v in F:
                    B[v] = cur[v]
                p = bt.cliques[c].parent
                if p >= 0:
                    for v in S:
                        if a.up_seps:
                            msgs[p].append((v, cur[v]))
                        else:   # message = product of this clique's proposals on the separator
                            pr = []
                            for k in fs:
                                if k == len(E):
                                    continue
                                i, j, mu, C = E[k]
                                if v == j and i in F:
                                    pr.append(conv(cur[i], mu, C, 0))
                                elif v == i and j in F:
                                    pr.append(conv(cur[j], mu, C, 1))
                            pr += [b for (mv, b) in msgs[c] if mv == v]
                            if pr:
                                msgs[p].append((v, product(pr)))
        print("pass %d up:   RMS %.3f m" % (ps, rms(B)))
        # ---- down pass
        for lvl in bt.levels[::-1]:
            for c in lvl:
                F, S = bt.cliques[c].frontals, bt.cliques[c].separators
                if not S:
                    continue
                fs = bt.cliques[c].factors
                for it in range(a.down):
                    for v in F:
                        pr = []
                        for k in fs:
                            if k == len(E):
                                if v == 0:
                                    pr.append(prior)
                                continue
                            i, j, mu, C = E[k]
                            if v == j:
                                pr.append(conv(B[i], mu, C, 0))
                            elif v == i:
                                pr.append(conv(B[j], mu, C, 1))
                        pr += [b for (mv, b) in msgs[c] if mv == v]
                        if pr:
                            B[v] = product(pr)
        print("pass %d down: RMS %.3f m" % (ps, rms(B)))


if __name__ == "__main__":
    main()
