"""Linear-Gaussian surrogate of the Bayes-tree solve with STRUCTURED separator messages (design tool, CPU only; not on the product path).

Every pass linearises all factors at the current pose estimates (information form on additive (dx, dy, dtheta) corrections) and runs
the multifrontal elimination over rome_jl_amd.tree.BayesTree -- which is EXACTLY one Gauss-Newton step when the separator messages are
the exact (dense) marginals.  The experiment replaces the exact message over a clique's separators S by a TREE-STRUCTURED approximation

        p(S)  ~  prod_{(j,k) in T} p(s_j, s_k) / prod_j p(s_j)^(deg_T(j) - 1)          (T a spanning tree over S)

 * star     T = star about the first separator (what tree.py's "relative" form sends: anchor marginal + p(s | anchor) per s)
 * mst      T = the spanning tree that keeps the tightest pairs (minimum log det of the conditional covariance of s_k given s_j)
 * exact    the dense marginal (reference: Gauss-Newton)

and reports the RMS of the translations to the parametric optimum after each pass: the BIAS of a message structure, free of
sampling noise.  A nonparametric implementation of a pair marginal = the samples of s_j^-1 * s_k under the clique's potential.

    python scripts/tree_linear_surrogate.py [--edges N] [--passes 4] [--forms star,mst,exact]
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scripts.tree_surrogate import load, wrap, parametric, oplus, conv, product   # noqa: E402


def linearise(E, X, prior):
    """-> per factor (vars, Lam (3m x 3m), eta (3m))"""
    out = []
    for (i, j, mu, C) in E:   # residual in p's frame (the measurement's own frame: constant information): r = p^-1 q (-) mu
        p, q = X[i], X[j]
        c, s = np.cos(p[2]), np.sin(p[2])
        dt = q[:2] - p[:2]
        Rt = np.array([[c, s], [-s, c]])
        e = np.concatenate([Rt @ dt - mu[:2], [wrap(q[2] - p[2] - mu[2])]])
        Jp = np.zeros((3, 3)); Jq = np.zeros((3, 3))
        Jp[:2, :2] = -Rt; Jp[:2, 2] = np.array([[-s, c], [-c, -s]]) @ dt; Jp[2, 2] = -1.0
        Jq[:2, :2] = Rt; Jq[2, 2] = 1.0
        Om = np.linalg.inv(C)
        J = np.hstack([Jp, Jq])
        out.append(((i, j), J.T @ Om @ J, -J.T @ Om @ e))
    e = X[0] - prior[0]; e[2] = wrap(e[2])
    Om = np.linalg.inv(prior[1])
    out.append(((0,), Om, -Om @ e))
    return out


def assemble(lin, n):
    import scipy.sparse as sp
    rows, cols, vals, g = [], [], [], np.zeros(3 * n)
    for vs, L, h in lin:
        ii = np.concatenate([np.arange(3 * v, 3 * v + 3) for v in vs])
        r, c = np.meshgrid(ii, ii, indexing="ij")
        rows.append(r.ravel()); cols.append(c.ravel()); vals.append(L.ravel()); g[ii] += h
    return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(3 * n, 3 * n)), g


def true_optimum(E, n, prior, X):
    """the MAP of the reference's cost (residual in the measurement's frame), converged with UNDAMPED Gauss-Newton steps at the end:
    a Levenberg-Marquardt run that stops on a relative cost decrease of 1e-4 ... 1e-9 sits 0.3 ... 1 m RMS away from it on
    Manhattan-3500 -- the rotation of the map about the prior pose costs ~1 unit of 3533 per sigma"""
    from scipy.sparse.linalg import spsolve
    import scipy.sparse as sp
    lam = 1e-3
    for it in range(60):
        H, g = assemble(linearise(E, X, prior), n)
        d = spsolve(H + lam * sp.diags(H.diagonal()), g).reshape(n, 3)
        X = X + d; X[:, 2] = wrap(X[:, 2])
        lam = max(lam * 0.3, 0.0 if np.abs(d).max() < 0.5 else 1e-9)
        if np.abs(d).max() < 1e-7:
            break
    return X


def schur(L, h, keep, drop):
    """marginal information over `keep` (index arrays into L)"""
    if len(drop) == 0:
        return L[np.ix_(keep, keep)], h[keep]
    A = L[np.ix_(drop, drop)]
    Bm = L[np.ix_(keep, drop)]
    Ai = np.linalg.inv(A + 1e-12 * np.eye(len(drop)))
    return L[np.ix_(keep, keep)] - Bm @ Ai @ Bm.T, h[keep] - Bm @ Ai @ h[drop]


def idx(vs_pos):
    return np.concatenate([np.arange(3 * k, 3 * k + 3) for k in vs_pos]) if len(vs_pos) else np.zeros(0, dtype=int)


def local_tree(V, S, local_edges, stats):
    """spanning tree over the separators S from shortest paths in the clique-LOCAL graph (the clique's own factors: weight 1; the
    children's message-tree edges: their path lengths) -- host-side, needs no beliefs.  -> (edges as positions in S, their lengths)"""
    import heapq
    adj = {v: [] for v in V}
    for a_, b_, w in local_edges:
        adj[a_].append((b_, w)); adj[b_].append((a_, w))
    D = {}
    for s0 in S:
        dist = {s0: 0.0}; pq = [(0.0, s0)]
        while pq:
            d0, u = heapq.heappop(pq)
            if d0 > dist.get(u, np.inf):
                continue
            for v, w in adj[u]:
                if d0 + w < dist.get(v, np.inf):
                    dist[v] = d0 + w; heapq.heappush(pq, (d0 + w, v))
        D[s0] = dist
    m = len(S)
    intree, edges, lens = {0}, [], []
    while len(intree) < m:
        best = None
        for j in intree:
            for k in range(m):
                if k not in intree:
                    w = D[S[j]].get(S[k], 1e9)
                    if best is None or w < best[0]:
                        best = (w, j, k)
        edges.append((best[1], best[2])); lens.append(best[0]); intree.add(best[2])
    stats["internal"] += len({j for j, _ in edges})
    stats["msgs"] += 1
    return edges, lens


def structured(L, h, m, form, stats, edges=None):
    """L, h over m separators -> the tree-structured approximation (same shape)"""
    if form == "exact" or m <= 2:
        return L, h
    pair = {}

    def pm(j, k):
        if (j, k) not in pair:
            keep = idx([j, k]); drop = idx([r for r in range(m) if r not in (j, k)])
            pair[(j, k)] = schur(L, h, keep, drop)
        return pair[(j, k)]

    if edges is not None:
        edges = [(min(j, k), max(j, k)) for j, k in edges]
    elif form == "star":
        edges = [(0, k) for k in range(1, m)]
    else:
        # weight: log det of the covariance of s_k given s_j under the pair marginal (symmetric enough; take the smaller direction)
        W = np.full((m, m), np.inf)
        for j in range(m):
            for k in range(j + 1, m):
                Lp, _ = pm(j, k)
                w1 = -np.linalg.slogdet(Lp[3:, 3:] + 1e-12 * np.eye(3))[1]
                w2 = -np.linalg.slogdet(Lp[:3, :3] + 1e-12 * np.eye(3))[1]
                W[j, k] = W[k, j] = min(w1, w2)
        intree, edges = {0}, []
        while len(intree) < m:   # Prim
            best = None
            for j in intree:
                for k in range(m):
                    if k not in intree and (best is None or W[j, k] < best[0]):
                        best = (W[j, k], j, k)
            edges.append((min(best[1], best[2]), max(best[1], best[2]))); intree.add(best[2])
    La, ha = np.zeros_like(L), np.zeros_like(h)
    deg = np.zeros(m, dtype=int)
    for j, k in edges:
        Lp, hp = pm(j, k)
        ii = idx([j, k])
        La[np.ix_(ii, ii)] += Lp; ha[ii] += hp
        deg[j] += 1; deg[k] += 1
    for j in range(m):
        if deg[j] > 1:
            Lj, hj = schur(L, h, idx([j]), idx([r for r in range(m) if r != j]))
            ii = idx([j])
            La[np.ix_(ii, ii)] -= (deg[j] - 1) * Lj; ha[ii] -= (deg[j] - 1) * hj
    stats["approx"] += 1
    return La, ha


def tree_step(bt, lin, nf_prior, n, form, stats):
    """one approximate elimination + back-substitution -> delta (n, 3)"""
    msgs = {c.id: [] for c in bt.cliques}
    medges = {c.id: [] for c in bt.cliques}       # clique -> [(u, v, length)] message-tree edges handed up by the children
    cond = {}
    for lvl in bt.levels:
        for c in lvl:
            cl = bt.cliques[c]
            V = list(cl.frontals) + list(cl.separators)
            pos = {v: k for k, v in enumerate(V)}
            L = np.zeros((3 * len(V), 3 * len(V))); h = np.zeros(3 * len(V))
            for fid in cl.factors:
                vs, Lf, hf = lin[fid]
                ii = idx([pos[v] for v in vs])
                L[np.ix_(ii, ii)] += Lf; h[ii] += hf
            for vs, Lm, hm in msgs[c]:
                ii = idx([pos[v] for v in vs])
                L[np.ix_(ii, ii)] += Lm; h[ii] += hm
            nF, nS = len(cl.frontals), len(cl.separators)
            iF, iS = idx(range(nF)), idx(range(nF, nF + nS))
            A = L[np.ix_(iF, iF)]
            Ai = np.linalg.inv(A)
            cond[c] = (Ai, L[np.ix_(iF, iS)], h[iF])
            if cl.parent >= 0:
                Ls, hs = schur(L, h, iS, iF)
                ed = None
                if form == "hop":
                    S = list(cl.separators)
                    loc = [(lin[fid][0][0], lin[fid][0][1], 1.0) for fid in cl.factors if len(lin[fid][0]) == 2] + medges[c]
                    ed, ln = local_tree(V, S, loc, stats) if nS > 1 else ([], [])
                    medges[cl.parent] += [(S[j], S[k], w) for (j, k), w in zip(ed, ln)]
                Ls, hs = structured(Ls, hs, nS, form, stats, edges=ed if nS > 2 else None)
                msgs[cl.parent].append((list(cl.separators), Ls, hs))
    d = np.zeros((n, 3))
    for lvl in bt.levels[::-1]:
        for c in lvl:
            cl = bt.cliques[c]
            Ai, Lfs, hf = cond[c]
            ds = np.concatenate([d[v] for v in cl.separators]) if cl.separators else np.zeros(0)
            df = Ai @ (hf - Lfs @ ds)
            for k, v in enumerate(cl.frontals):
                d[v] = df[3 * k:3 * k + 3]
    return d


def init_pass(E, n, prior):
    B = {0: prior}
    adj = {v: [] for v in range(n)}
    for k, (i, j, mu, C) in enumerate(E):
        adj[i].append((k, j, 1)); adj[j].append((k, i, 0))
    while len(B) < n:
        new = {}
        for v in range(n):
            if v in B:
                continue
            pr = [conv(B[o], E[k][2], E[k][3], d) for k, o, d in adj[v] if o in B]
            if pr:
                new[v] = product(pr)
        B.update(new)
    return np.array([B[v][0] for v in range(n)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--g2o", default="tests/golden/manhattan.g2o")
    ap.add_argument("--edges", type=int, default=None)
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--forms", default="star,mst,exact")
    ap.add_argument("--at-optimum", action="store_true", help="start AT the parametric optimum: the drift away from it is the fixed-point bias")
    a = ap.parse_args()
    from rome_jl_amd import tree as T
    E = load(a.g2o, a.edges)
    n = 1 + max(max(i, j) for i, j, _, _ in E)
    prior = (np.zeros(3), np.diag([0.01, 0.01, 0.0025]))
    t0 = time.time()
    Xw, _ = parametric(E, n, prior)          # (tree_surrogate's solver: a start, not the reference)
    Xp = true_optimum(E, n, prior, Xw)
    print("MAP: converged in %.1fs, %.2f m RMS from the loosely converged start" % (time.time() - t0, np.sqrt(np.mean(np.sum((Xw[:, :2] - Xp[:, :2]) ** 2, axis=1)))))
    factors = [(k, (i, j)) for k, (i, j, _, _) in enumerate(E)] + [(len(E), (0,))]
    bt = T.BayesTree.build(list(range(n)), factors, order="mmd")
    print(bt.summary())
    rms = lambda X: np.sqrt(np.mean(np.sum((X[:, :2] - Xp[:, :2]) ** 2, axis=1)))   # noqa: E731
    X0 = Xp.copy() if a.at_optimum else init_pass(E, n, prior)
    print("start: RMS %.3f m" % rms(X0))
    for form in a.forms.split(","):
        X = X0.copy()
        out = []
        for ps in range(a.passes):
            stats = {"approx": 0, "internal": 0, "msgs": 0}
            t0 = time.time()
            lin = linearise(E, X, prior)
            d = tree_step(bt, lin, len(E), n, form, stats)
            # a trust-region-free step is what a sampled solve takes as well; clip heading corrections to keep the linearisation sane
            X = X + d; X[:, 2] = wrap(X[:, 2])
            out.append(rms(X))
            print("  %-6s pass %d: RMS %.3f m   (max |d| %.2f, %d structured messages, %.1fs)" % (form, ps, out[-1], np.abs(d[:, :2]).max(), stats["approx"], time.time() - t0) + ("  anchors/message %.2f" % (stats["internal"] / max(stats["msgs"], 1)) if stats["msgs"] else ""), flush=True)


if __name__ == "__main__":
    main()
