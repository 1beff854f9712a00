"""Linear-Gaussian surrogate of VARIABLE ELIMINATION IN RELATIVE-FACTOR ALGEBRA (design tool, CPU only).

The graph's edges carry pairwise information blocks.  Eliminating v with neighbours u_1..u_m leaves a dense marginal over the
neighbours; it is replaced by a TREE of pairwise marginals (Chow-Liu node removal: Kretzschmar & Stachniss' pose-graph compression):
  "tight"  the star centred at v's tightest neighbour u* (edges (u*, u_k)): the minimum spanning tree when the pair (j, k) costs
           spread(z_j) + spread(z_k) -- in samples: z'_k = z*^-1 (+) z_k, pure compositions
  "mst"    Prim over the exact pairwise marginals (log det of the conditional covariance)
  "exact"  the dense marginal (= sparse Cholesky: one Gauss-Newton step)
Prior variables are eliminated last.  Reports the RMS of the translations to the MAP after each linearise-eliminate-substitute pass.
    python scripts/relative_elimination_surrogate.py [--forms tight,mst,exact] [--passes 3] [--edges N]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scripts.tree_surrogate import load, wrap, parametric   # noqa: E402
from scripts.tree_linear_surrogate import linearise, true_optimum, init_pass, schur, idx   # noqa: E402


def eliminate(lin, n, order, form, stats):
    E = {}                                  # (a, b) a < b -> [L 6x6, h 6]
    U = {v: [np.zeros((3, 3)), np.zeros(3)] for v in range(n)}
    nb = {v: set() for v in range(n)}

    def add_edge(a, b, L, h):
        if a > b:
            P = np.r_[3:6, 0:3]
            a, b, L, h = b, a, L[np.ix_(P, P)], h[P]
        if (a, b) in E:
            E[(a, b)][0] += L; E[(a, b)][1] += h
        else:
            E[(a, b)] = [L.copy(), h.copy()]; nb[a].add(b); nb[b].add(a)

    for vs, L, h in lin:
        if len(vs) == 1:
            U[vs[0]][0] += L; U[vs[0]][1] += h
        else:
            add_edge(vs[0], vs[1], L, h)
    cond = {}
    done_order = []

    def batches():
        if order is not None:
            for v in order:
                yield [v]
            return
        # ROUNDS: independent sets of the lowest-degree variables of the CURRENT (contracted) graph; prior variables last
        alive = set(range(n))
        pri = {v for v in range(n) if np.abs(U[v][0]).max() > 0}
        while alive - pri:
            cand = sorted(alive - pri, key=lambda v: (len(nb[v]), v))
            dmin = len(nb[cand[0]])
            sel, blocked = [], set()
            for v in cand:
                if len(nb[v]) > max(2, dmin + 100):
                    break
                if v in blocked:
                    continue
                sel.append(v); blocked.add(v); blocked |= nb[v]
            stats.setdefault("rounds", []).append(len(sel))
            yield sel
            alive -= set(sel)
        yield sorted(alive)

    for v in (v for b in batches() for v in b):
        done_order.append(v)
        N_ = sorted(nb[v])
        m = len(N_)
        V = [v] + N_
        L = np.zeros((3 * (m + 1), 3 * (m + 1))); h = np.zeros(3 * (m + 1))
        L[:3, :3] += U[v][0]; h[:3] += U[v][1]
        edge_blocks = []
        for k, u in enumerate(N_):
            key = (min(u, v), max(u, v))
            Le, he = E.pop(key)
            if v > u:
                P = np.r_[3:6, 0:3]; Le, he = Le[np.ix_(P, P)], he[P]
            ii = np.r_[0:3, 3 * (k + 1):3 * (k + 2)]
            L[np.ix_(ii, ii)] += Le; h[ii] += he
            edge_blocks.append((Le.copy(), he.copy()))
            nb[u].discard(v)
        Ai = np.linalg.inv(L[:3, :3])
        cond[v] = (Ai, L[:3, 3:], h[:3], N_, edge_blocks, (U[v][0].copy(), U[v][1].copy()))
        if m == 0:
            continue
        keep = np.arange(3, 3 * (m + 1))
        Ls, hs = schur(L, h, keep, np.arange(3))
        has_unary = np.abs(U[v][0]).max() > 0
        if m == 1 or form == "exact" or has_unary or len(done_order) > n - stats.get("exact_core", 0):
            # (dense; a unary on v makes the marginal absolute: kept exact here -- prior variables are eliminated last anyway)
            for j in range(m):
                U[N_[j]][0] += 0  # placeholder
            # distribute the dense marginal as unary + pair blocks
            for j in range(m):
                jj = np.arange(3 * j, 3 * j + 3)
                U[N_[j]][0] += Ls[np.ix_(jj, jj)]; U[N_[j]][1] += hs[jj]
                for k in range(j + 1, m):
                    kk = np.arange(3 * k, 3 * k + 3)
                    Lp = np.zeros((6, 6)); Lp[:3, 3:] = Ls[np.ix_(jj, kk)]; Lp[3:, :3] = Ls[np.ix_(kk, jj)]
                    if np.abs(Lp).max() > 0:
                        add_edge(N_[j], N_[k], Lp, np.zeros(6))
            continue
        pm = {}

        def pair(j, k):
            if (j, k) not in pm:
                pm[(j, k)] = schur(Ls, hs, idx([j, k]), idx([r for r in range(m) if r not in (j, k)]))
            return pm[(j, k)]
        if form == "tight":
            # tightness of v -- u_k: log det of the conditional covariance of u_k given v under the edge alone
            t = []
            for k in range(m):
                ii = np.r_[0:3, 3 * (k + 1):3 * (k + 2)]
                Lk = L[np.ix_(ii, ii)]
                t.append(-np.linalg.slogdet(Lk[3:, 3:] + 1e-12 * np.eye(3))[1])
            c = int(np.argmin(t))
            edges = [(min(c, k), max(c, k)) for k in range(m) if k != c]
        else:
            W = np.full((m, m), np.inf)
            for j in range(m):
                for k in range(j + 1, m):
                    Lp, _ = pair(j, k)
                    W[j, k] = W[k, j] = min(-np.linalg.slogdet(Lp[3:, 3:] + 1e-12 * np.eye(3))[1], -np.linalg.slogdet(Lp[:3, :3] + 1e-12 * np.eye(3))[1])
            intree, edges = {0}, []
            while len(intree) < m:
                best = min(((W[j, k], j, k) for j in intree for k in range(m) if k not in intree))
                edges.append((min(best[1], best[2]), max(best[1], best[2]))); intree.add(best[2])
        stats["approx"] += 1 if m > 2 else 0
        deg = np.zeros(m, dtype=int)
        for j, k in edges:
            Lp, hp = pair(j, k)
            add_edge(N_[j], N_[k], Lp, hp); deg[j] += 1; deg[k] += 1
        for j in range(m):
            if deg[j] > 1:
                Lj, hj = schur(Ls, hs, idx([j]), idx([r for r in range(m) if r != j]))
                U[N_[j]][0] -= (deg[j] - 1) * Lj; U[N_[j]][1] -= (deg[j] - 1) * hj
    d = np.zeros((n, 3))
    S = {}
    for v in reversed(done_order):
        Ai, Lvn, hv, N_, eb, (Uv, uv) = cond[v]
        dn = np.concatenate([d[u] for u in N_]) if N_ else np.zeros(0)
        if not stats.get("belief_down"):
            d[v] = Ai @ (hv - Lvn @ dn)
            continue
        # belief form: every neighbour's posterior BELIEF (mean, covariance) convolved through its edge, proposals multiplied -- what a
        # conv + product down pass does (the neighbours' covariances leak into the weights)
        Li, hi = Uv.copy(), uv.copy()
        for (Le, he), u in zip(eb, N_):
            # edge as a conditional of v given u: information Lvv, mean Lvv^-1 (h_v - Lvu u)
            Lvv, Lvu = Le[:3, :3], Le[:3, 3:]
            if np.linalg.matrix_rank(Lvv) < 3:
                continue
            Ci = np.linalg.inv(Lvv)
            A = -Ci @ Lvu
            mean = Ci @ he[:3] + A @ d[u]
            cov = Ci + A @ S[u] @ A.T
            I = np.linalg.inv(cov)
            Li += I; hi += I @ mean
        S[v] = np.linalg.inv(Li)
        d[v] = S[v] @ hi
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--g2o", default="tests/golden/manhattan.g2o")
    ap.add_argument("--edges", type=int, default=None)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--forms", default="tight,mst,exact")
    ap.add_argument("--core", type=int, default=0, help="the last CORE eliminations are exact (dense)")
    a = ap.parse_args()
    from rome_jl_amd import tree as T
    E = load(a.g2o, a.edges)
    n = 1 + max(max(i, j) for i, j, _, _ in E)
    prior = (np.zeros(3), np.diag([0.01, 0.01, 0.0025]))
    Xw, _ = parametric(E, n, prior)
    Xp = true_optimum(E, n, prior, Xw)
    nbr = {v: set() for v in range(n)}
    for i, j, _, _ in E:
        nbr[i].add(j); nbr[j].add(i)
    order = T.min_degree_order(list(range(n)), nbr, last=(0,))
    rms = lambda X: np.sqrt(np.mean(np.sum((X[:, :2] - Xp[:, :2]) ** 2, axis=1)))   # noqa: E731
    X0 = init_pass(E, n, prior)
    print("start (init pass): RMS %.3f m" % rms(X0))
    for form in a.forms.split(","):
        X = X0.copy()
        for ps in range(a.passes):
            stats = {"approx": 0, "belief_down": form.endswith("B"), "exact_core": a.core}
            form_ = form.rstrip("B")
            dyn = form_.endswith("R"); form_ = form_.rstrip("R")
            t0 = time.time()
            d = eliminate(linearise(E, X, prior), n, None if dyn else order, form_, stats)
            if dyn:
                print("         rounds:", len(stats["rounds"]), stats["rounds"][:12], "...")
            X = X + d; X[:, 2] = wrap(X[:, 2])
            print("  %-6s pass %d: RMS %.3f m  (max |d| %.2f, %d approximated eliminations, %.1fs)" % (form, ps, rms(X), np.abs(d[:, :2]).max(), stats["approx"], time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
