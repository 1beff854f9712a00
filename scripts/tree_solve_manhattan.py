"""Manhattan-3500 through the Bayes-tree solve from initAll (no dead reckoning, no parametric start): RMS of the pose means to the
parametric solution per pass, wall-clock per stage, frontier-width histogram and ms per level.
    python scripts/tree_solve_manhattan.py [--messages relative|marginal] [--passes 3] [--edges N] [--out gpurun_out/r05_tree_solve.txt]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rome_jl_amd as R   # noqa: E402
from rome_jl_amd.tree import TreeSolver   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--messages", default="relative")
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--edges", type=int, default=None)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--out", default=None)
ap.add_argument("--levels", action="store_true", help="time every level (synchronises after each: slower than the solve itself)")
a = ap.parse_args()
N = 100
G2O = os.path.join(ROOT, "tests", "golden", "manhattan.g2o")
fg = R.loadG2o(G2O, N=N, max_edges=a.edges)
t0 = time.perf_counter(); xp = R.solveGraphParametric(R.dead_reckon_init(R.loadG2o(G2O, N=N, max_edges=a.edges), seed=1)); t_par = time.perf_counter() - t0
labels = list(fg.variables)
mp = np.array([xp[l] for l in labels])
lines = []


def say(s):
    print(s, flush=True); lines.append(s)


def rms(get):
    bel = np.stack([get(l) for l in labels])
    m, _ = R.belief_stats(bel)
    return float(np.sqrt(np.mean(np.sum((m[:, :2] - mp[:, :2]) ** 2, axis=1)))), float(np.median(bel[:, :2].std(axis=2)))


t0 = time.perf_counter(); R.initAllOrdered(fg, seed=a.seed); t_init = time.perf_counter() - t0
r, s = rms(fg.getVal)
say("Manhattan (%d poses, %d factors), N=%d; parametric reference %.2f s" % (len(labels), len(fg.factors), N, t_par))
say("initAllOrdered (host plans + device init pass): %.2f s -> RMS to the parametric solution %.3f m, median belief std %.3f m" % (t_init, r, s))
t0 = time.perf_counter(); ts = TreeSolver(fg, messages=a.messages); t_build = time.perf_counter() - t0
say(ts.tree.summary())
say("tree + level plans built on the host in %.2f s: %s" % (t_build, ts.stats()))
w = np.array([len(l) for l in ts.tree.levels])
say("frontier width by level (leaves first): " + " ".join(str(x) for x in w))
ctx = ts.store.ctx
ts.upload()
for ps in range(a.passes):
    o = R.make_opts(N=N, seed=100 + ps)
    ctx.synchronize(); t0 = time.perf_counter(); ts.up(o); ctx.synchronize(); tu = time.perf_counter() - t0
    t0 = time.perf_counter(); ts.down(o); ctx.synchronize(); td = time.perf_counter() - t0
    ts.download()
    r, s = rms(fg.getVal)
    say("pass %d (%s messages): up %.3f s + down %.3f s = %.3f s -> RMS %.3f m, median belief std %.3f m" % (ps, a.messages, tu, td, tu + td, r, s))
if a.levels:
    o = R.make_opts(N=N, seed=999)
    say("per level (up pass, synchronised after every level): width, update steps, rows, ms")
    for h, (pre, pl, post, sp) in enumerate(zip(ts.up_pre, ts.up_plans, ts.up_post, ts.up_specs)):
        ctx.synchronize(); t0 = time.perf_counter()
        for op in pre:
            op.run()
        if pl is not None:
            ts._run(pl, o)
        for op in post:
            op.run()
        ctx.synchronize()
        say("  level %2d: width %4d, steps %2d, rows %5d, %.3f ms" % (h, w[h], len(set(sp.groups)) * sp.gibbs_iters, len(sp.pairs), 1e3 * (time.perf_counter() - t0)))
if a.levels:
    say("per level (down pass, root first): width, update steps, rows, largest product (proposals), ms")
    for h in range(len(ts.down_plans) - 1, -1, -1):
        pl, sp = ts.down_plans[h], ts.down_specs[h]
        if pl is None:
            continue
        kmax = max((sum(1 for p in sp.pairs if p[1] == l) + sum(1 for m in sp.smsgs if m[1] == l)) for l in sp.order)
        ctx.synchronize(); t0 = time.perf_counter()
        ts._run(pl, o)
        ctx.synchronize()
        say("  level %2d: width %4d, steps %2d, rows %5d, largest product %3d, %.3f ms" % (h, w[h], len(set(sp.groups)) * sp.gibbs_iters, len(sp.pairs), kmax, 1e3 * (time.perf_counter() - t0)))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "a") as f:
        f.write("\n".join(lines) + "\n\n")
