"""Manhattan-3500 through `solveTree(messages="elimination")` (rome_jl_amd/elimination.py): build seconds, seconds per pass, and the RMS of
the pose means to the MAP (solveGraphParametric incl. its undamped polish), raw and after the best rigid alignment -- for INDEPENDENT
passes (what one pass gives, seed by seed) and for the POOLED sequence (what solve(passes=8) gives).
    python scripts/elimination_manhattan.py [--passes 8] [--structures 1,4] [--edges N] [--out gpurun_out/r06_elimination.txt]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rome_jl_amd as R   # noqa: E402
from rome_jl_amd.elimination import RelativeEliminationSolver   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=8)
ap.add_argument("--structures", default="1,4")
ap.add_argument("--edges", type=int, default=None)
ap.add_argument("--out", default=None)
a = ap.parse_args()
N = 100
G2O = os.path.join(ROOT, "tests", "golden", "manhattan.g2o")
lines = []


def say(s):
    print(s, flush=True); lines.append(s)


fgp = R.dead_reckon_init(R.loadG2o(G2O, N=N, max_edges=a.edges), seed=1)
t0 = time.perf_counter(); xp = R.solveGraphParametric(fgp); t_par = time.perf_counter() - t0
labels = list(fgp.variables)
mp = np.array([xp[l] for l in labels])
say("MAP reference (LM + undamped polish): %.2f s" % t_par)


def rms(fg):
    bel = np.stack([fg.getVal(l) for l in labels])
    m, _ = R.belief_stats(bel)
    d = m[:, :2] - mp[:, :2]
    raw = float(np.sqrt(np.mean(np.sum(d ** 2, axis=1))))
    A, B = m[:, :2] - m[:, :2].mean(0), mp[:, :2] - mp[:, :2].mean(0)
    U, _, Vt = np.linalg.svd(A.T @ B)
    Rm = U @ np.diag([1, np.sign(np.linalg.det(U @ Vt))]) @ Vt
    return raw, float(np.sqrt(np.mean(np.sum((A @ Rm - B) ** 2, axis=1))))


for K in [int(x) for x in a.structures.split(",")]:
    fg = R.loadG2o(G2O, N=N, max_edges=a.edges)
    t0 = time.perf_counter(); es = RelativeEliminationSolver(fg, structures=K); tb = time.perf_counter() - t0
    say("structures=%d: built in %.2f s (host structure + device plans): %s" % (K, tb, es.stats()))
    ctx = es.store.ctx
    ind, secs = [], []
    for ps in range(a.passes):
        es.reset()
        es.passes_pooled = 0
        ctx.synchronize(); t0 = time.perf_counter()
        # (independent passes walk through the structures as the pooled sequence does)
        for kind, x in es.steps[ps % K]:
            es._run(x, R.make_opts(N=N, seed=500 + ps)) if kind == "plan" else x.run()
        ctx.synchronize(); secs.append(time.perf_counter() - t0)
        es.download(fg); ind.append(rms(fg))
    say("  independent passes: %.3f s per pass (median)" % np.median(secs))
    say("    raw     " + " ".join("%.2f" % r for r, _ in ind) + "   median %.2f min %.2f max %.2f" % (np.median([r for r, _ in ind]), min(r for r, _ in ind), max(r for r, _ in ind)))
    say("    aligned " + " ".join("%.2f" % x for _, x in ind) + "   median %.2f" % np.median([x for _, x in ind]))
    es.reset()
    pooled = []
    for ps in range(a.passes):
        es.solve(R.make_opts(N=N, seed=500 + ps)); es.download(fg); pooled.append(rms(fg))
    say("  pooled sequence (solve(passes=%d)):" % a.passes)
    say("    raw     " + " ".join("%.2f" % r for r, _ in pooled))
    say("    aligned " + " ".join("%.2f" % x for _, x in pooled))
    del es
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "a") as f:
        f.write("\n".join(lines) + "\n\n")
