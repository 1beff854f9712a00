"""Manhattan-3500 through the Bayes-tree solve, message structures side by side: per pass the RMS of the pose means to the MAP
(solveGraphParametric with its undamped polish steps), raw and after the best rigid alignment, and the seconds per pass.
    python scripts/tree_forms.py [--forms star,hop] [--passes 8] [--seeds 1,2] [--edges N] [--out gpurun_out/r06_tree_forms.txt]"""
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
ap.add_argument("--forms", default="star,hop")
ap.add_argument("--passes", type=int, default=8)
ap.add_argument("--seeds", default="1")
ap.add_argument("--edges", type=int, default=None)
ap.add_argument("--fresh", action="store_true", help="every pass from the init beliefs (independent passes)")
ap.add_argument("--kw", default="", help="extra TreeSolver keywords, k=v,k=v (ints)")
ap.add_argument("--out", default=None)
a = ap.parse_args()
N = 100
G2O = os.path.join(ROOT, "tests", "golden", "manhattan.g2o")
lines = []


def say(s):
    print(s, flush=True); lines.append(s)


fgp = R.dead_reckon_init(R.loadG2o(G2O, N=N, max_edges=a.edges), seed=1)
t0 = time.perf_counter(); x0 = R.solveGraphParametric(fgp, polish=0); t_a = time.perf_counter() - t0
t0 = time.perf_counter(); xp = R.solveGraphParametric(fgp); t_b = time.perf_counter() - t0
labels = list(fgp.variables)
mp = np.array([xp[l] for l in labels]); m0 = np.array([x0[l] for l in labels])
say("parametric reference: damped LM only %.2f s, with the undamped polish %.2f s; the two differ by %.3f m RMS"
    % (t_a, t_b, np.sqrt(np.mean(np.sum((mp[:, :2] - m0[:, :2]) ** 2, axis=1)))))


def rms(fg):
    bel = np.stack([fg.getVal(l) for l in labels])
    m, _ = R.belief_stats(bel)
    d = m[:, :2] - mp[:, :2]
    raw = float(np.sqrt(np.mean(np.sum(d ** 2, axis=1))))
    A, B = m[:, :2] - m[:, :2].mean(0), mp[:, :2] - mp[:, :2].mean(0)     # best rigid alignment (Kabsch)
    U, _, Vt = np.linalg.svd(A.T @ B)
    Rm = U @ np.diag([1, np.sign(np.linalg.det(U @ Vt))]) @ Vt
    al = float(np.sqrt(np.mean(np.sum((A @ Rm - B) ** 2, axis=1))))
    return raw, al


kw = {k: int(v) for k, v in (kv.split("=") for kv in a.kw.split(",") if kv)}
for seed in [int(x) for x in a.seeds.split(",")]:
    fg = R.loadG2o(G2O, N=N, max_edges=a.edges)
    R.initAllOrdered(fg, seed=seed)
    init_vals = {l: fg.getVal(l).copy() for l in labels}
    say("seed %d: initAllOrdered -> RMS %.3f m raw, %.3f m aligned" % ((seed,) + rms(fg)))
    for form in a.forms.split(","):
        for l in labels:
            fg.vals[l] = init_vals[l].copy()
        t0 = time.perf_counter()
        ts = TreeSolver(fg, messages="relative" if form != "marginal" else "marginal", **({"message_tree": form} if form != "marginal" else {}), **kw)
        tb = time.perf_counter() - t0
        st = ts.stats()
        ts.upload()
        ctx = ts.store.ctx
        raws, als, secs = [], [], []
        for ps in range(a.passes):
            if a.fresh and ps:
                for l in labels:
                    fg.vals[l] = init_vals[l].copy()
                ts.upload()
            o = R.make_opts(N=N, seed=1000 * seed + ps)
            ctx.synchronize(); t0 = time.perf_counter(); ts.solve(o); ctx.synchronize(); secs.append(time.perf_counter() - t0)
            ts.download()
            r_, a_ = rms(fg); raws.append(r_); als.append(a_)
        say("  %-8s build %.2f s, up/down steps %d/%d, rows %d/%d, %d blocks; s/pass median %.3f" % (form, tb, st["up_steps"], st["down_steps"], st["up_rows"], st["down_rows"], st["blocks"], np.median(secs)))
        say("           raw     " + " ".join("%.2f" % x for x in raws) + "   median %.2f min %.2f max %.2f" % (np.median(raws), min(raws), max(raws)))
        say("           aligned " + " ".join("%.2f" % x for x in als) + "   median %.2f" % np.median(als))
        del ts
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "a") as f:
        f.write("\n".join(lines) + "\n\n")
