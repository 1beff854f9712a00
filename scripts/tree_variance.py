import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import rome_jl_amd as R
from rome_jl_amd.tree import TreeSolver
N = 100
G2O = "/root/repo/tests/golden/manhattan.g2o"
fg = R.loadG2o(G2O, N=N)
xp = R.solveGraphParametric(R.dead_reckon_init(R.loadG2o(G2O, N=N), seed=1))
labels = list(fg.variables); mp = np.array([xp[l] for l in labels])
if len(sys.argv) > 4 and sys.argv[4] == "tight":      # experiment: a prior that pins the gauge (what remains is not gauge noise)
    for k, (fl, ls, f) in enumerate(fg.factors):
        if isinstance(f, R.PriorPose2):
            fg.factors[k] = (fl, ls, R.PriorPose2(R.MvNormal(f.Z.mu, np.diag([1e-6, 1e-6, 1e-6])))); fg._findex[fl] = fg.factors[k]
    print("tight prior")
if len(sys.argv) > 4 and sys.argv[4] == "loose":      # experiment: the prior carries (almost) nothing -- the solve fixes the map up to a rigid gauge
    for k, (fl, ls, f) in enumerate(fg.factors):
        if isinstance(f, R.PriorPose2):
            fg.factors[k] = (fl, ls, R.PriorPose2(R.MvNormal(f.Z.mu, np.diag([1e2, 1e2, 1e0])))); fg._findex[fl] = fg.factors[k]
    print("loose prior")
R.initAllOrdered(fg, seed=1)
ts = TreeSolver(fg, messages="relative", rootIters=int(sys.argv[1]), refineIters=int(sys.argv[2]), relIters=int(os.environ.get("REL", "0")), max_product=int(os.environ.get("MAXPROD", "8")), last=(("x0",) if len(sys.argv) > 3 and sys.argv[3] == "last" else ()))
print(ts.tree.summary())
print("rootIters", sys.argv[1], "refineIters", sys.argv[2])
root = [v for c in ts.tree.cliques if c.parent < 0 for v in c.frontals]
ridx = [labels.index(v) for v in root]
lvl_of = {v: c.level for c in ts.tree.cliques for v in c.frontals}
lv = np.array([lvl_of[l] for l in labels])
from rome_jl_amd.parametric import _Problem
Pb = _Problem(fg)
def map_cost(m):
    r, _ = Pb.linearize(Pb.pack({l: m[k] for k, l in enumerate(labels)}))
    return float(r @ r)
print("cost of the parametric solution %.1f" % map_cost(mp))
init_vals = {l: v.copy() for l, v in fg.vals.items()}
ts.upload()
import time as _t
acc = None
for ps in range(int(os.environ.get('PASSES', '6'))):
    if os.environ.get("FRESH") == "1":
        fg.vals = {l: v.copy() for l, v in init_vals.items()}; ts.upload()      # every pass from the SAME init beliefs: independent estimates
    ts.store.ctx.synchronize(); _t0 = _t.perf_counter(); ts.solve(R.make_opts(N=N, seed=500 + ps)); ts.store.ctx.synchronize(); print("   pass %.4f s" % (_t.perf_counter() - _t0)); ts.download()
    bel = np.stack([fg.getVal(l) for l in labels]); m, _ = R.belief_stats(bel)
    e = np.sqrt(np.sum((m[:, :2] - mp[:, :2]) ** 2, axis=1))
    if len(sys.argv) > 4 and sys.argv[4] == "loose":   # gauge fix: the rigid transform that puts the estimate of x0 on the prior mean
        k0 = labels.index("x0"); th = -m[k0, 2]; c_, s_ = np.cos(th), np.sin(th)
        d = m[:, :2] - m[k0, :2]
        m = m.copy(); m[:, 0] = c_ * d[:, 0] - s_ * d[:, 1]; m[:, 1] = s_ * d[:, 0] + c_ * d[:, 1]
        e = np.sqrt(np.sum((m[:, :2] - mp[:, :2]) ** 2, axis=1))
        print("   after the gauge fix at x0: RMS %.3f" % np.sqrt(np.mean(e**2)))
    acc = m[:, :2].copy() if acc is None else acc + m[:, :2]
    print("   mean of the pose means over %d passes: RMS %.3f" % (ps + 1, np.sqrt(np.mean(np.sum((acc / (ps + 1) - mp[:, :2]) ** 2, axis=1)))))
    A, Bm = m[:, :2] - m[:, :2].mean(0), mp[:, :2] - mp[:, :2].mean(0)
    Uu, _, Vt = np.linalg.svd(A.T @ Bm); Rr = (Uu @ Vt).T
    if np.linalg.det(Rr) < 0: Rr = (Uu @ np.diag([1, -1]) @ Vt).T
    al = np.sqrt(np.mean(np.sum((A @ Rr.T - Bm) ** 2, axis=1)))
    print("   MAP cost of the pose means: %.3e" % map_cost(m))
    print("pass %d: RMS all %.3f aligned %.3f root %.3f | by level 39..30: %s | lvl 20 %.2f lvl 10 %.2f lvl 0 %.2f" % (ps, np.sqrt(np.mean(e**2)), al, np.sqrt(np.mean(e[ridx]**2)),
          " ".join("%.2f" % np.sqrt(np.mean(e[lv == h]**2)) for h in range(39, 29, -1)), np.sqrt(np.mean(e[lv == 20]**2)), np.sqrt(np.mean(e[lv == 10]**2)), np.sqrt(np.mean(e[lv == 0]**2))), flush=True)
