import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, rome_jl_amd as R
from rome_jl_amd.tree import TreeSolver
d = np.load("/root/repo/tests/golden/manhattan500_reference_solve.npz")
ref = np.ascontiguousarray(d["particles"].astype(np.float64).transpose(0, 2, 1)); V, _, N = ref.shape
fg = R.initfg(N)
for k in range(V): fg.addVariable("x%d" % k, R.Pose2)
fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(d["prior_mu"], d["prior_cov"])))
for (i, j), m, c in zip(d["edges"], d["mu"], d["cov"]): fg.addFactor(["x%d" % i, "x%d" % j], R.Pose2Pose2(R.MvNormal(m, c)))
fgp = R.initfg(N); fgp.variables, fgp.factors = fg.variables, fg.factors
xp = R.solveGraphParametric(R.dead_reckon_init(fgp, seed=1)); X = np.array([xp["x%d" % k] for k in range(V)])
rms = lambda M: float(np.sqrt(np.mean(np.sum((M[:, :2] - X[:, :2]) ** 2, axis=1))))
print("reference ppe.mean vs parametric: %.3f m; ppe.suggested %.3f" % (rms(d["ppe"][:, 2]), rms(d["ppe"][:, 0])))
fg.vals = {}; R.initAllOrdered(fg, seed=3)
m, _ = R.belief_stats(np.stack([fg.getVal("x%d" % k) for k in range(V)])); print("initAllOrdered: %.3f m" % rms(m))
for msg in ("relative", "marginal"):
    fg2 = R.initfg(N); fg2.variables, fg2.factors = fg.variables, fg.factors; fg2.vals = {l: v.copy() for l, v in fg.vals.items()}
    ts = TreeSolver(fg2, messages=msg); print(ts.tree.summary()); ts.upload(); out = []
    for ps in range(6):
        ts.solve(R.make_opts(N=N, seed=40 + ps)); ts.download()
        m, _ = R.belief_stats(np.stack([fg2.getVal("x%d" % k) for k in range(V)])); out.append(rms(m))
    print(msg, "passes:", " ".join("%.3f" % x for x in out))
