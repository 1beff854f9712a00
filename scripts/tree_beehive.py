import sys
sys.path.insert(0, "/root/repo")
import numpy as np, rome_jl_amd as R
from rome_jl_amd.tree import TreeSolver
for kw in (dict(messages="relative", rootIters=0, refineIters=0), dict(messages="relative"), dict(messages="relative", rootIters=1, refineIters=0), dict(messages="relative", rootIters=0, refineIters=1), dict(messages="marginal")):
    out = []
    for seed in range(4):
        fg = R.synth_beehive_mh(20, N=100)
        R.initAllOrdered(fg, seed=2 + seed)
        sim = fg._sim
        e0 = np.median([np.hypot(*(fg.getVal(l)[:2].mean(1) - np.asarray(sim[l])[:2])) for l in fg.variables if fg.variables[l] is R.Pose2])
        ts = TreeSolver(fg, **kw); ts.upload(); ts.solve(R.make_opts(N=100, seed=3 + seed)); ts.download()
        err = [np.hypot(*(fg.getVal(l)[:2].mean(1) - np.asarray(sim[l])[:2])) for l in fg.variables if fg.variables[l] is R.Pose2]
        out.append((round(float(e0), 2), round(float(np.median(err)), 2)))
    print(kw, "(init median err, after) per seed:", out)
