import sys
sys.path.insert(0, "/root/repo")
import numpy as np, rome_jl_amd as R
from rome_jl_amd.tree import TreeSolver
HEX = {"x0": (0, 0), "x1": (10, 0), "x2": (15, 8.66), "x3": (10, 17.32), "x4": (0, 17.32), "x5": (-5, 8.66), "x6": (0, 0), "l1": (20, 0)}
for kw in (dict(messages="marginal"), dict(messages="relative", refineIters=0, rootIters=0), dict(messages="relative", refineIters=1, rootIters=0), dict(messages="relative", refineIters=0, rootIters=1),
           dict(messages="relative", refineIters=1, rootIters=1), dict(messages="relative", refineIters=2, rootIters=0), dict(messages="relative", refineIters=3, rootIters=3)):
    worst = []
    for seed in range(5):
        fg = R.generateGraph_Hexagonal(N=100)
        R.initAllOrdered(fg, seed=4 + seed)
        ts = TreeSolver(fg, **kw); ts.upload(); ts.solve(R.make_opts(N=100, seed=31 + seed)); ts.download()
        fr = {l: float(np.mean((np.abs(fg.getVal(l)[0] - x) < 3.0) & (np.abs(fg.getVal(l)[1] - y) < 3.0))) for l, (x, y) in HEX.items()}
        worst.append(min(fr.values()))
    print(kw, "min window fraction over variables, 5 seeds:", np.round(worst, 2))
