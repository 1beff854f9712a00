import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, scipy.sparse as sp
from scipy.sparse.linalg import splu
import rome_jl_amd as R
from rome_jl_amd import parametric as PM, api
fgh = R.synth_helix3d(P=10000, N=8); R.dead_reckon_init_pose3(fgh, seed=7)
P = PM._Problem(fgh)
rng = np.random.default_rng(0)
blocks = []
for k, g in P.groups.items():
    dz, dr, da, db = api._LIN_DIMS[k]; F = len(g["a"])
    blocks.append((rng.standard_normal((F, dr, da)), rng.standard_normal((F, dr, db)) if db else None))
with PM._blas_single_thread():
    H = P.normal_matrix(blocks); Hd = P.damped(H, 1e-3); rhs = np.ones(P.n)
    base = dict(SymmetricMode=True, DiagPivotThresh=0.0)
    ref = None
    combos = [("default", dict(options=base))] + [("panel=%d relax=%d" % (pn, rl), dict(options=base, panel_size=pn, relax=rl)) for pn in (1, 2, 4, 8) for rl in (1, 4, 8)]
    for name, kw in combos:
        ts = []
        for rep in range(3):
            t = time.perf_counter(); lu = splu(Hd, permc_spec="NATURAL", **kw); t1 = time.perf_counter(); x = lu.solve(rhs); t2 = time.perf_counter(); ts.append((t1 - t, t2 - t1))
        if ref is None: ref = x
        print("%-28s factor %.4f s  solve %.4f s  nnz(L+U) %d  |x - ref| %.2e" % (name, min(a for a, _ in ts), min(b for _, b in ts), lu.L.nnz + lu.U.nnz, abs(x - ref).max()))
