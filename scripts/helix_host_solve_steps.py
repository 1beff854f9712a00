import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, scipy.sparse as sp
from scipy.sparse.linalg import splu
import rome_jl_amd as R
from rome_jl_amd import parametric as PM
fgh = R.synth_helix3d(P=10000, N=8)
R.dead_reckon_init_pose3(fgh, seed=7)
t=time.perf_counter(); P = PM._Problem(fgh); print("setup %.3f" % (time.perf_counter()-t))
# random values in the pattern
rng = np.random.default_rng(0)
vals = rng.standard_normal(len(P.csr_perm))
J = sp.csr_matrix((vals, P.csr_indices, P.csr_indptr), shape=(P.m, P.n)); J.has_sorted_indices = True
for rep in range(2):
    t=time.perf_counter(); H = (J.T @ J).tocsc(); t1=time.perf_counter(); D = sp.diags(H.diagonal() + 1e-12); Hd = (H + 1e-3 * D).tocsc(); t2=time.perf_counter()
    lu = splu(Hd, permc_spec="NATURAL", options=dict(SymmetricMode=True, DiagPivotThresh=0.0)); t3=time.perf_counter(); x = lu.solve(np.ones(P.n)); t4=time.perf_counter()
    print("JtJ %.3f  damp %.3f  splu %.3f  solve %.3f   nnz(H) %d nnz(L+U) %d" % (t1-t, t2-t1, t3-t2, t4-t3, H.nnz, lu.L.nnz + lu.U.nnz))
# banded: bandwidth in the elimination order vs natural pose order
Hn = Hd.tocoo(); bw = np.abs(Hn.row - Hn.col).max(); print("bandwidth in the MMD order", bw)
# the blockwise normal matrix on the same pattern
from rome_jl_amd import api
blocks = []
for k, g in P.groups.items():
    dz, dr, da, db = api._LIN_DIMS[k]
    F = len(g["a"])
    blocks.append((rng.standard_normal((F, dr, da)), rng.standard_normal((F, dr, db)) if db else None))
for rep in range(3):
    t = time.perf_counter(); H = P.normal_matrix(blocks); t1 = time.perf_counter(); Hd = P.damped(H, 1e-3); t2 = time.perf_counter()
    lu = splu(Hd, permc_spec="NATURAL", options=dict(SymmetricMode=True, DiagPivotThresh=0.0)); t3 = time.perf_counter()
    print("blockwise H %.4f  damp %.4f  splu %.3f" % (t1 - t, t2 - t1, t3 - t2))
for spec in ("MMD_AT_PLUS_A", "COLAMD"):
    t = time.perf_counter(); lu = splu(Hd, permc_spec=spec, options=dict(SymmetricMode=True, DiagPivotThresh=0.0)); print(spec, "splu %.3f nnz(L+U) %d" % (time.perf_counter() - t, lu.L.nnz + lu.U.nnz))
import scipy.sparse.csgraph as cg
t = time.perf_counter(); pr = cg.reverse_cuthill_mckee(Hd.tocsr(), symmetric_mode=True); Hr = Hd[pr][:, pr].tocsc(); c = Hr.tocoo(); print("RCM %.3f s bandwidth %d" % (time.perf_counter() - t, np.abs(c.row - c.col).max()))
t = time.perf_counter(); lu = splu(Hr, permc_spec="NATURAL", options=dict(SymmetricMode=True, DiagPivotThresh=0.0)); print("RCM order splu %.3f nnz(L+U) %d" % (time.perf_counter() - t, lu.L.nnz + lu.U.nnz))
from scipy.linalg import cholesky_banded, cho_solve_banded
bw = int(np.abs(c.row - c.col).max())
ab = np.zeros((bw + 1, P.n)); m_ = c.row >= c.col; ab[(c.row - c.col)[m_], c.col[m_]] = c.data[m_]
t = time.perf_counter(); cb = cholesky_banded(ab, lower=True, check_finite=False); t1 = time.perf_counter(); x = cho_solve_banded((cb, True), np.ones(P.n), check_finite=False); print("banded Cholesky %.3f s, solve %.3f s" % (t1 - t, time.perf_counter() - t1))
