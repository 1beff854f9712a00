#!/usr/bin/env python3
"""wall time of the Gibbs product alone (trees + order + sampling launches) on one Manhattan sweep's proposals"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, rome_jl_amd as R
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
fg = R.loadG2o(os.path.join(root, "tests/golden/manhattan.g2o"), N=100); R.dead_reckon_init(fg, seed=1)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=100, solver=1, seed=11)
dg.conv_step(o, 0)
def run(n, **kw):
    torch.cuda.synchronize(); t = time.perf_counter()
    for s in range(n): dg.product_step(o, s, **kw)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for kw in (dict(bandwidth="lcv", product="importance"), dict(bandwidth="lcv", product="gibbs")):
    run(3, **kw); print(kw, "%.3f ms" % run(20, **kw))
