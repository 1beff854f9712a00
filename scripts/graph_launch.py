#!/usr/bin/env python3
"""Does a hipGraph of K dependent sweeps shorten the launch period?  K sweeps of the Manhattan table captured into one graph
(torch.cuda.CUDAGraph on the stream the rome_ctx is bound to) against the same K stream launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R
fg = R.loadG2o(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "manhattan.g2o"), N=100)
R.dead_reckon_init(fg, seed=11)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
tb = dg.tab["p2p2"]
out = torch.empty((tb["C"], 3, 100), dtype=torch.float64, device="cuda")
plan = dg.plan_sweep_pose2pose2(R.make_opts(N=100, solver=R.SOLVER_NEWTON), out)
for _ in range(4000): plan()
torch.cuda.synchronize()
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for K in (20, 200):
    ms_stream = timed(plan, 2000)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(50): plan()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(K): plan()
    g.replay(); torch.cuda.synchronize()
    ms_graph = timed(g.replay, 200) / K
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); one = (time.perf_counter() - t0) * 1e3 / K
    print("K=%d: stream launches %.2f us/sweep; hipGraph replay %.2f us/sweep (single replay incl. sync: %.2f us/sweep)" % (K, 1e3 * ms_stream, 1e3 * ms_graph, 1e3 * one))
