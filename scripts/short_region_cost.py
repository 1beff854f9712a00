#!/usr/bin/env python3
"""Where the fixed cost of a SHORT timed region goes (the driver's --steps 20: 0.2 ms between two synchronizes): host-side pieces
measured one by one on the Manhattan sweep.  Output -> profiles/r04_short_region_cost.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R

N = 100
fg = R.loadG2o(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/manhattan.g2o"), N=N)
R.dead_reckon_init(fg, seed=11)
ctx = R.Context(0)
dg = R.DeviceGraph(fg, ctx=ctx); dg.upload_beliefs(fg)
tb = dg.tab["p2p2"]
prop = torch.empty((tb["C"], 3, N), dtype=torch.float64, device="cuda")
sweep = dg.plan_sweep_pose2pose2(R.make_opts(N=N, seed=1), prop)
for _ in range(6000):
    sweep()
torch.cuda.synchronize()
pc = time.perf_counter


def med(f, n=200):
    ts = []
    for _ in range(n):
        ts.append(f())
    return 1e6 * float(np.median(ts))


def t_sync_idle():
    torch.cuda.synchronize(); a = pc(); torch.cuda.synchronize(); return pc() - a


def t_record():
    e = torch.cuda.Event(enable_timing=True); torch.cuda.synchronize(); a = pc(); e.record(); b = pc(); torch.cuda.synchronize(); return b - a


def t_launch_call():
    torch.cuda.synchronize(); a = pc(); sweep(); b = pc(); torch.cuda.synchronize(); return b - a


def t_one_sweep_synced():
    torch.cuda.synchronize(); a = pc(); sweep(); torch.cuda.synchronize(); return pc() - a


def region(K, events, poll):
    def f():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a = pc()
        if events: e0.record()
        for _ in range(K):
            sweep()
        if events or poll: e1.record()
        if poll:
            while not e1.query():
                pass
        torch.cuda.synchronize()
        return pc() - a
    return f


out = ["short timed regions of the Manhattan-3500 sweep (one launch per step), medians of 200, us:",
       "  torch.cuda.synchronize() on an idle device          %.2f" % med(t_sync_idle),
       "  Event.record() host call                            %.2f" % med(t_record),
       "  one launch: host call                               %.2f" % med(t_launch_call),
       "  one launch, synchronize to synchronize              %.2f" % med(t_one_sweep_synced)]
for K in (1, 20, 100, 1000):
    n = 200 if K <= 100 else 20
    a = med(region(K, False, False), n); b = med(region(K, True, False), n); c = med(region(K, True, True), n)
    out.append("  K = %4d steps: no events %.1f (%.2f / step) | two events %.1f (%.2f) | two events + poll %.1f (%.2f)" % (K, a, a / K, b, b / K, c, c / K))
txt = "\n".join(out)
print(txt)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/r04_short_region_cost.txt", "w").write(txt + "\n")
