#!/usr/bin/env python3
"""Sweep timings of the Pose3Pose3 (helix, 10k poses) and Pose2Point2BearingRange (MIT-like) kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); one = max(time.perf_counter() - t0, 1e-6)
    for _ in range(min(5000, int(0.15 / one))):   # steady state: ~0.15 s of back-to-back launches before timing (queued, then drained)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

fg = R.synth_helix3d(P=10000, N=100); R.dead_reckon_init_pose3(fg, seed=2)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
tb = dg.tab["p3p3"]; out = torch.empty((tb["C"], 6, 100), dtype=torch.float64, device="cuda")
for name, sv, reps in (("closed_form", 0, 20), ("newton", 1, 20), ("nelder_mead", 2, 2)):
    o = R.make_opts(N=100, solver=sv)
    ms = timeit(lambda: dg.sweep_pose3pose3(o, out=out), reps)
    print("Pose3Pose3 helix: %6d convs %-11s %9.3f ms/sweep  %.3e conv/s  %7.1f GB/s algorithmic" % (tb["C"], name, ms, tb["C"] / ms * 1e3, tb["C_rel"] * 100 * 144 / ms / 1e6))
del dg
fg = R.synth_mit_br(P=8080, n_landmarks=2000, N=100); R.dead_reckon_init(fg, seed=4)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
F = dg.tab["br"]["F"]
for d in (0, 1):
    for name, sv, reps in (("closed_form", 0, 50), ("newton", 1, 50), ("nelder_mead", 2, 5)):
        o = R.make_opts(N=100, solver=sv)
        ms = timeit(lambda: dg.sweep_bearingrange(o, d), reps)
        print("BearingRange dir %d: %6d convs %-11s %9.3f ms/sweep  %.3e conv/s" % (d, F, name, ms, F / ms * 1e3))
