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
QUICK = os.environ.get("ROME_OTHER_QUICK") == "1"   # profiling passes: few launches, no Nelder-Mead on the big tables
SOLVERS = (("closed_form", 0), ("newton", 1), ("gauss_newton", 3), ("nelder_mead", 2))
# algorithmic bytes per particle (in-kernel RNG): unique-root closed form / Newton do not read the start point u0
B_P3 = {"closed_form": 96, "newton": 96, "gauss_newton": 144, "nelder_mead": 144}
B_BR0 = {"closed_form": 40, "newton": 40, "gauss_newton": 56, "nelder_mead": 56}   # pose 24 (+ u0 16) + landmark 16
B_BR1 = {k: 64 for k in B_P3}                                                       # landmark 16 + u0 24 + pose 24
for name, sv in SOLVERS:
    if QUICK and name == "nelder_mead": continue
    reps = 2 if name == "nelder_mead" else (3 if QUICK else 20)
    o = R.make_opts(N=100, solver=sv)
    ms = timeit(lambda: dg.sweep_pose3pose3(o, out=out), reps)
    print("Pose3Pose3 helix: %6d convs %-12s %9.4f ms/sweep  %.3e conv/s  %7.1f GB/s algorithmic (%d B/particle) = %.2f of 8 TB/s" % (tb["C"], name, ms, tb["C"] / ms * 1e3, tb["C_rel"] * 100 * B_P3[name] / ms / 1e6, B_P3[name], tb["C_rel"] * 100 * B_P3[name] / ms / 1e6 / 8000))
del dg
fg = R.synth_mit_br(P=8080, n_landmarks=2000, N=100); R.dead_reckon_init(fg, seed=4)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
F = dg.tab["br"]["F"]
for d in (0, 1):
    for name, sv in SOLVERS:
        reps = 5 if name == "nelder_mead" else (3 if QUICK else 50)
        o = R.make_opts(N=100, solver=sv)
        ms = timeit(lambda: dg.sweep_bearingrange(o, d), reps)
        B = (B_BR0 if d == 0 else B_BR1)[name]
        print("BearingRange dir %d: %6d convs %-12s %9.4f ms/sweep  %.3e conv/s  %7.1f GB/s algorithmic (%d B/particle) = %.2f of 8 TB/s" % (d, F, name, ms, F / ms * 1e3, F * 100 * B / ms / 1e6, B, F * 100 * B / ms / 1e6 / 8000))

# ---- the whole sweep of the MIT-like graph: three per-family launches vs ONE fused launch (rome_sweep_pose2_dev) ----
o = R.make_opts(N=100, solver=1)
C2, Fb, Fb0 = dg.tab["p2p2"]["C"], dg.tab["br"]["F"], dg.tab["br"]["F0"]
a2 = torch.empty((C2, 3, 100), dtype=torch.float64, device="cuda"); a1 = torch.empty((Fb, 3, 100), dtype=torch.float64, device="cuda")
a0 = torch.empty((Fb0, 2, 100), dtype=torch.float64, device="cuda")
def three():
    dg.sweep_pose2pose2(o, out=a2); dg.sweep_bearingrange(o, 1, out=a1); dg.sweep_bearingrange(o, 0, out=a0)
alg = dg.tab["p2p2"]["C_rel"] * 100 * 48 + dg.tab["p2p2"]["P"] * 2400 + Fb * 100 * 64 + Fb0 * 100 * 40
og = R.make_opts(N=100, solver=3)
def three_gn():
    dg.sweep_pose2pose2(og, out=a2); dg.sweep_bearingrange(og, 1, out=a1); dg.sweep_bearingrange(og, 0, out=a0)
alg_gn = dg.tab["p2p2"]["C_rel"] * 100 * 72 + dg.tab["p2p2"]["P"] * 2400 + Fb * 100 * 64 + Fb0 * 100 * 56
for name, fn in (("GAUSS_NEWTON three launches (per family)", three_gn), ("GAUSS_NEWTON ONE fused launch (k_sweep_fused<GN>)", lambda: dg.sweep_graph_pose2(og, a2, a1, a0))):
    ms = timeit(fn, 3 if QUICK else 50)
    n = C2 + Fb + Fb0
    print("MIT-like graph sweep: %6d convs %-52s %9.4f ms/sweep  %.3e conv/s  %7.1f GB/s algorithmic = %.2f of 8 TB/s" % (n, name, ms, n / ms * 1e3, alg_gn / ms / 1e6, alg_gn / ms / 1e6 / 8000))
for name, fn in (("three launches (per family)", three), ("ONE fused launch (k_sweep_fused)", lambda: dg.sweep_graph_pose2(o, a2, a1, a0))):
    ms = timeit(fn, 3 if QUICK else 50)
    n = C2 + Fb + Fb0
    print("MIT-like graph sweep: %6d convs (%d p2p2 + %d br->pose + %d br->landmark) %-34s %9.4f ms/sweep  %.3e conv/s  %7.1f GB/s algorithmic = %.2f of 8 TB/s"
          % (n, C2, Fb, Fb0, name, ms, n / ms * 1e3, alg / ms / 1e6, alg / ms / 1e6 / 8000))
