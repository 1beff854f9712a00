#!/usr/bin/env python3
"""Scaling sets of SURVEY §8(d)-6: Pose2Pose2 sweeps over F ∈ {5453, 2^16, 2^20} factors, N=100, on one GPU.
Prints per-sweep time, convolutions/s and algorithmic GB/s for the closed-form, Newton and Gauss-Newton (functor-iterating) solvers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R

SIZES = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (5453, 1 << 16, 1 << 20)
for F in SIZES:
    pk, bel = R.synth_pose2_tables(F)
    dg = R.DeviceGraph(pk)
    dg.bel[R.Pose2].copy_(torch.as_tensor(bel))
    tb = dg.tab["p2p2"]
    out = dg.prop[R.Pose2][:tb["C"]]
    for name, sv in (("closed_form", R.SOLVER_CLOSED_FORM), ("newton", R.SOLVER_NEWTON), ("gauss_newton", R.SOLVER_GAUSS_NEWTON)):
        plan = dg.plan_sweep_pose2pose2(R.make_opts(N=100, solver=sv), out)
        plan(); torch.cuda.synchronize()
        t0 = time.perf_counter(); plan(); torch.cuda.synchronize(); one = max(time.perf_counter() - t0, 1e-6)
        for _ in range(min(5000, int(0.15 / one))):   # steady state: ~0.15 s of back-to-back launches before timing
            plan()
        torch.cuda.synchronize()
        reps = 200 if F < (1 << 16) else (50 if F < (1 << 20) else 10)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): plan()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        # closed form / Newton: fixed 24 + proposal 24 bytes per particle (u0 not read); Gauss-Newton also reads the start points (+24)
        alg = tb["C_rel"] * 100 * (72 if name == "gauss_newton" else 48) + tb["P"] * 100 * 24
        print("F=%8d poses=%8d convs=%8d %-12s %9.3f ms/sweep  %.3e conv/s  %7.1f GB/s algorithmic  (store %.1f MB, proposals %.1f MB)"
              % (F, bel.shape[0], tb["C"], name, ms, tb["C"] / ms * 1e3, alg / ms / 1e6, bel.nbytes / 1e6, out.numel() * 8 / 1e6))
    del dg
    torch.cuda.empty_cache()
