#!/usr/bin/env python3
"""Latency / throughput of the host-pointer entry point rome_conv_pose2pose2 (what the Julia shim calls per factor)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rome_jl_amd as R

rng = np.random.default_rng(0)
N = 100
for C_ in (1, 8, 64, 512, 4096):
    mu = np.tile([10.0, 0, np.pi / 3], (C_, 1)); cov = np.tile(np.diag([0.01, 0.01, 0.01]), (C_, 1, 1))
    fixed = rng.standard_normal((C_, 3, N)); target = rng.standard_normal((C_, 3, N))
    dirs = np.zeros(C_, np.int32)
    for layout, name in ((R.LAYOUT_SOA, "soa"), (R.LAYOUT_AOS, "aos")):
        o = R.make_opts(N=N, solver=1, seed=1); o.layout = layout
        f = fixed if layout == R.LAYOUT_SOA else np.ascontiguousarray(fixed.transpose(0, 2, 1))
        t = target if layout == R.LAYOUT_SOA else np.ascontiguousarray(target.transpose(0, 2, 1))
        for _ in range(3):
            R.conv_pose2pose2(o, mu, cov, f, t, dirs=dirs)
        reps = max(3, min(200, 2000 // C_))
        a = time.perf_counter()
        for _ in range(reps):
            R.conv_pose2pose2(o, mu, cov, f, t, dirs=dirs)
        dt = (time.perf_counter() - a) / reps
        print("C=%5d %s: %8.1f us/call  %9.3e conv/s" % (C_, name, dt * 1e6, C_ / dt))
