"""Pose3Pose3 functor-iterating sweep (Gauss-Newton) on the 10k-pose helix: ms per sweep and fraction of the HBM roofline (144 B per particle)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R
fg = R.synth_helix3d(P=10000, N=100); R.dead_reckon_init_pose3(fg, seed=2)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
tb = dg.tab["p3p3"]; out = torch.empty((tb["C"], 6, 100), dtype=torch.float64, device="cuda")
for name, sv, B in (("closed_form", 0, 96), ("gauss_newton", 3, 144)):
    o = R.make_opts(N=100, solver=sv)
    for _ in range(200): dg.sweep_pose3pose3(o, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): dg.sweep_pose3pose3(o, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print("%s Pose3Pose3 helix: %d convs %-12s %.4f ms/sweep  %.3e conv/s  %.2f of 8 TB/s (%d B/particle)" % (os.environ.get("ROME_MI355_LIB", "default"), tb["C"], name, ms, tb["C"] / ms * 1e3, tb["C_rel"] * 100 * B / ms / 1e6 / 8000, B))
