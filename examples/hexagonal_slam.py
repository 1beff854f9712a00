#!/usr/bin/env python3
"""Counterpart of the reference's examples/Hexagonal2D_SLAM.jl (BASELINE configs[0]): a robot drives a hexagon of six 10 m legs,
sights one landmark at the start and again when it returns; all beliefs live on the GPU, the factor convolutions are the HIP path.

    python examples/hexagonal_slam.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import rome_jl_amd as R  # noqa: E402

fg = R.generateGraph_Hexagonal(N=100)               # 7 Pose2, 1 Point2, prior + 6 Pose2Pose2 + 2 Pose2Point2BearingRange
R.dead_reckon_init(fg, seed=1)                      # initAll!: beliefs propagated along the odometry
dg = R.DeviceGraph(fg)
dg.upload_beliefs(fg)
dg.solve(R.make_opts(N=100, solver=R.SOLVER_NEWTON, seed=7), n_sweeps=12)
mean2, std2 = dg.belief_stats(R.Pose2)
meanl, stdl = dg.belief_stats(R.Point2)
for label, m, s in zip(dg.packed.labels[R.Pose2], mean2.cpu().numpy(), std2.cpu().numpy()):
    print("%-3s  mean (%7.2f, %7.2f, %6.2f)   std (%.2f, %.2f, %.2f)" % (label, *m, *s))
for label, m, s in zip(dg.packed.labels[R.Point2], meanl.cpu().numpy(), stdl.cpu().numpy()):
    print("%-3s  mean (%7.2f, %7.2f)           std (%.2f, %.2f)" % (label, *m, *s))
# a single convolution through the public API, as IIF's approxConv(fg, :x0x1f1, :x1)
dg.download_beliefs(fg)
pts = R.approxConv(fg, "x0x1f1", "x1", seed=3)
print("approxConv(x0x1f1 -> x1): mean", np.round(pts.mean(axis=1), 2))
