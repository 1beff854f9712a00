#!/usr/bin/env python3
"""Counterpart of the reference's examples/ManhattanDatasetBatch.jl (BASELINE configs[1]) on the M3500 dataset: load the g2o file,
anchor x0, solve parametrically (batched residual/Jacobian kernel + sparse LM), dress the solution with non-parametric beliefs
(convolution sweeps + proposal products on the GPU) and write the estimates back as a g2o file with VERTEX_SE2 records.

    python examples/manhattan_batch.py [path/to/file.g2o] [out.g2o]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rome_jl_amd as R  # noqa: E402

argv = [a for a in sys.argv[1:] if a != "--tree"]
path = argv[0] if len(argv) > 0 else os.path.join(ROOT, "tests", "golden", "manhattan.g2o")
out = argv[1] if len(argv) > 1 else "/tmp/manhattan_solved.g2o"
fg = R.loadG2o(path, N=100)                                   # x0 + PriorPose2(N(0, diag(0.1², 0.1², 0.05²))) + every EDGE_SE2
if "--tree" in sys.argv:     # the reference's own call sequence (examples/ManhattanDatasetBatch.jl:43): tree = solveTree!(fg) -- no parametric start
    t = time.perf_counter()
    ts_ = R.solveTree(fg, seed=11)   # (messages="auto": a Pose2Pose2 / PriorPose2 graph takes the elimination form -- no init pass)
    tt = time.perf_counter() - t
    labels = sorted(fg.variables, key=lambda s: int(s[1:]))
    mean, std = R.belief_stats(np.stack([fg.getVal(l) for l in labels]))
    print("%d poses, %d factors: solveTree (%s) %.2f s wall-clock" % (len(labels), len(fg.factors), ts_.stats(), tt))
    R.exportG2o(fg, filename=out, estimates={l: mean[k] for k, l in enumerate(labels)}, varIntLabel={l: int(l[1:]) for l in labels})
    print("wrote", out)
    sys.exit(0)
R.dead_reckon_init(fg, seed=1)
t = time.perf_counter(); xp = R.solveGraphParametric(fg); tp = time.perf_counter() - t
dg = R.DeviceGraph(fg)
dg.init_from_means(xp)
torch.cuda.synchronize(); t = time.perf_counter()
dg.solve(R.make_opts(N=100, solver=R.SOLVER_NEWTON, seed=11), n_sweeps=10)
torch.cuda.synchronize(); ts = time.perf_counter() - t
mean, std = dg.belief_stats(R.Pose2)
mean, std = mean.cpu().numpy(), std.cpu().numpy()
labels = dg.packed.labels[R.Pose2]
print("%d poses, %d factors: parametric solve %.2f s, 10 non-parametric sweeps %.1f ms" % (len(labels), len(fg.factors), tp, 1e3 * ts))
print("map extent x [%.1f, %.1f]  y [%.1f, %.1f];  mean belief std (%.3f, %.3f, %.4f)" %
      (mean[:, 0].min(), mean[:, 0].max(), mean[:, 1].min(), mean[:, 1].max(), *std.mean(axis=0)))
order = sorted(labels, key=lambda s: int(s[1:]))
R.exportG2o(fg, filename=out, estimates={l: mean[labels.index(l)] for l in order}, varIntLabel={l: int(l[1:]) for l in order})
print("wrote", out)
