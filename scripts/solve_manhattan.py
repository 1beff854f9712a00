#!/usr/bin/env python3
"""Whole-graph nonparametric solve of the synthetic Manhattan graph on the GPU: wall-clock + accuracy."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R

def kabsch_rms(A, B):
    ca, cb = A.mean(0), B.mean(0)
    U, _, Vt = np.linalg.svd((A - ca).T @ (B - cb))
    Rm = (U @ np.diag([1, np.sign(np.linalg.det(U @ Vt))]) @ Vt).T
    return np.sqrt(np.mean(np.sum(((A - ca) @ Rm.T + cb - B) ** 2, axis=1)))

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 1954
fg = R.synth_manhattan(P=P, loops=loops)
R.dead_reckon_init(fg, seed=1)
gt = np.array([fg.ground_truth["x%d" % k][:2] for k in range(P)])
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=100, solver=1, seed=11)
m, _ = dg.belief_stats(R.Pose2); print("init rms", kabsch_rms(m.cpu().numpy()[:, :2], gt))
tot = 0
for chunk in (5, 5, 10, 20, 40, 80):
    torch.cuda.synchronize(); t = time.perf_counter()
    for s in range(chunk):
        dg.conv_step(o, tot + s); dg.product_step(o, tot + s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    tot += chunk
    m, sd = dg.belief_stats(R.Pose2)
    print("sweeps %3d  %.2f ms/sweep  rms %.3f  mean std %s" % (tot, 1e3 * dt / chunk, kabsch_rms(m.cpu().numpy()[:, :2], gt), sd.mean(0).cpu().numpy().round(3)))

# ---- pipeline as IIF allows it: parametric solve first, nonparametric sweeps initialised from it ----
t = time.perf_counter(); xp = R.solveGraphParametric(fg); tp = time.perf_counter() - t
mp = np.array([xp["x%d" % k][:2] for k in range(P)])
print("parametric solve: %.2f s, rms %.3f" % (tp, kabsch_rms(mp, gt)))
dg.init_from_means(xp)
torch.cuda.synchronize(); t = time.perf_counter()
dg.solve(o, n_sweeps=10)
torch.cuda.synchronize(); dt = time.perf_counter() - t
m, sd = dg.belief_stats(R.Pose2)
print("parametric init + 10 sweeps: %.2f ms, rms %.3f, mean std %s" % (1e3 * dt, kabsch_rms(m.cpu().numpy()[:, :2], gt), sd.mean(0).cpu().numpy().round(3)))
