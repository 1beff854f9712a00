#!/usr/bin/env python3
"""Run-to-convergence of the device solve loop (reference operations: manikde! bandwidths + multiscale Gibbs product) on the M3500
Manhattan graph from dead-reckoned beliefs: distance of the pose means to the parametric solution per iteration block."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R

fg = R.loadG2o(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "manhattan.g2o"), N=100)
R.dead_reckon_init(fg, seed=11)
P = 3500
t = time.perf_counter(); xp = R.solveGraphParametric(fg); tp = time.perf_counter() - t
mp = np.array([xp["x%d" % k] for k in range(P)])
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=100, solver=1, seed=11)
def dist():
    m, _ = dg.belief_stats(R.Pose2); m = m.cpu().numpy()
    d = m - mp; d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    return np.sqrt(np.mean(np.sum(d[:, :2] ** 2, 1))), np.abs(d[:, :2]).max(), m
print("parametric solve %.2f s" % tp)
r0, mx0, mprev = dist(); print("iter 0 rms-to-parametric %.3f max %.3f" % (r0, mx0))
tot = 0; t_all = 0
product = sys.argv[1] if len(sys.argv) > 1 else "gibbs"
for chunk in (10, 10, 20, 40, 80, 160, 320, 640):
    torch.cuda.synchronize(); t = time.perf_counter()
    for s in range(chunk):
        dg.conv_step(o, tot + s)
        if product == "gibbs": dg.product_step(o, tot + s, "lcv", "gibbs")
        else: dg.product_step(o, tot + s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t; t_all += dt
    tot += chunk
    r, mx, m = dist()
    dm = m - mprev; dm[:, 2] = np.arctan2(np.sin(dm[:, 2]), np.cos(dm[:, 2])); mprev = m
    print("iter %4d  %.2f ms/iter  total %.3f s  rms-to-parametric %.3f  max %.3f  mean-change rms %.4f" % (tot, 1e3 * dt / chunk, t_all, r, mx, np.sqrt(np.mean(dm[:, :2] ** 2))))

# ---- from the parametric solution (IIF: initParametricFrom! / autoinit from solveGraphParametric!) ----
dg.init_from_means(xp)
r, mx, mprev = dist(); print("parametric init: rms-to-parametric %.4f max %.3f" % (r, mx))
tot = 0; t_all = 0; prev_r = r
for blk in range(12):
    torch.cuda.synchronize(); t = time.perf_counter()
    for s in range(5):
        dg.conv_step(o, 5000 + tot + s); dg.product_step(o, 5000 + tot + s, "lcv", "gibbs")
    torch.cuda.synchronize(); dt = time.perf_counter() - t; t_all += dt; tot += 5
    r, mx, m = dist()
    dm = m - mprev; dm[:, 2] = np.arctan2(np.sin(dm[:, 2]), np.cos(dm[:, 2])); mprev = m
    _, sd = dg.belief_stats(R.Pose2)
    print("iter %3d total %.4f s rms-to-parametric %.4f (change %.4f) max %.3f mean-change rms %.4f  mean std %s" % (tot, t_all, r, r - prev_r, mx, np.sqrt(np.mean(dm[:, :2] ** 2)), sd.mean(0).cpu().numpy().round(3)))
    prev_r = r
