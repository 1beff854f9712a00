#!/usr/bin/env python3
"""The reference's own solved graph (tests/golden/manhattan500_reference_solve.npz, made from
examples/fg-after-solve.tar.gz) against this library: convolution consistency, parametric solve and the
device-resident nonparametric solve on the same 500 Pose2Pose2 + 1 PriorPose2 factors."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R

d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/manhattan500_reference_solve.npz"))
E, MU, COV = d["edges"], d["mu"], d["cov"]
REF = d["particles"].astype(np.float64)          # [V, N, 3]
V, N = REF.shape[0], REF.shape[1]
ppe = d["ppe"][:, 0]


def wrap(a):
    return np.arctan2(np.sin(a), np.cos(a))


WINV = np.linalg.inv(COV)


def chi2(X):
    """Sum of whitened squared Pose2Pose2 residuals at the point estimates X [V, 3]."""
    p, q = X[E[:, 0]], X[E[:, 1]]
    c, s = np.cos(p[:, 2]), np.sin(p[:, 2])
    r = np.stack([p[:, 0] + c * MU[:, 0] - s * MU[:, 1] - q[:, 0], p[:, 1] + s * MU[:, 0] + c * MU[:, 1] - q[:, 1],
                  wrap(p[:, 2] + MU[:, 2] - q[:, 2])], 1)
    return float(np.einsum("fi,fij,fj->", r, WINV, r))


fg = R.initfg(N)
for k in range(V):
    fg.addVariable("x%d" % k, R.Pose2)
fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(d["prior_mu"], d["prior_cov"])))
for (i, j), m, c in zip(E, MU, COV):
    fg.addFactor(["x%d" % i, "x%d" % j], R.Pose2Pose2(R.MvNormal(m, c)))
for k in range(V):
    fg.initVariable("x%d" % k, REF[k].T.copy())

# ---- parametric (Gaussian MAP) solution vs the reference's posterior point estimates
R.dead_reckon_init(fg, seed=1)
t = time.perf_counter(); xp = R.solveGraphParametric(fg); tp = time.perf_counter() - t
mp = np.array([xp["x%d" % k] for k in range(V)])
dd = mp - ppe; dd[:, 2] = wrap(dd[:, 2])
print("parametric solve %.2f s: |Δ| vs reference PPE  rms xy %.3f m  max xy %.3f m  rms θ %.4f  max θ %.4f" %
      (tp, np.sqrt((dd[:, :2] ** 2).sum(1).mean()), np.abs(dd[:, :2]).max(), np.sqrt((dd[:, 2] ** 2).mean()), np.abs(dd[:, 2]).max()))
ref_sd = np.stack([REF[:, :, 0].std(1), REF[:, :, 1].std(1)], 1)
print("chi2 over the 500 factors: reference PPE %.3e, parametric %.3e" % (chi2(ppe), chi2(mp)))
print("reference posterior std: mean", ref_sd.mean(0).round(3), "max", ref_sd.max(0).round(3))

# ---- nonparametric device solve
dg = R.DeviceGraph(fg)
for init in ("dead_reckoning", "parametric"):
    if init == "dead_reckoning":
        R.dead_reckon_init(fg, seed=1); dg.upload_beliefs(fg)
    else:
        dg.init_from_means(xp)
    o = R.make_opts(N=N, solver=1, seed=5)
    tot = 0
    for chunk in (10, 40, 150, 800):
        torch.cuda.synchronize(); t = time.perf_counter()
        for s in range(chunk):
            dg.conv_step(o, tot + s); dg.product_step(o, tot + s, os.environ.get("ROME_BW", "silverman"))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        tot += chunk
        m, sd = dg.belief_stats(R.Pose2)
        m = m.cpu().numpy()[:V]; sd = sd.cpu().numpy()[:V]
        dd = m - ppe; dd[:, 2] = wrap(dd[:, 2])
        dm = m - mp; dm[:, 2] = wrap(dm[:, 2])
        print("%-14s sweeps %4d %.2f ms/sweep: chi2 %.3e | rms xy to parametric %.3f, to reference PPE %.3f | std ratio (ours/ref) median %s" %
              (init, tot, 1e3 * dt / chunk, chi2(m), np.sqrt((dm[:, :2] ** 2).sum(1).mean()),
               np.sqrt((dd[:, :2] ** 2).sum(1).mean()), np.median(sd[:, :2] / ref_sd, 0).round(2)))
