"""Gauss-Newton on the SE(3) helix: actual against PREDICTED cost decrease of every undamped step (rho ~ 1: the linear model is right and the slow
tail is the problem's; rho << 1: residual / Jacobian / retraction do not belong together), and the same with the step applied in the pose's OWN frame"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rome_jl_amd as R
from rome_jl_amd import parametric as PM
P_ = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
fg = R.synth_helix3d(P=P_, N=8); R.dead_reckon_init_pose3(fg, seed=7)
ctx = R.default_context()
P = PM._Problem(fg)
X = P.pack(PM.initParametric(fg))
r, J = P.linearize(X, ctx, None)
cost = float(r @ r)
for it in range(12):
    H = P.normal_matrix(P.blocks); g = J.T @ r
    d = np.empty(P.n); dp = P.solve_spd(P.damped(H, 1e-12), -g); d[P.perm] = dp
    pred = -(2 * g @ dp + dp @ (H @ dp))              # cost - |r + J d|^2
    Xn = P.retract(X, d)
    rn, Jn = P.linearize(Xn, ctx, None)
    cn = float(rn @ rn)
    lin = r + J @ dp
    print("iter %2d cost %.4f -> %.4f  actual decrease %.4f  predicted %.4f  rho %.3f  |d|max %.3e  |rn - (r + J d)| / |J d| = %.3e" %
          (it, cost, cn, cost - cn, pred, (cost - cn) / pred if pred else float('nan'), np.abs(d).max(), np.linalg.norm(rn - lin) / max(np.linalg.norm(J @ dp), 1e-300)))
    X, r, J, cost = Xn, rn, Jn, cn
