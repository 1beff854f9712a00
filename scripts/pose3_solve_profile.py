"""Pose3 solve iteration on the synthetic helix (BASELINE configs[4] shape: 10^4 Pose3, ~1.2e4 Pose3Pose3): convolution sweep and
dim-6 proposal product, ms per launch (N = 100)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R

N = 100
fg = R.synth_helix3d(P=10000, N=N, seed=3)
R.dead_reckon_init_pose3(fg, seed=2)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=N, solver=1, seed=5)
dg.conv_step(o, 0); dg.product_step(o, 0); torch.cuda.synchronize()
for name, fn in (("conv_step", lambda s: dg.conv_step(o, s)), ("product_step", lambda s: dg.product_step(o, s)),
                 ("iteration", lambda s: (dg.conv_step(o, s), dg.product_step(o, s)))):
    t = time.perf_counter()
    for s in range(20):
        fn(s)
    torch.cuda.synchronize()
    print("Pose3 helix (%d poses, %d convolutions): %-12s %.3f ms" % (len(dg.packed.labels[R.Pose3]), dg.tab["p3p3"]["C"], name, (time.perf_counter() - t) / 20 * 1e3))
dg.conv_step(o, 0); dg.product_step(o, 0, "lcv", "gibbs"); torch.cuda.synchronize()
t = time.perf_counter()
for s in range(5):
    dg.conv_step(o, s); dg.product_step(o, s, "lcv", "gibbs")
torch.cuda.synchronize()
print("Pose3 helix: iteration with manikde! bandwidths + Gibbs product on SE(3)  %.3f ms" % ((time.perf_counter() - t) / 5 * 1e3))
