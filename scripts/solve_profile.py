#!/usr/bin/env python3
"""Time split of the device solve loop on the synthetic Manhattan graph (conv sweep vs product), with and
without hipGraph replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R

fg = R.synth_manhattan(); R.dead_reckon_init(fg, seed=1)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=100, solver=1, seed=11)

def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); one = max(time.perf_counter() - t0, 1e-6)
    for _ in range(min(5000, int(0.15 / one))):   # steady state: ~0.15 s of back-to-back launches before timing (queued, then drained)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print("conv sweep      %.3f ms" % timeit(lambda: dg.conv_step(o, 0)))
print("product         %.3f ms" % timeit(lambda: dg.product_step(o, 0)))
print("conv + product  %.3f ms" % timeit(lambda: (dg.conv_step(o, 0), dg.product_step(o, 0))))
t = time.perf_counter(); dg.solve(o, 50); torch.cuda.synchronize(); print("solve(50) wall  %.3f ms/sweep" % ((time.perf_counter() - t) * 20))
saved = dg.bel[R.Pose2].clone()
dg.conv_step(o, 0)
print("manikde! bandwidths + importance product  %.3f ms" % timeit(lambda: dg.product_step(o, 0, "lcv"), 10))
dg.bel[R.Pose2].copy_(saved); dg.conv_step(o, 0)
print("manikde! bandwidths + Gibbs product (manifoldProduct, Niter=1)  %.3f ms" % timeit(lambda: dg.product_step(o, 0, "lcv", "gibbs"), 10))
dg.bel[R.Pose2].copy_(saved)
K = np.diff(dg.csr[R.Pose2]["ptr_h"])
print("proposals per variable: max %d, mean %.2f, histogram %s" % (K.max(), K.mean(), np.bincount(K).tolist()))
