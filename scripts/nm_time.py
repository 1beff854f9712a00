"""Nelder-Mead sweeps (the reference's optimizer) of the three Pose2 / Point2 factor directions: ms per sweep (kernel A/B: ROME_NM_MINWAVES)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R
tag = os.environ.get("ROME_MI355_LIB", "shipped")
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
o = R.make_opts(N=100, solver=2)
fg = R.loadG2o(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "manhattan.g2o"), N=100); R.dead_reckon_init(fg, seed=1)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
out = torch.empty((dg.tab["p2p2"]["C"], 3, 100), dtype=torch.float64, device="cuda")
print("%s Pose2Pose2 NM (Manhattan, %d convs): %.3f ms" % (tag, dg.tab["p2p2"]["C"], timeit(lambda: dg.sweep_pose2pose2(o, out=out), 10)))
del dg
fg = R.synth_mit_br(P=8080, n_landmarks=2000, N=100); R.dead_reckon_init(fg, seed=4)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
for d in (1, 0):
    print("%s bearing-range dir %d NM (%d convs): %.3f ms" % (tag, d, dg.tab["br"]["F"], timeit(lambda: dg.sweep_bearingrange(o, d), 5)))
