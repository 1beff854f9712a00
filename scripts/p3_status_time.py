"""Pose3Pose3 packed sweep on the 10k helix with a status array (NEWTON: the residual at every returned root) against the plain sweep"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rome_jl_amd as R
fg = R.synth_helix3d(P=10000, N=100); R.dead_reckon_init_pose3(fg, seed=2)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
tb = dg.tab["p3p3"]; out = torch.empty((tb["C"], 6, 100), dtype=torch.float64, device="cuda")
st = torch.zeros((tb["C"], 100), dtype=torch.int32, device="cuda")
o = R.make_opts(N=100, solver=1)
for name, kw in (("newton", {}), ("newton + status", dict(status=st))):
    for _ in range(200): dg.sweep_pose3pose3(o, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): dg.sweep_pose3pose3(o, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    print("%-16s %.1f us per sweep   unconverged %d" % (name, 1e3 * e0.elapsed_time(e1) / 200, int(st.sum())))
