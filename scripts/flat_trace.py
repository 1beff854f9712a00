#!/usr/bin/env python3
"""Per-block timeline of one steady-state launch of the packed sweep kernel (experiment build with -DROME_FLAT_TRACE,
ROME_MI355_LIB=scripts/ubench/lib_trace.so): when does each block start / finish relative to the first one?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R

fg = R.loadG2o(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "manhattan.g2o"), N=100)
R.dead_reckon_init(fg, seed=11)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
tb = dg.tab["p2p2"]
out = torch.empty((tb["C"], 3, 100), dtype=torch.float64, device="cuda")
st = torch.zeros((tb["C"], 100), dtype=torch.int32, device="cuda")
plan0 = dg.plan_sweep_pose2pose2(R.make_opts(N=100, solver=R.SOLVER_CLOSED_FORM), out)
plan = dg.plan_sweep_pose2pose2(R.make_opts(N=100, solver=R.SOLVER_CLOSED_FORM), out, status=st)
for _ in range(4000): plan0()
plan(); torch.cuda.synchronize()
nb = (tb["C"] + 4) // 5
tr = st.cpu().numpy().reshape(-1).view(np.uint64)[:4 * nb].reshape(nb, 4)
t0 = tr[:, 0].astype(np.int64); t1 = tr[:, 1].astype(np.int64); t2 = tr[:, 2].astype(np.int64)
base = t0.min()
tick = 10.0  # ns per wall_clock64 tick (100 MHz)
print("blocks %d  first start 0  last start %.0f ns  last end %.0f ns" % (nb, (t0.max() - base) * tick, (t2.max() - base) * tick))
for q in (0, 10, 25, 50, 75, 90, 95, 99, 100):
    print("  pct %3d: start %7.0f  compute-done %7.0f  end %7.0f  (block duration %6.0f)" %
          (q, np.percentile(t0 - base, q) * tick, np.percentile(t1 - base, q) * tick, np.percentile(t2 - base, q) * tick, np.percentile(t2 - t0, q) * tick))
late = np.argsort(t0)[-140:]
print("latest-starting 140 blocks: start %.0f..%.0f ns, duration median %.0f ns" % ((t0[late].min() - base) * tick, (t0[late].max() - base) * tick, np.median((t2 - t0)[late]) * tick))
hw = tr[:, 3] & 0xFFFFFFFF; xcc = tr[:, 3] >> 32
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
ids = xcc * 1000 + se * 100 + sh * 16 + cu
u, cnt = np.unique(ids, return_counts=True)
print("distinct (xcc,se,sh,cu): %d; blocks per CU min %d max %d mean %.2f" % (len(u), cnt.min(), cnt.max(), cnt.mean()))
print("xcc of block b == b %% 8: %.3f" % np.mean(xcc == (np.arange(nb) % 8)))
