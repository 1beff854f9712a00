#!/usr/bin/env python3
"""bearing-range -> pose sweep (k_conv<BR<1>, closed form>) on the MIT-like tables: ms per sweep of the library named by ROME_MI355_LIB.
Used with the experiment builds -DROME_EXPERIMENT_NO_SPREAD / -DROME_EXPERIMENT_NO_ENTROPY_RNG (scripts/build_variant.sh) to BOUND what
cheaper wave reductions (row packing, single-precision or incremental spread sums) or a cheaper jitter generator could save."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rome_jl_amd as R
fg = R.synth_mit_br(P=8080, n_landmarks=2000, N=100); R.dead_reckon_init(fg, seed=4)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
F = dg.tab["br"]["F"]
o = R.make_opts(N=100, solver=1)
fn = lambda: dg.sweep_bearingrange(o, 1)
fn(); torch.cuda.synchronize()
for _ in range(3000): fn()
torch.cuda.synchronize()
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): fn()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1))
print("%-40s %d rows: %s us per sweep" % (os.path.basename(os.environ.get("ROME_MI355_LIB", "shipped")), F, " ".join("%.2f" % r for r in res)))
