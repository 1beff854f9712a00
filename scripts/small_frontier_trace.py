#!/usr/bin/env python3
"""one up-solve plan over a frontier of the 36-pose honeycomb + multihypo graph (BASELINE configs[3] at the reference's size): run under
rocprofv3 --kernel-trace to see which kernel's LATENCY a small frontier pays (scripts/rocpd_summary.py on the result)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R
from rome_jl_amd.clique import DeviceStore, UpsolvePlan
N = 100
fg = R.synth_beehive_mh(int(os.environ.get("POSES", "36")), N=N); R.dead_reckon_init(fg, seed=3)
rng = np.random.default_rng(2)
for l, t in fg.variables.items():
    if t is R.Point2:
        fg.initVariable(l, np.asarray(fg._sim[l])[:, None] + 0.5 * rng.standard_normal((2, N)))
nbr = {l: set() for l in fg.variables}
for _, labels, _ in fg.factors:
    for a in labels:
        nbr[a].update(b for b in labels if b != a)
chosen, blocked = [], set()
for l in fg.variables:
    if l not in blocked:
        chosen.append(l); blocked.add(l); blocked.update(nbr[l])
store = DeviceStore(fg)
plan = UpsolvePlan(store, [[l] for l in chosen], gibbsIters=3)
for w in range(6):
    plan.run(R.make_opts(N=N, seed=w))
torch.cuda.synchronize()
print("frontier of %d cliques, 6 runs x 3 Gibbs iterations" % len(chosen))
