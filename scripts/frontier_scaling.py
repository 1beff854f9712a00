#!/usr/bin/env python3
"""Up-solve plan time against frontier size: independent single-frontal cliques of Manhattan-3500 (N = 100, gibbsIters = 3), device-resident
store, one UpsolvePlan per size -- where a frontier stops paying latency (one block per kernel) and starts paying throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rome_jl_amd as R
from rome_jl_amd.clique import DeviceStore, UpsolvePlan
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
N = 100
fg = R.loadG2o(os.path.join(root, "tests/golden/manhattan.g2o"), N=N); R.dead_reckon_init(fg, seed=11)
nbr = {l: set() for l in fg.variables}
for _, labels, _ in fg.factors:
    for a in labels:
        nbr[a].update(b for b in labels if b != a)
chosen, blocked = [], set()
for l in fg.variables:
    if l not in blocked:
        chosen.append(l); blocked.add(l); blocked.update(nbr[l])
store = DeviceStore(fg)
print("cliques   ms per plan run (3 Gibbs iterations)   us per clique")
for n in (1, 4, 16, 64, 256, 1024, len(chosen)):
    plan = UpsolvePlan(store, [[l] for l in chosen[:n]], gibbsIters=3)
    for w in range(3):
        plan.run(R.make_opts(N=N, seed=w))
    torch.cuda.synchronize(); store.ctx.synchronize()
    ts = []
    for rep in range(7):
        t0 = time.perf_counter()
        plan.run(R.make_opts(N=N, seed=10 + rep)); store.ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    ms = 1e3 * float(np.median(ts))
    print("%7d   %8.3f   %8.2f" % (n, ms, 1e3 * ms / n))
