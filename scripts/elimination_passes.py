"""Manhattan-3500 elimination solve, PASSES only (for rocprofv3): builds the solver, runs a warm-up pass, then --passes timed passes;
with --steps prints the wall-clock of every schedule step of one pass (synchronised after each: slower than the pass itself).
    python scripts/elimination_passes.py [--passes 20] [--steps] [--out gpurun_out/r06_elimination_steps.txt]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rome_jl_amd as R   # noqa: E402
from rome_jl_amd.elimination import RelativeEliminationSolver   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=20)
ap.add_argument("--steps", action="store_true")
ap.add_argument("--out", default=None)
a = ap.parse_args()
N = 100
fg = R.loadG2o(os.path.join(ROOT, "tests", "golden", "manhattan.g2o"), N=N)
t0 = time.perf_counter(); es = RelativeEliminationSolver(fg); tb = time.perf_counter() - t0
ctx = es.store.ctx
lines = ["built in %.3f s: %s" % (tb, es.stats())]
es.solve(R.make_opts(N=N, seed=1)); ctx.synchronize()
t0 = time.perf_counter()
for ps in range(a.passes):
    es.reset(); es.solve(R.make_opts(N=N, seed=2 + ps))
ctx.synchronize()
lines.append("%d passes: %.2f ms per pass" % (a.passes, 1e3 * (time.perf_counter() - t0) / a.passes))
if a.steps:
    o = R.make_opts(N=N, seed=99)
    lines.append("per schedule step (synchronised): kind, destinations / entries, rows, ms")
    for (kind, x), (k2, spec) in zip(es.steps[0], es.schedules[0]):
        ctx.synchronize(); t0 = time.perf_counter()
        es._run(x, o) if kind == "plan" else x.run()
        ctx.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
        if kind == "plan":
            lines.append("  plan     %5d destinations %6d rows %3d groups  %.3f ms" % (len(spec.order), len(spec.pairs), len(set(spec.groups)), dt))
        else:
            lines.append("  %-8s %5d entries                         %.3f ms" % (k2, len(spec), dt))
print("\n".join(lines))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    open(a.out, "w").write("\n".join(lines) + "\n")
