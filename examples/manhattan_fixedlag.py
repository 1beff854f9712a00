#!/usr/bin/env python3
"""Counterpart of the reference's examples/ManhattanDatasetFixedLag.jl: incremental fixed-lag operation on Manhattan records --
instructions are parsed one by one (parseG2oInstruction), every new pose is initialised by convolving its odometry factor
(approxConv), every `stride` steps the oldest variables beyond the window are frozen (fifoFreeze, qfl) and the window is re-solved on
the GPU (DeviceGraph.set_frozen + solve); the result is stored as a DFG archive (saveDFG, the format the reference's example writes as
`fg-after-solve<step>`) and as a g2o file.

    python examples/manhattan_fixedlag.py [n_instructions=200] [qfl=20] [stride=10] [file.g2o]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rome_jl_amd as R  # noqa: E402


def run(n_instructions=200, qfl=20, stride=10, path=None, out_prefix="/tmp/manhattan_fixedlag", N=100, sweeps=4, verbose=True):
    path = path or os.path.join(ROOT, "tests", "golden", "manhattan.g2o")
    # time order: an edge becomes available when its later pose exists (what manhattan_incremental.g2o is sorted by)
    instructions = sorted((i for i in R.importG2o(path) if i[0] == "EDGE_SE2"), key=lambda i: (max(int(i[1]), int(i[2])), min(int(i[1]), int(i[2]))))
    instructions = instructions[:n_instructions]
    fg = R.initfg(N)
    fg.addVariable("x0", R.Pose2)
    fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), np.diag(np.square([0.1, 0.1, 0.05])))))   # ManhattanDatasetFixedLag.jl:41-42
    fg.initVariable("x0", R.approxConv(fg, fg.factors[0][0], "x0", seed=1))
    t_solve, n_solves = 0.0, 0
    for step, ins in enumerate(instructions, 1):
        before = set(fg.ls())
        R.parseG2oInstruction(fg, ins)
        for l in fg.ls():
            if l not in before:                                 # new pose: proposal through the factor that introduced it
                fg.initVariable(l, R.approxConv(fg, fg.factors[-1][0], l, seed=1000 + step))
        if step % stride == 0 or step == len(instructions):
            frozen = R.fifoFreeze(fg, qfl=qfl)
            t = time.perf_counter()
            dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
            dg.set_frozen(frozen)
            dg.solve(R.make_opts(N=N, solver=R.SOLVER_NEWTON, seed=step), n_sweeps=sweeps)
            dg.download_beliefs(fg)
            t_solve += time.perf_counter() - t; n_solves += 1
            if verbose:
                print("step %4d: %4d variables (%4d frozen), %4d factors, window solve %.1f ms" %
                      (step, len(fg.ls()), len(frozen), len(fg.factors), 1e3 * (time.perf_counter() - t)))
    est = R.calcPPE(np.stack([fg.getVal(l) for l in fg.ls()]))
    fg.ppes = {l: {"default": {k: est[k][i] for k in est}} for i, l in enumerate(fg.ls())}
    fg.bws = dict(zip(fg.ls(), R.kde_bandwidth(np.stack([fg.getVal(l) for l in fg.ls()]))))
    arch = R.saveDFG(fg, out_prefix + "-fg-after-solve%04d.tar.gz" % len(instructions))
    g2o = out_prefix + ".g2o"
    R.exportG2o(fg, filename=g2o, estimates={l: est["suggested"][i] for i, l in enumerate(fg.ls())}, varIntLabel={l: int(l[1:]) for l in fg.ls()})
    if verbose:
        print("%d window solves, %.1f ms each on average; wrote %s and %s" % (n_solves, 1e3 * t_solve / max(n_solves, 1), arch, g2o))
    return fg, arch, g2o


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]) if len(a) > 0 else 200, int(a[1]) if len(a) > 1 else 20, int(a[2]) if len(a) > 2 else 10, a[3] if len(a) > 3 else None)
