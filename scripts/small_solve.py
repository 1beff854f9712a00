import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch, rome_jl_amd as R
for name, fg in (("hexagon", R.generateGraph_Hexagonal(N=100)), ("beehive_mh(36)", R.synth_beehive_mh(36, N=100))):
    R.dead_reckon_init(fg, seed=5)
    for l, t in fg.variables.items():
        if t is R.Point2 and not fg.isInitialized(l):
            fg.initVariable(l, np.asarray(fg._sim[l] if hasattr(fg, "_sim") and l in fg._sim else [20.0, 0.0])[:, None] + np.random.default_rng(1).standard_normal((2, 100)))
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    o = R.make_opts(N=100, solver=1, seed=3)
    dg.solve(o, n_sweeps=3, bandwidth="lcv", product="gibbs"); torch.cuda.synchronize()
    t = time.perf_counter(); dg.solve(o, n_sweeps=50, bandwidth="lcv", product="gibbs"); torch.cuda.synchronize()
    print("%s: %.3f ms per solve iteration (conv + manikde! + Gibbs), %d variables" % (name, (time.perf_counter() - t) / 50 * 1e3, len(fg.variables)))
