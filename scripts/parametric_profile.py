#!/usr/bin/env python3
"""Where the time of solveGraphParametric goes on the synthetic Manhattan graph (cProfile, top entries)."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rome_jl_amd as R
fg = R.synth_manhattan(); R.dead_reckon_init(fg, seed=1)
R.solveGraphParametric(R.synth_manhattan(P=200, loops=40))   # warm up library / context
t = time.perf_counter(); pr = cProfile.Profile(); pr.enable()
xp = R.solveGraphParametric(fg)
pr.disable(); print("total %.3f s" % (time.perf_counter() - t))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
