"""the whole user-facing call R.solveTree(fg) on Manhattan-3500 (N = 100): structure + plans + one pass + download + PPEs, first call in a warm process
and a second graph object (everything rebuilt); then a re-solve on the returned solver"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rome_jl_amd as R
G2O = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "manhattan.g2o")
R.default_context()
import cProfile, pstats
for rep in range(3):
    fg = R.loadG2o(G2O, N=100)
    t = time.perf_counter(); es = R.solveTree(fg); t1 = time.perf_counter(); R.solveTree(fg, tree=es, seed=77); t2 = time.perf_counter()
    print("solveTree(fg): %.3f s (build %s)   re-solve on the same structure: %.3f s" % (t1 - t, {k: round(v, 3) for k, v in es.build_s.items()}, t2 - t1))
fg = R.loadG2o(G2O, N=100)
pr = cProfile.Profile(); pr.enable(); R.solveTree(fg); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
