"""Where a LONE block of k_product_gibbs spends its time (experiment build: scripts/build_variant.sh gtrace UNIT=rome_gibbs -DROME_GIBBS_TRACE,
ROME_MI355_LIB=scripts/ubench/lib_gtrace.so): one variable with K proposals, N = 100, Pose2; wall_clock64 stamps of block 0 / thread 0 summed per
phase over the levels.  -> profiles/r06_gibbs_phase_trace.txt"""
import ctypes as C, os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import test_gpu_gibbs as T
import rome_jl_amd as R, oracle as ro
from rome_jl_amd import _lib
T.R, T.torch = R, torch
R.default_context()
lib = _lib.load()
names = ["arguments", "row check", "logn + tree constants", "labels + level 0", "(a) point", "stage level", "(c) labels | point", "(d) Gibbs sweep"]
for K in (2, 3, 6, 8):
    rng = np.random.default_rng(K)
    ptr, rows, prop = T._problem(3, 100, [K], rng, True)
    bw = ro.kde_bandwidths(prop, 0b100)
    bel_in = rng.standard_normal((1, 3, 100))
    acc = np.zeros(16)
    reps = 30
    for r in range(reps + 3):
        T._device_product(3, 100, ptr, rows, prop, bw, bel_in, 0b100, 1, seed=11 + r)
        buf = (C.c_ulonglong * 16)()
        assert lib.rome_debug_gibbs_trace(buf) == 0
        if r >= 3:
            acc += np.array(list(buf), dtype=np.float64)
    us = acc / reps * 0.01
    print("K = %d: block 0 total %.1f us  |  " % (K, us[:8].sum()) + "  ".join("%s %.1f" % (n, u) for n, u in zip(names, us)))
