"""Times rome_kde_bandwidth_dev at Manhattan scale: the 3500 Pose2 beliefs and the 10 907 convolution proposals of one sweep
(what `manikde!` does after every convolution in the reference), N = 100.  Prints ms per call and LL evaluations per second."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rome_jl_amd as R
from rome_jl_amd import _lib

ctx = R.default_context()
lib = _lib.load()
N = 100
rng = np.random.default_rng(0)
for V, what in ((3500, "beliefs"), (10907, "proposals")):
    bel = np.empty((V, 3, N))
    bel[:, 0] = rng.normal(0, 0.1, (V, N)); bel[:, 1] = rng.normal(0, 0.1, (V, N)); bel[:, 2] = rng.normal(0, 0.02, (V, N))
    d = torch.as_tensor(bel, device="cuda"); bw = torch.empty((V, 3), dtype=torch.float64, device="cuda")
    for tols in ((0.0, 0.0), (1e-2, 1e-2)):
        def call():
            _lib.check(lib.rome_kde_bandwidth_dev(ctx.handle, 3, V, N, d.data_ptr(), 0b100, tols[0], tols[1], bw.data_ptr()), ctx.handle)
        call(); ctx.synchronize() if hasattr(ctx, "synchronize") else torch.cuda.synchronize(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            call()
        ctx.synchronize() if hasattr(ctx, "synchronize") else None
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print("%s V=%d tol=%s: %.3f ms per call (%.1f ns per (belief, coordinate))" % (what, V, tols, dt * 1e3, dt * 1e9 / (3 * V)))
    mx = torch.empty((V, 3), dtype=torch.float64, device="cuda")
    def callm():
        _lib.check(lib.rome_kde_max_dev(ctx.handle, 3, V, N, d.data_ptr(), bw.data_ptr(), 0, mx.data_ptr()), ctx.handle)
    callm(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        callm()
    torch.cuda.synchronize()
    print("%s V=%d getKDEMax: %.3f ms per call" % (what, V, (time.perf_counter() - t) / 5 * 1e3))
