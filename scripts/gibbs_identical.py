"""fraction of product samples identical (1e-9) between the device's multiscale Gibbs product and the oracle's restatement, on the problems of
tests/test_gpu_gibbs.py::test_device_equals_oracle_sample_by_sample (the device evaluates exp / log with the hardware transcendentals,
the oracle with the specified single-precision polynomials: a draw can differ when a uniform lands within an ulp of a boundary)"""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import test_gpu_gibbs as T
import rome_jl_amd as R, oracle as ro
T.R, T.torch = R, torch
R.default_context()
tot = same_n = 0
for dim, N, iters in [(3, 100, 1), (3, 100, 2), (2, 100, 1), (3, 64, 1), (3, 37, 1), (3, 128, 1)]:
    for rep in range(4):
        rng = np.random.default_rng(100 * dim + N + iters + 1000 * rep)
        circ = 0b100 if dim == 3 else 0
        Ks = [2, 3, 0, 1, 5, 11, 2, 4, 7, 3] * 3
        ptr, rows, prop = T._problem(dim, N, Ks, rng, bool(circ))
        bw = ro.kde_bandwidths(prop, circ)
        bel_in = rng.standard_normal((len(Ks), dim, N))
        got = T._device_product(dim, N, ptr, rows, prop, bw, bel_in, circ, iters)
        ref = ro.product_msgibbs(ro.make_opts(N=N, seed=11, stream_offset=5), dim, ptr, rows, prop, bw, bel_in, circ, iters)
        d = got - ref
        if circ: d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
        same = np.abs(d).max(axis=1) < 1e-9
        use = np.array(Ks) >= 2
        tot += same[use].size; same_n += int(same[use].sum())
print("identical samples: %d of %d = %.5f" % (same_n, tot, same_n / tot))
