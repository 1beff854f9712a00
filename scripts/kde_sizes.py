import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch, rome_jl_amd as R
from rome_jl_amd import _lib
ctx = R.default_context(); lib = _lib.load()
rng = np.random.default_rng(0)
for N in (32, 50, 64, 65, 100, 128):
    V = 4000
    bel = np.empty((V, 3, N)); bel[:, 0] = rng.normal(0, 0.1, (V, N)); bel[:, 1] = rng.normal(0, 0.1, (V, N)); bel[:, 2] = rng.normal(0, 0.02, (V, N))
    d = torch.as_tensor(bel, device="cuda"); bw = torch.empty((V, 3), dtype=torch.float64, device="cuda")
    call = lambda: _lib.check(lib.rome_kde_bandwidth_dev(ctx.handle, 3, V, N, d.data_ptr(), 0b100, 0.0, 0.0, bw.data_ptr()), ctx.handle)
    call(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): call()
    torch.cuda.synchronize(); print("N=%d: %.3f ms per %d beliefs" % (N, (time.perf_counter() - t) / 5 * 1e3, V))
