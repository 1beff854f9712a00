"""Gibbs product throughput against the LDS footprint of a block (= resident waves per SIMD): 3500 variables x 3 proposals, the
launch sized for max_k = 3 .. 14 proposals (what the largest variable of a graph dictates for everyone).  profiles/r02_gibbs_occupancy.txt"""
import sys, time, numpy as np, ctypes as C
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, rome_jl_amd as R
from rome_jl_amd import _lib
ctx = R.default_context(); dev = torch.device("cuda", 0); lib = _lib.load()
rng = np.random.default_rng(0)
N = 100
def run(Ks, max_k_override=None):
    V = len(Ks); ptr = np.concatenate([[0], np.cumsum(Ks)]).astype(np.int32); rows = np.arange(ptr[-1], dtype=np.int32)
    centre = rng.normal(0, 3, (V, 3))
    prop = np.concatenate([centre[v][None, :, None] + rng.normal(0, 0.3, (K, 3, 1)) + rng.uniform(0.1, 0.6, (K, 3, 1)) * rng.standard_normal((K, 3, N)) for v, K in enumerate(Ks)])
    bw = np.full((len(prop), 3), 0.15)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    tp, tb = t(prop, torch.float64), t(bw, torch.float64); ti = torch.zeros((V, 3, N), dtype=torch.float64, device=dev); out = torch.empty_like(ti)
    tptr, trows = t(ptr, torch.int32), t(rows, torch.int32)
    o = R.make_opts(N=N, seed=1); ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    mk = int(max_k_override or max(Ks))
    def call(): _lib.check(lib.rome_product_gibbs_dev(ctx.handle, C.byref(o), 3, V, tptr.data_ptr(), trows.data_ptr(), tp.data_ptr(), tb.data_ptr(), len(prop), ti.data_ptr(), out.data_ptr(), 4, 1, mk), ctx.handle)
    call(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): call()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e3
Ks = [3] * 3500
for mk in (3, 4, 6, 8, 11, 14):
    print("K=3 x 3500, LDS sized for max_k=%d: %.3f ms" % (mk, run(Ks, mk)))
