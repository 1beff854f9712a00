"""N>1 path on CPU: 2 ranks over gloo.  Covers the sharding helpers and the separator-belief
all-gather (`rome_jl_amd.distributed`), with the oracle standing in -- as test infrastructure only --
for the per-rank sweep compute, so that the sharded result can be checked against the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rome_jl_amd as R
        from rome_jl_amd.distributed import (chain_segment_exchange, shard_range, owner_of, shard_convolutions_by_target)
        import oracle as ro
        N = 32
        # ---- (a) separator exchange between chained segments ----
        V = 6  # 4 owned poses + ghost_prev (4) + ghost_next (5)
        bel = torch.full((V, 3, N), float(rank), dtype=torch.float64)
        prop = torch.zeros((7, 3, N), dtype=torch.float64)       # proposal table: row 2 = first pose, row 5 = last pose
        prop[2] = 100.0 + rank; prop[5] = 200.0 + rank
        ex = chain_segment_exchange(torch, dist, world, rank, N, "cpu", [2, 5], ghost_prev=4, ghost_next=5)
        ex.complete(bel)                                           # nothing posted yet: no-op
        ok_a = bool((bel == float(rank)).all())
        ex.post(prop)                                              # asynchronous collective ...
        prop[2] = -1.0                                             # ... is not affected by later writes to the source
        ex.complete(bel)
        ok_a = ok_a and bool((bel[4] == 200.0 + (rank - 1) % world).all() and (bel[5] == 100.0 + (rank + 1) % world).all()
                             and (bel[:4] == float(rank)).all())
        prop[2] = 300.0 + rank
        ex.exchange(prop, bel)                                     # synchronous form
        ok_a = ok_a and bool((bel[5] == 300.0 + (rank + 1) % world).all())
        # ---- (b) strong-scaling shard of one graph by target ownership == unsharded sweep ----
        fg = R.synth_manhattan(P=60, loops=25, seed=5, N=N)
        R.dead_reckon_init(fg, seed=2)
        pk = R.PackedGraph(fg)
        factor, dr, fixed, target = R.PackedGraph.conv_table(pk.p2p2)
        L = R.cholesky_lower(pk.p2p2["cov"])
        belief = pk.beliefs(fg, R.Pose2)
        rows = shard_convolutions_by_target(target, 60, world, rank)
        o = ro.make_opts(N=N, solver=ro.SOLVER_NEWTON, seed=77)
        full = ro.conv_pose2pose2(o, pk.p2p2["mu"], L, belief, fixed, target, dr, factor=factor)
        mine = np.zeros_like(full)
        for r_ in rows:   # global conv id = Philox stream -> partition-independent results
            oo = ro.make_opts(N=N, solver=ro.SOLVER_NEWTON, seed=77, stream_offset=int(r_))
            mine[r_] = ro.conv_pose2pose2(oo, pk.p2p2["mu"], L, belief, fixed[r_:r_ + 1], target[r_:r_ + 1], dr[r_:r_ + 1], factor=factor[r_:r_ + 1])[0]
        t = torch.from_numpy(mine)
        dist.all_reduce(t)  # disjoint rows -> sum assembles the full table
        ok_b = bool(np.array_equal(t.numpy(), full))
        lo, hi = shard_range(60, world, rank)
        ok_c = all(owner_of(i, 60, world) == rank for i in range(lo, hi)) and (hi - lo) == 30
        cover = torch.zeros(len(target)); cover[torch.from_numpy(rows)] = 1; dist.all_reduce(cover)
        ok_d = bool((cover == 1).all())
        ret[rank] = (ok_a, ok_b, ok_c, ok_d)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding_and_separator_exchange():
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: (True, True, True, True), 1: (True, True, True, True)}, dict(ret)


def test_shard_helpers_cover_and_balance():
    sys.path.insert(0, ROOT)
    from rome_jl_amd.distributed import shard_range, owner_of
    for n in (1, 7, 8, 3500, 10907):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            for i in range(0, n, max(1, n // 50)):
                a, b = spans[owner_of(i, n, w)]
                assert a <= i < b
