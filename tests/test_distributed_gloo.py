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


# ---------------------------------------------------------------------------------------------------------
# PipelinedSegmentSweep (the bench's N>1 driver) on CPU: a stand-in "device graph" whose launch plan runs the
# ORACLE on torch CPU tensors (test infrastructure), so the double-buffered separator pipeline -- mirror rows ->
# send buffer -> all_gather -> ghost blocks in the tail of the belief store -> cut factors -- is exercised for real
# over gloo with 2 ranks and compared with a single-process emulation of the same schedule.
def _segment_problem(R, rank, N):
    fg = R.synth_manhattan(P=40, loops=3, seed=100 + rank, N=N)
    cov = np.diag([1 / 44.6, 1 / 399.0, 1 / 9591.0])
    fg.addVariable("ghost_prev", R.Pose2); fg.addVariable("ghost_next", R.Pose2)
    fg.addFactor(["ghost_prev", "x0"], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], cov)))
    fg.addFactor(["x39", "ghost_next"], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], cov)))
    R.dead_reckon_init(fg, seed=11 + rank)
    return fg


class _OracleDG:
    """Just enough of DeviceGraph for PipelinedSegmentSweep, on CPU tensors, compute through the oracle."""

    def __init__(self, R, fg, stream_base):
        import oracle as ro
        self.torch, self.N, self.R, self.ro = torch, fg.N, R, ro
        pk = R.PackedGraph(fg)
        self.packed = pk
        factor, dr, fixed, target = R.PackedGraph.conv_table(pk.p2p2)
        F, P = pk.p2p2["F"], pk.prior2["F"]
        self.mu = np.concatenate([pk.p2p2["mu"], pk.prior2["mu"]]); cov = np.concatenate([pk.p2p2["cov"], pk.prior2["cov"]])
        self.L = np.array([ro.cholesky_lower(c) for c in cov])
        i32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int32))
        self.tab = {"p2p2": dict(F=F, P=P, C=2 * F + P, C_rel=2 * F, factor=i32(np.concatenate([factor, F + np.arange(P)])),
                                 dir=i32(np.concatenate([dr, np.full(P, 2)])), fixed=i32(np.concatenate([fixed, pk.prior2["var"]])),
                                 target=i32(np.concatenate([target, pk.prior2["var"]])), mu=None, L=None)}
        self.bel = {R.Pose2: torch.as_tensor(pk.beliefs(fg, R.Pose2))}
        self._lib = type("L", (), {"rome_conv_pose2pose2_dev": None})()
        self.stream_base = stream_base

    def _plan(self, fn, opts, **kw):
        ro, N = self.ro, self.N
        fixed, target, store, out = kw["fixed_var"].numpy(), kw["target_var"].numpy(), kw["bel_fixed"], kw["out"]
        factor, dr = kw["factor"].numpy(), kw["dir"].numpy()
        mirror_rows, mirror_out = kw["mirror_row"], kw["mirror_out"]
        rel = dr != 2

        def launch():
            o = ro.make_opts(N=N, solver=ro.SOLVER_NEWTON, seed=5, stream_offset=self.stream_base)
            res = np.zeros((len(dr), 3, N))
            res[rel] = ro.conv_pose2pose2(o, self.mu, self.L, store.numpy(), fixed[rel], target[rel], dr[rel], factor=factor[rel])
            # conv_pose2pose2 numbers its Philox streams by position in the call: re-run the prior rows on their own ids
            for k in np.nonzero(~rel)[0]:
                ok = ro.make_opts(N=N, seed=5, stream_offset=self.stream_base + int(k))
                res[k] = ro.sample_priorpose2(ok, self.mu[factor[k]], self.L[factor[k]])[0]
            out.copy_(torch.as_tensor(res))
            for m, r in enumerate(mirror_rows):
                mirror_out[m].copy_(out[r])
        return launch


def _pipe_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rome_jl_amd as R
        from rome_jl_amd.distributed import PipelinedSegmentSweep
        N = 16
        fg = _segment_problem(R, rank, N)
        dg = _OracleDG(R, fg, stream_base=rank << 32)
        pk = dg.packed
        sep_rows = [1, 2 * (40 - 2)]       # x0 <- (x0->x1, dir 1) ; x39 <- (x38->x39, dir 0)
        pipe = PipelinedSegmentSweep(dg, None, dist, world, rank, sep_rows, pk.index["ghost_prev"], pk.index["ghost_next"])
        hist = []
        for k in range(5):
            pipe.step()
            hist.append(pipe.prop.clone())
        pipe.drain()
        ret[rank] = [h.numpy() for h in hist]
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_pipelined_segment_sweep_two_ranks_matches_single_process_emulation():
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_pipe_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    # single-process emulation of the same schedule: sweep k of rank r reads the separators its neighbours produced in sweep k-2
    sys.path.insert(0, ROOT)
    import rome_jl_amd as R
    N = 16
    dgs = []
    for r in range(world):
        fg = _segment_problem(R, r, N)
        dgs.append(_OracleDG(R, fg, stream_base=r << 32))
    sep_rows = [1, 2 * (40 - 2)]
    init_ghost = [(d.bel[R.Pose2][d.packed.index["ghost_prev"]].clone(), d.bel[R.Pose2][d.packed.index["ghost_next"]].clone()) for d in dgs]
    props = [[] for _ in range(world)]
    for k in range(5):
        for r, d in enumerate(dgs):
            pk = d.packed
            store = d.bel[R.Pose2].clone()
            if k >= 2:
                store[pk.index["ghost_prev"]] = torch.as_tensor(props[(r - 1) % world][k - 2][sep_rows[1]])
                store[pk.index["ghost_next"]] = torch.as_tensor(props[(r + 1) % world][k - 2][sep_rows[0]])
            else:
                store[pk.index["ghost_prev"]], store[pk.index["ghost_next"]] = init_ghost[r]
            tb = d.tab["p2p2"]
            out = torch.zeros((tb["C"], 3, N), dtype=torch.float64)
            d._plan(None, None, fixed_var=tb["fixed"], target_var=tb["target"], factor=tb["factor"], dir=tb["dir"],
                    bel_fixed=store, out=out, mirror_row=(), mirror_out=None)()
            props[r].append(out.numpy())
    for r in range(world):
        for k in range(5):
            assert np.array_equal(ret[r][k], props[r][k]), (r, k)
    # the cut factors really see the neighbour: sweep 2 differs from what the initial ghosts would give
    assert not np.array_equal(ret[0][2], ret[0][0])


# ------------------------------------------------------------------ row-sharded linearisation of the parametric solver
def _cpu_linearize(kind, mu, W, xa, xb=None, ctx=None):
    """CPU stand-in (test infrastructure) for one rank's `rome_linearize` on PriorPose2 / Pose2Pose2 rows: whitened residual and
    Jacobians in the tangent convention of the solver (x ⊕ δ = (t + δt, θ + δθ))."""
    sys.path.insert(0, ROOT)
    from rome_jl_amd import _lib
    wrap = lambda a: np.arctan2(np.sin(a), np.cos(a))
    F = len(mu)
    if kind == _lib.FACTOR_PRIORPOSE2:
        r = np.stack([mu[:, 0] - xa[:, 0], mu[:, 1] - xa[:, 1], wrap(mu[:, 2] - xa[:, 2])], 1)
        Ja = np.tile(-np.eye(3), (F, 1, 1)); Jb = None
    elif kind == _lib.FACTOR_POSE2POSE2:
        c, s = np.cos(xa[:, 2]), np.sin(xa[:, 2])
        r = np.stack([xa[:, 0] + c * mu[:, 0] - s * mu[:, 1] - xb[:, 0], xa[:, 1] + s * mu[:, 0] + c * mu[:, 1] - xb[:, 1],
                      wrap(xa[:, 2] + mu[:, 2] - xb[:, 2])], 1)
        Ja = np.tile(np.eye(3), (F, 1, 1)); Ja[:, 0, 2] = -s * mu[:, 0] - c * mu[:, 1]; Ja[:, 1, 2] = c * mu[:, 0] - s * mu[:, 1]
        Jb = np.tile(-np.eye(3), (F, 1, 1))
    else:
        raise NotImplementedError(kind)
    W = np.asarray(W)
    return np.einsum("fij,fj->fi", W, r), W @ Ja, (None if Jb is None else W @ Jb)


def _lin_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rome_jl_amd as R
        from rome_jl_amd.distributed import LinearizeShard
        fg = R.synth_manhattan(P=150, loops=45, seed=21)
        shard = LinearizeShard(torch, dist, world, rank, device="cpu", kernel=_cpu_linearize)
        x = R.solveGraphParametric(fg, shard=shard)
        ret[rank] = np.array([x["x%d" % k] for k in range(150)])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_linearisation_gives_the_unsharded_solution(world):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_lin_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import rome_jl_amd as R
    from rome_jl_amd.distributed import LinearizeShard

    class _NoDist:   # world 1: the same code path without a process group
        class ReduceOp:
            MAX = None

        @staticmethod
        def all_reduce(t, op=None):
            return None

        @staticmethod
        def all_gather_into_tensor(out, inp):
            out.copy_(inp)
    fg = R.synth_manhattan(P=150, loops=45, seed=21)
    x = R.solveGraphParametric(fg, shard=LinearizeShard(torch, _NoDist, 1, 0, kernel=_cpu_linearize))
    ref = np.array([x["x%d" % k] for k in range(150)])
    gt = np.array([fg.ground_truth["x%d" % k] for k in range(150)])
    assert np.sqrt(((ref[:, :2] - gt[:, :2]) ** 2).sum(1).mean()) < 1.0       # the stand-in solves the graph
    for r in range(world):
        assert np.array_equal(ret[r], ref), r                                   # every rank, bit for bit
