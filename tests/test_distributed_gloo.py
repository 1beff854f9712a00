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
    try:
        fg = R.synth_manhattan(P=40, loops=3, seed=100 + rank, N=N)
    except RuntimeError:                       # (a 40-pose walk that never comes back to a cell: an open chain segment)
        fg = R.synth_manhattan(P=40, loops=0, seed=100 + rank, N=N)
    cov = np.diag([1 / 44.6, 1 / 399.0, 1 / 9591.0])
    fg.addVariable("ghost_prev", R.Pose2); fg.addVariable("ghost_next", R.Pose2)
    fg.addFactor(["ghost_prev", "x0"], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], cov)))
    fg.addFactor(["x39", "ghost_next"], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], cov)))
    R.dead_reckon_init(fg, seed=11 + rank)
    return fg


def _OracleDG(R, fg, stream_base):
    """CPU stand-in for DeviceGraph (tests/dist_standin.py): the launch plan runs the ORACLE on torch CPU tensors."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_standin import OracleDG
    dg = OracleDG(R, fg, seed=5)
    dg.stream_base = stream_base
    return dg


def _pipe_worker(rank, world, port, ret, depth=2, steps=5):
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
        import oracle as ro
        pipe = PipelinedSegmentSweep(dg, ro.make_opts(N=N, stream_offset=rank << 32), dist, world, rank, sep_rows,
                                     pk.index["ghost_prev"], pk.index["ghost_next"], depth=depth)
        hist = []
        for k in range(steps):
            pipe.step()
            hist.append(pipe.prop.clone())
        pipe.drain()
        ret[rank] = [h.numpy() for h in hist]
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,depth,steps", [(2, 4, 9), (3, 4, 9), (8, 4, 6)])
def test_pipelined_segment_sweep_two_ranks_matches_single_process_emulation(world, depth, steps):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_pipe_worker, args=(world, _free_port(), ret, depth, steps), nprocs=world, join=True)
    # single-process emulation of the same schedule: sweep k of rank r reads the separators its neighbours produced in sweep k-depth
    # (bench.py runs four slots on more than two ranks)
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
    for k in range(steps):
        for r, d in enumerate(dgs):
            pk = d.packed
            store = d.bel[R.Pose2].clone()
            if k >= depth:
                store[pk.index["ghost_prev"]] = torch.as_tensor(props[(r - 1) % world][k - depth][sep_rows[1]])
                store[pk.index["ghost_next"]] = torch.as_tensor(props[(r + 1) % world][k - depth][sep_rows[0]])
            else:
                store[pk.index["ghost_prev"]], store[pk.index["ghost_next"]] = init_ghost[r]
            tb = d.family_table("p2p2")
            out = torch.zeros((tb["n"], 3, N), dtype=torch.float64)
            import oracle as ro
            d._plan("p2p2", ro.make_opts(N=N, stream_offset=r << 32), rows4=tb["rows4"], mu=tb["mu"], L=tb["L"],
                    bel_fixed=store, bel_target=store, out=out)()
            props[r].append(out.numpy())
    for r in range(world):
        for k in range(steps):
            assert np.array_equal(ret[r][k], props[r][k]), (r, k)
    # the cut factors really see the neighbour: sweep `depth` differs from what the initial ghosts would give
    assert not np.array_equal(ret[0][depth], ret[0][0])


# ---------------------------------------------------------------------------------------------------------
# SeparatorPipeline on a bearing-range graph (BASELINE configs[3]/[4] shape: poses + landmarks, both bearing-range directions and
# the odometry family in one step, ONE all-gather carrying a Pose2 and a Point2 separator), 2 ranks over gloo against a
# single-process emulation of the same schedule.
def _br_segment(R, rank, N):
    rng = np.random.default_rng(40 + rank)
    fg = R.initfg(N)
    P = 12
    fg.addVariable("x0", R.Pose2)
    fg.addFactor(["x0"], R.PriorPose2(R.MvNormal([10.0 * rank, 0.0, 0.0], np.diag([0.01, 0.01, 0.0025]))))
    cov = np.diag(np.square([0.15, 0.12, 0.0106]))
    for k in range(1, P):
        fg.addVariable("x%d" % k, R.Pose2)
        fg.addFactor(["x%d" % (k - 1), "x%d" % k], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.1 * (k % 3 - 1)], cov)))
    for j in range(3):
        fg.addVariable("l%d" % j, R.Point2)
    # ghosts: the neighbour's last pose and a landmark that the neighbour also sights
    fg.addVariable("ghost_pose", R.Pose2); fg.addVariable("ghost_lm", R.Point2)
    fg.addFactor(["ghost_pose", "x0"], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], cov)))
    sight = [("x1", "l0"), ("x2", "l0"), ("x4", "l1"), ("x6", "l1"), ("x9", "l2"), ("x10", "l2"), ("x11", "ghost_lm"), ("x3", "ghost_lm")]
    for xp, lm in sight:
        fg.addFactor([xp, lm], R.Pose2Point2BearingRange(R.Normal(float(rng.uniform(-1, 1)), 0.03), R.Normal(float(rng.uniform(3, 8)), 0.5)))
    # a data-association ambiguity whose alternative is the GHOST landmark (IIF multihypo, test/testMultimodalRangeBearing.jl:53): the
    # alternative column of both bearing-range tables must follow the ghost into the receive buffers
    fg.addFactor(["x7", "l1", "ghost_lm"], R.Pose2Point2BearingRange(R.Normal(0.3, 0.03), R.Normal(5.0, 0.5)), multihypo=[1.0, 0.6, 0.4])
    R.dead_reckon_init(fg, seed=3 + rank)
    for j, l in enumerate(["l0", "l1", "l2", "ghost_lm"]):
        fg.initVariable(l, np.array([[2.0 + 3 * j], [4.0]]) + rng.standard_normal((2, N)))
    return fg


def _br_layout(pk, d):
    """published rows: Pose2 slot 0 = proposal of x11 from its odometry (p2p2 row), Point2 slot 0 = proposal of l2 from x10 (br0 row)"""
    f_last = int(np.nonzero((pk.p2p2["var_from"] == pk.index["x10"]) & (pk.p2p2["var_to"] == pk.index["x11"]))[0][0])
    r0 = d.family_table("br0")["rows4"].numpy()
    row_l2 = int(np.nonzero((r0[:, 2] == pk.index["x10"]) & (r0[:, 3] == pk.index["l2"]))[0][0])
    return 2 * f_last, row_l2


def _br_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rome_jl_amd as R
        import oracle as ro
        from rome_jl_amd.distributed import SeparatorPipeline
        N = 16
        fg = _br_segment(R, rank, N)
        dg = _OracleDG(R, fg, 0)
        pk = dg.packed
        row_pose, row_lm = _br_layout(pk, dg)
        pipe = SeparatorPipeline(dg, ro.make_opts(N=N, stream_offset=rank << 32), dist, world, rank,
                                 publish=[("p2p2", row_pose), ("br0", row_lm)],
                                 ghosts=[(R.Pose2, pk.index["ghost_pose"], rank - 1, 0), (R.Point2, pk.index["ghost_lm"], rank - 1, 0)])
        hist = []
        for k in range(4):
            pipe.step()
            hist.append({f: pipe.out[k % 2][f].clone().numpy() for f in pipe.families})
        pipe.drain()
        ret[rank] = (hist, bool(np.array_equal(dg.bel[R.Pose2].numpy(), pk.beliefs(fg, R.Pose2))))   # dg.bel untouched
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_separator_pipeline_bearing_range_two_ranks_matches_emulation():
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_br_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import rome_jl_amd as R
    import oracle as ro
    N = 16
    dgs = [_OracleDG(R, _br_segment(R, r, N), 0) for r in range(world)]
    lay = [_br_layout(d.packed, d) for d in dgs]
    hist = [[] for _ in range(world)]
    for k in range(4):
        for r, d in enumerate(dgs):
            pk = d.packed
            bel = {vt: d.bel[vt].clone() for vt in (R.Pose2, R.Point2)}
            if k >= 2:   # step k reads what the previous rank published in step k-2
                src = (r - 1) % world
                bel[R.Pose2][pk.index["ghost_pose"]] = torch.as_tensor(hist[src][k - 2]["p2p2"][lay[src][0]])
                bel[R.Point2][pk.index["ghost_lm"]] = torch.as_tensor(hist[src][k - 2]["br0"][lay[src][1]])
            outs = {}
            for f in d.families():
                tb = d.family_table(f)
                out = torch.zeros((tb["n"], tb["vt_target"].dim, N), dtype=torch.float64)
                mh = {} if tb["alt"] is None else dict(alt_var=tb["alt"], hypo_w=tb["w"])
                d._plan(tb["fn"], ro.make_opts(N=N, stream_offset=r << 32), rows4=tb["rows4"], mu=tb["mu"], L=tb["L"],
                        bel_fixed=bel[tb["vt_fixed"]], bel_target=bel[tb["vt_target"]], out=out, **mh)()
                outs[f] = out.numpy()
            hist[r].append(outs)
    for r in range(world):
        got, untouched = ret[r]
        assert untouched
        for k in range(4):
            for f in ("p2p2", "br1", "br0"):
                assert np.array_equal(got[k][f], hist[r][k][f]), (r, k, f)
    # the ghost landmark really arrives: the pose proposals of the ghost sighting change once the message is in
    assert not np.array_equal(ret[0][0][2]["br1"], ret[0][0][0]["br1"])


# ---------------------------------------------------------------------------------------------------------
# TargetShardedSweep: strong scaling of ONE graph.  2 and 3 ranks sweep disjoint row ranges of the target-sorted table and
# all-gather the owned belief blocks; the assembled proposal table equals the unsharded sweep of the same sorted table.
def _strong_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rome_jl_amd as R
        import oracle as ro
        from rome_jl_amd.distributed import TargetShardedSweep
        N = 16
        fg = R.synth_manhattan(P=50, loops=20, seed=9, N=N)
        R.dead_reckon_init(fg, seed=2)
        dg = _OracleDG(R, fg, 0)
        sh = TargetShardedSweep(dg, ro.make_opts(N=N, stream_offset=7), dist, world, rank)
        V = dg.bel[R.Pose2].shape[0]
        sh.step(); sh.wait()
        t = sh.prop.clone()
        dist.all_reduce(t)                               # disjoint row ranges -> the sum assembles the table
        same_store = bool(np.array_equal(sh.store[:V].numpy(), dg.bel[R.Pose2].numpy()))   # the all-gather of unchanged blocks is a no-op
        sh.mine.mul_(0.0).add_(float(rank + 1))          # pretend a product updated the owned beliefs ...
        sh.exchange(); sh.wait()                         # ... and publish them
        owners_ok = all(bool((sh.store[r * sh.q:min((r + 1) * sh.q, V)] == float(r + 1)).all()) for r in range(world))
        ret[rank] = (t.numpy(), sh.order.copy(), (sh.row_lo, sh.row_hi), owners_ok and same_store)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_target_sharded_sweep_assembles_the_unsharded_table(world):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_strong_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import rome_jl_amd as R
    import oracle as ro
    N = 16
    fg = R.synth_manhattan(P=50, loops=20, seed=9, N=N)
    R.dead_reckon_init(fg, seed=2)
    dg = _OracleDG(R, fg, 0)
    tb = dg.family_table("p2p2")
    rows = tb["rows4"].numpy()
    order = np.argsort(rows[:, 3], kind="stable")
    out = torch.zeros((tb["n"], 3, N), dtype=torch.float64)
    dg._plan("p2p2", ro.make_opts(N=N, stream_offset=7), rows4=torch.as_tensor(rows[order]), mu=tb["mu"], L=tb["L"],
             bel_fixed=dg.bel[R.Pose2], bel_target=dg.bel[R.Pose2], out=out)()
    spans = sorted(ret[r][2] for r in range(world))
    assert spans[0][0] == 0 and spans[-1][1] == tb["n"] and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    for r in range(world):
        table, order_r, _, owners_ok = ret[r]
        assert owners_ok and np.array_equal(order_r, order)
        assert np.array_equal(table, out.numpy())        # sweep inputs are the initial beliefs: same table as the unsharded sweep


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


# ---------------------------------------------------------------------------------------------------------
# BASELINE configs[3] sharded: the honeycomb lattice with multihypo re-sightings (R.synth_beehive_mh), one segment per rank, cut
# through the lattice: SIX pose separators and FIVE landmark separators per rank (more than the four rows the old mirror_row form
# could publish -> rome_conv_dev.mirror_map), both bearing-range directions with hypotheses, ONE all-gather per step.
_BH_POSES = (2, 5, 8, 11, 14, 17)
_BH_LMS = 5


def _beehive_segment(R, rank, N):
    rng = np.random.default_rng(90 + rank)
    fg = R.synth_beehive_mh(20, N=N)
    lms = [l for l, t in fg.variables.items() if t is R.Point2]
    leg = R.Pose2Pose2(R.MvNormal([10.0, 0.0, np.pi / 3], np.diag(np.square([0.1, 0.1, 0.1]))))
    sight = lambda: R.Pose2Point2BearingRange(R.Normal(0.0, 0.03), R.Normal(20.0, 0.5))
    for k, p in enumerate(_BH_POSES):      # cut legs: the neighbour's pose k enters as a ghost one leg before x_p
        fg.addVariable("gp%d" % k, R.Pose2)
        fg.addFactor(["gp%d" % k, "x%d" % p], leg)
    for k in range(_BH_LMS):               # cut sightings: a neighbour's landmark, twice as the ALTERNATIVE of an ambiguous sighting
        fg.addVariable("gl%d" % k, R.Point2)
        if k % 2 == 0:
            fg.addFactor(["x%d" % (3 * k + 1), lms[k], "gl%d" % k], sight(), multihypo=[1.0, 0.5, 0.5])
        else:
            fg.addFactor(["x%d" % (3 * k + 1), "gl%d" % k], sight())
    R.dead_reckon_init(fg, seed=7 + rank)
    sim = fg._sim
    for l, t in fg.variables.items():
        if t is R.Point2:
            c = np.asarray(sim[l]) if l in sim else np.array([5.0 * rank, 3.0])
            fg.initVariable(l, c[:, None] + 0.5 * rng.standard_normal((2, N)))
    return fg


def _beehive_layout(pk, d):
    """published rows: Pose2 slots = the proposals of x_p from their incoming honeycomb leg (p2p2 rows), Point2 slots = the first
    bearing-range -> landmark row of the first five landmarks"""
    rows_p = []
    for p in _BH_POSES:
        f = int(np.nonzero((pk.p2p2["var_from"] == pk.index["x%d" % (p - 1)]) & (pk.p2p2["var_to"] == pk.index["x%d" % p]))[0][0])
        rows_p.append(2 * f)
    r0 = d.family_table("br0")["rows4"].numpy()
    lm_idx = [pk.index[l] for l in pk.labels[__import__("rome_jl_amd").Point2][:_BH_LMS]]
    rows_l = [int(np.nonzero(r0[:, 3] == li)[0][0]) for li in lm_idx]
    return rows_p, rows_l


def _beehive_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rome_jl_amd as R
        import oracle as ro
        from rome_jl_amd.distributed import SeparatorPipeline
        N = 16
        fg = _beehive_segment(R, rank, N)
        dg = _OracleDG(R, fg, 0)
        pk = dg.packed
        rows_p, rows_l = _beehive_layout(pk, dg)
        pipe = SeparatorPipeline(dg, ro.make_opts(N=N, stream_offset=rank << 32), dist, world, rank,
                                 publish=[("p2p2", r) for r in rows_p] + [("br0", r) for r in rows_l],
                                 ghosts=[(R.Pose2, pk.index["gp%d" % k], rank - 1, k) for k in range(len(_BH_POSES))] +
                                        [(R.Point2, pk.index["gl%d" % k], rank - 1, k) for k in range(_BH_LMS)])
        hist = []
        for k in range(4):
            pipe.step()
            hist.append({f: pipe.out[k % 2][f].clone().numpy() for f in pipe.families})
        pipe.drain()
        ret[rank] = hist
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_beehive_multihypo_lattice_cut_two_ranks_matches_emulation():
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_beehive_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import rome_jl_amd as R
    import oracle as ro
    N = 16
    dgs = [_OracleDG(R, _beehive_segment(R, r, N), 0) for r in range(world)]
    lay = [_beehive_layout(d.packed, d) for d in dgs]
    assert any(d.family_table("br1")["alt"] is not None for d in dgs)
    hist = [[] for _ in range(world)]
    for k in range(4):
        for r, d in enumerate(dgs):
            pk = d.packed
            bel = {vt: d.bel[vt].clone() for vt in (R.Pose2, R.Point2)}
            if k >= 2:   # step k reads what the previous rank published in step k-2
                src = (r - 1) % world
                for s, row in enumerate(lay[src][0]):
                    bel[R.Pose2][pk.index["gp%d" % s]] = torch.as_tensor(hist[src][k - 2]["p2p2"][row])
                for s, row in enumerate(lay[src][1]):
                    bel[R.Point2][pk.index["gl%d" % s]] = torch.as_tensor(hist[src][k - 2]["br0"][row])
            outs = {}
            for f in d.families():
                tb = d.family_table(f)
                out = torch.zeros((tb["n"], tb["vt_target"].dim, N), dtype=torch.float64)
                mh = {} if tb["alt"] is None else dict(alt_var=tb["alt"], hypo_w=tb["w"])
                d._plan(tb["fn"], ro.make_opts(N=N, stream_offset=r << 32), rows4=tb["rows4"], mu=tb["mu"], L=tb["L"],
                        bel_fixed=bel[tb["vt_fixed"]], bel_target=bel[tb["vt_target"]], out=out, **mh)()
                outs[f] = out.numpy()
            hist[r].append(outs)
    for r in range(world):
        for k in range(4):
            for f in ("p2p2", "br1", "br0"):
                assert np.array_equal(ret[r][k][f], hist[r][k][f]), (r, k, f)
    # the ghosts really arrive: proposals through the cut legs change once the messages are in
    assert not np.array_equal(ret[0][2]["p2p2"], ret[0][0]["p2p2"]) and not np.array_equal(ret[1][2]["br1"], ret[1][0]["br1"])


# ---------------------------------------------------------------------------------------------------------
# TargetShardedSweep.solve_step: the unit that strong-scales -- sweep of the owned rows, manikde! bandwidths, multiscale Gibbs product
# of the owned variables, ONE all-gather of the changed beliefs -- over 3 ranks equals the single-rank sequence (partition-independent
# Philox streams: row index / global variable id).
def _solve_step_worker(rank, world, port, ret, P=40):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rome_jl_amd as R
        import oracle as ro
        from rome_jl_amd.distributed import TargetShardedSweep
        N = 16
        fg = R.synth_manhattan(P=P, loops=15, seed=9, N=N)
        R.dead_reckon_init(fg, seed=2)
        dg = _OracleDG(R, fg, 5)
        o = ro.make_opts(N=N, seed=5, stream_offset=11)
        sh = TargetShardedSweep(dg, o, dist, world, rank)
        for s in range(2):
            sh.solve_step(o, sweep=s)
        sh.wait()
        ret[rank] = (sh.store[:dg.bel[R.Pose2].shape[0]].numpy().copy(), (sh.row_lo, sh.row_hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,P", [(3, 40), (8, 41)])     # world 8, 41 poses: q = 6, rank 6 owns 5 variables, rank 7 NONE (an empty share)
def test_target_sharded_solve_step_any_world_equals_one_rank(world, P):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_solve_step_worker, args=(world, _free_port(), ret, P), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import rome_jl_amd as R
    import oracle as ro
    from rome_jl_amd.distributed import TargetShardedSweep
    N = 16
    fg = R.synth_manhattan(P=P, loops=15, seed=9, N=N)
    R.dead_reckon_init(fg, seed=2)
    dg = _OracleDG(R, fg, 5)
    o = ro.make_opts(N=N, seed=5, stream_offset=11)
    one = TargetShardedSweep(dg, o, None, 1, 0)
    before = one.store.clone()
    for s in range(2):
        one.solve_step(o, sweep=s)
    V = dg.bel[R.Pose2].shape[0]
    ref = one.store[:V].numpy()
    assert not np.array_equal(ref, before[:V].numpy())
    spans = sorted(ret[r][1] for r in range(world))
    assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1)) and spans[-1][1] == one.prop.shape[0]
    if world == 8:
        assert any(lo == hi for lo, hi in spans)                # the empty share really occurred
    for r in range(world):
        assert np.array_equal(ret[r][0], ref), r


# ---------------------------------------------------------------------------------------------------------
# FrontierShard (device-resident form): the independent cliques of a frontier dealt round-robin to the ranks, each up-solving its share
# through an up-solve PLAN over a resident store (here: oracle-backed stand-ins, tests/dist_standin.py), the new frontal beliefs mirrored
# into the rank's slice of the receive buffer, ONE all-gather in place, ONE scatter into the store.  Philox streams are positions in the
# WHOLE frontier's tables (plan_frontier), so ANY number of ranks -- 2, or 8 with empty shares -- reproduces the single unsharded plan
# bit for bit.
def _frontier_graph(R, N):
    fg = R.generateGraph_Hexagonal(N=N)
    R.dead_reckon_init(fg, seed=5)
    fg.initVariable("l1", np.array([[20.0], [0.0]]) + np.random.default_rng(1).standard_normal((2, N)))
    return fg


_FRONTIERS = [[["x0", "x1"], ["x3"], ["x5"]], [["x2"], ["x4"], ["x6", "l1"]]]   # two tree levels: the second reads what the first wrote


def _frontier_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rome_jl_amd as R
        from rome_jl_amd.distributed import FrontierShard
        from dist_standin import OracleStore, OraclePlan, OracleScatter
        N = 32
        fg = _frontier_graph(R, N)
        store = OracleStore(R, fg)
        sh = FrontierShard(store, torch, dist, world, rank, plan_cls=OraclePlan, scatter_cls=OracleScatter)
        plans = [sh.plan(f, gibbsIters=2) for f in _FRONTIERS]
        for p in range(2):                                     # two passes over the two levels: the plans are reused
            for k, pl in enumerate(plans):
                sh.step(pl, R.make_opts(N=N, seed=21 + p, stream_offset=(2 * p + k) << 40))
        ret[rank] = {l: store.get(l).copy() for f in _FRONTIERS for c in f for l in c}
    finally:
        dist.destroy_process_group()


def _frontier_reference(R, N):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_standin import OracleStore, OraclePlan
    fg = _frontier_graph(R, N)
    store = OracleStore(R, fg)
    plans = [OraclePlan(store, f, gibbsIters=2) for f in _FRONTIERS]      # ONE unsharded plan per frontier
    for p in range(2):
        for k, pl in enumerate(plans):
            pl.run(R.make_opts(N=N, seed=21 + p, stream_offset=(2 * p + k) << 40))
    return fg, {l: store.get(l).copy() for f in _FRONTIERS for c in f for l in c}


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 8])
def test_frontier_shard_any_world_equals_the_single_unsharded_plan(world):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_frontier_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import rome_jl_amd as R
    N = 32
    fg, ref = _frontier_reference(R, N)
    assert set(ref) == {"x0", "x1", "x2", "x3", "x4", "x5", "x6", "l1"}
    for r in range(world):                                     # (world 8: ranks 3..7 have empty shares of a 3-clique frontier)
        assert set(ret[r]) == set(ref)
        for l in ref:
            assert np.array_equal(ret[r][l], ref[l]), (world, r, l)
    assert not np.array_equal(ref["x3"], fg.getVal("x3"))


def test_frontier_shard_validates_independence_over_the_whole_frontier():
    """a factor linking frontals of two cliques is refused whichever rank owns them (plan_frontier checks the full clique list)"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rome_jl_amd as R
    from rome_jl_amd.distributed import FrontierShard
    from dist_standin import OracleStore, OraclePlan, OracleScatter
    fg = _frontier_graph(R, 16)
    for rank in range(2):
        sh = FrontierShard(OracleStore(R, fg), torch, None, 2, rank, plan_cls=OraclePlan, scatter_cls=OracleScatter)
        with pytest.raises(ValueError):
            sh.plan([["x0"], ["x1"]])                          # x0 -- x1 share an odometry factor; each rank owns one of them


# ---------------------------------------------------------------------------------------------------------
# The LEVELS of a Bayes tree through FrontierShard (rome_jl_amd.tree.TreeSolver(shard=...)): multi-frontal cliques, separator copies,
# absolute and relative messages of the children, sampled-measurement rows -- every level dealt to the ranks by clique, the blocks it
# writes exchanged in the PACKED per-type layout (Pose2 3 slots, Point2 2), block operations on every rank.  Any world size reproduces
# the unsharded tree solve bit for bit (Philox streams = positions in the whole level's tables).
def _tree_graph(R, kind, N):
    if kind == "hexagon":            # Pose2 + a Point2 landmark among the separators (bearing-range relative messages)
        fg = R.generateGraph_Hexagonal(N=N)
        R.dead_reckon_init(fg, seed=5)
        fg.initVariable("l1", np.array([[20.0], [0.0]]) + np.random.default_rng(1).standard_normal((2, N)))
        return fg
    fg = R.initfg(N)                 # the first 40 poses of manhattan.g2o with the loop closures between them
    ids = set()
    rows = []
    for ln in open(os.path.join(ROOT, "tests", "golden", "manhattan.g2o")):
        t = ln.split()
        if t and t[0] == "EDGE_SE2" and int(t[1]) < 40 and int(t[2]) < 40:
            rows.append(t); ids.update((int(t[1]), int(t[2])))
    for k in sorted(ids):
        fg.addVariable("x%d" % k, R.Pose2)
    fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), np.diag([0.01, 0.01, 0.0025]))))
    for t in rows:
        u = [float(x) for x in t[6:12]]
        info = np.array([[u[0], u[1], u[2]], [u[1], u[3], u[4]], [u[2], u[4], u[5]]])
        C = np.linalg.inv(info)
        fg.addFactor(["x%s" % t[1], "x%s" % t[2]], R.Pose2Pose2(R.MvNormal(np.array([float(x) for x in t[3:6]]), 0.5 * (C + C.T))))
    R.dead_reckon_init(fg, seed=3)
    return fg


def _tree_solve(R, kind, messages, N, shard=None):
    from rome_jl_amd.tree import TreeSolver
    from dist_standin import OracleTreeBackend
    fg = _tree_graph(R, kind, N)
    if messages == "elimination":       # variable elimination in relative-factor algebra: every round's plans dealt to the ranks by variable,
        from rome_jl_amd.elimination import RelativeEliminationSolver   # compose / anchor / mix block operations on every rank
        ts = RelativeEliminationSolver(fg, backend=OracleTreeBackend(R), shard=shard, structures=2)
    else:
        ts = TreeSolver(fg, messages=messages, backend=OracleTreeBackend(R), shard=shard, gibbsIters=2)
    ts.upload()
    ts.solve(R.make_opts(N=N, seed=9), passes=2)
    return ts, {l: ts.store.get(l).copy() for l in fg.variables}


def _tree_worker(rank, world, port, kind, messages, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rome_jl_amd as R
        from rome_jl_amd.distributed import FrontierShard
        from dist_standin import OracleTreeScatter
        ts, out = _tree_solve(R, kind, messages, 16, shard=lambda store: FrontierShard(store, torch, dist, world, rank, scatter_cls=OracleTreeScatter))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,kind,messages", [(2, "manhattan", "relative"), (8, "manhattan", "relative"), (2, "hexagon", "relative"), (2, "manhattan", "marginal"),
                                                 (2, "manhattan", "elimination"), (8, "manhattan", "elimination")])
def test_tree_levels_sharded_by_clique_equal_the_unsharded_tree_solve(world, kind, messages):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_tree_worker, args=(world, _free_port(), kind, messages, ret), nprocs=world, join=True)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rome_jl_amd as R
    ts, ref = _tree_solve(R, kind, messages, 16)
    if messages == "elimination":
        assert ts.stats()["rounds"] >= 3 and ts.stats()["compositions"] > 5 and ts.passes_pooled == 2
    else:
        assert len(ts.tree.levels) >= 3 and any(len(c.frontals) + len(c.separators) >= 3 for c in ts.tree.cliques)
    if messages == "relative":
        assert ts.stats()["relative_messages"] > 0
    for r in range(world):
        assert set(ret[r]) == set(ref)
        for l in ref:
            assert np.array_equal(ret[r][l], ref[l]), (world, kind, r, l)
