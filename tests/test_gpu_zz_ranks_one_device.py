"""The N > 1 drivers with the REAL kernels in TWO PROCESSES.  The test box has one GPU and RCCL refuses two ranks on one device, so
both ranks run their launches on cuda:0 and the communicator is `rome_jl_amd.rccl.HostStagedComm`: RcclComm's interface
(`all_gather_f64(send_ptr, recv_ptr, count, stream)`) with device -> host -> gloo -> host -> device as the transport.  Everything else is the shipped path: rank-
dependent tables and shares, ghost blocks, mirror writes of the sweep / product kernels into the exchange buffer, the in-place receive
layout, the scatter plan, the depth-fold buffered pipeline.  What the RCCL transport itself adds is covered by the one-rank direct-RCCL
tests (test_gpu_pipeline.py, test_gpu_clique_upsolve.py, test_gpu_config4.py); the same drivers over oracle stand-ins at world 2 / 3 / 8 run in
tests/test_distributed_gloo.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def HostStagedComm(torch, dist, world, ctx):
    from rome_jl_amd.rccl import HostStagedComm as H
    return H(torch, dist, world, ctx)


def _init(rank, world, port):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import rome_jl_amd as R
    return torch, dist, R


# ---------------------------------------------------------------------------------------------------------------------------------
# weak scaling: a ring of Manhattan-shaped segments, separator beliefs published by the sweep kernel itself (mirror_map)
def _segment(R, rank, N):
    try:
        fg = R.synth_manhattan(P=40, loops=3, seed=100 + rank, N=N)
    except RuntimeError:
        fg = R.synth_manhattan(P=40, loops=0, seed=100 + rank, N=N)
    cov = np.diag([1 / 44.6, 1 / 399.0, 1 / 9591.0])
    fg.addVariable("ghost_prev", R.Pose2); fg.addVariable("ghost_next", R.Pose2)
    fg.addFactor(["ghost_prev", "x0"], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], cov)))
    fg.addFactor(["x39", "ghost_next"], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], cov)))
    R.dead_reckon_init(fg, seed=11 + rank)
    return fg


_SEP = [1, 2 * (40 - 2)]     # x0 <- (x0 -> x1, dir 1); x39 <- (x38 -> x39, dir 0)


def _pipe_worker(rank, world, port, ret, depth, steps, N):
    torch, dist, R = _init(rank, world, port)
    try:
        from rome_jl_amd.distributed import PipelinedSegmentSweep
        dev = torch.device("cuda", 0)
        fg = _segment(R, rank, N)
        ctx = R.Context(0)
        dg = R.DeviceGraph(fg, device=dev, ctx=ctx); dg.upload_beliefs(fg)
        pk = dg.packed
        comms = [HostStagedComm(torch, dist, world, ctx) for _ in range(depth)]
        o = R.make_opts(N=N, seed=9, stream_offset=rank << 32)
        pipe = PipelinedSegmentSweep(dg, o, dist, world, rank, _SEP, pk.index["ghost_prev"], pk.index["ghost_next"], depth=depth, rccl_comms=comms)
        assert pipe.comms is not None                              # the direct-communicator branch of step(), not the torch fallback
        hist = []
        for k in range(steps):
            pipe.step()
            torch.cuda.synchronize()
            hist.append(pipe.prop.cpu().numpy().copy())
        pipe.drain()
        ret[rank] = (hist, sum(c.calls for c in comms))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("depth,steps", [(2, 5), (4, 9)])
def test_pipelined_segment_sweep_two_processes_real_kernels(depth, steps):
    import torch
    import torch.multiprocessing as mp
    import rome_jl_amd as R
    world, N = 2, 100
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_pipe_worker, args=(world, _free_port(), ret, depth, steps, N), nprocs=world, join=True)
    # single-process emulation of the schedule with the same kernels: sweep k of rank r reads what its neighbours published in sweep k - depth
    dev = torch.device("cuda", 0)
    dgs = []
    for r in range(world):
        fg = _segment(R, r, N)
        dg = R.DeviceGraph(fg, device=dev); dg.upload_beliefs(fg)
        dgs.append(dg)
    props = [[] for _ in range(world)]
    for k in range(steps):
        for r, dg in enumerate(dgs):
            pk = dg.packed
            store = dg.bel[R.Pose2].clone()
            if k >= depth:
                store[pk.index["ghost_prev"]] = torch.as_tensor(props[(r - 1) % world][k - depth][_SEP[1]], device=dev)
                store[pk.index["ghost_next"]] = torch.as_tensor(props[(r + 1) % world][k - depth][_SEP[0]], device=dev)
            tb = dg.family_table("p2p2")
            out = torch.zeros((tb["n"], 3, N), dtype=torch.float64, device=dev)
            dg._plan(tb["fn"], R.make_opts(N=N, seed=9, stream_offset=r << 32), n_conv=tb["n"], dir_all=tb["dir_all"], rows4=tb["rows4"],
                     mu=tb["mu"], L=tb["L"], bel_fixed=store, bel_target=store, out=out)()
            torch.cuda.synchronize()
            props[r].append(out.cpu().numpy())
    for r in range(world):
        hist, calls = ret[r]
        assert calls == steps
        for k in range(steps):
            assert np.array_equal(hist[k], props[r][k]), (r, k)
    assert not np.array_equal(ret[0][0][depth], ret[0][0][0])      # the cut factors really see the neighbour's separator


# ---------------------------------------------------------------------------------------------------------------------------------
# strong scaling: one graph, rows sharded by target ownership, the solve iteration (sweep + manikde! bandwidths + multiscale Gibbs
# product of the owned variables) then ONE all-gather in place
def _solve_worker(rank, world, port, ret, P, N):
    torch, dist, R = _init(rank, world, port)
    try:
        from rome_jl_amd.distributed import TargetShardedSweep
        dev = torch.device("cuda", 0)
        fg = R.synth_manhattan(P=P, loops=15, seed=9, N=N)
        R.dead_reckon_init(fg, seed=2)
        ctx = R.Context(0)
        dg = R.DeviceGraph(fg, device=dev, ctx=ctx); dg.upload_beliefs(fg)
        o = R.make_opts(N=N, seed=5, stream_offset=11)
        comm = HostStagedComm(torch, dist, world, ctx)
        sh = TargetShardedSweep(dg, o, dist, world, rank, rccl_comm=comm)
        assert sh.comm is comm
        for s in range(2):
            sh.solve_step(o, sweep=s)
        sh.wait(); torch.cuda.synchronize()
        ret[rank] = (sh.store[:dg.bel[R.Pose2].shape[0]].cpu().numpy().copy(), (sh.row_lo, sh.row_hi), comm.calls)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,P", [(2, 40), (3, 41)])
def test_target_sharded_solve_step_processes_equal_one_rank_real_kernels(world, P):
    import torch
    import torch.multiprocessing as mp
    import rome_jl_amd as R
    from rome_jl_amd.distributed import TargetShardedSweep
    N = 100
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_solve_worker, args=(world, _free_port(), ret, P, N), nprocs=world, join=True)
    dev = torch.device("cuda", 0)
    fg = R.synth_manhattan(P=P, loops=15, seed=9, N=N)
    R.dead_reckon_init(fg, seed=2)
    dg = R.DeviceGraph(fg, device=dev); dg.upload_beliefs(fg)
    o = R.make_opts(N=N, seed=5, stream_offset=11)
    one = TargetShardedSweep(dg, o, None, 1, 0)
    before = one.store.clone()
    for s in range(2):
        one.solve_step(o, sweep=s)
    torch.cuda.synchronize()
    V = dg.bel[R.Pose2].shape[0]
    ref = one.store[:V].cpu().numpy()
    assert not np.array_equal(ref, before[:V].cpu().numpy())
    spans = sorted(ret[r][1] for r in range(world))
    assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1)) and spans[-1][1] == one.prop.shape[0]
    for r in range(world):
        assert ret[r][2] == 2                                       # one exchange per solve iteration
        assert np.array_equal(ret[r][0], ref), r                    # any number of ranks == one rank, bit for bit


# ---------------------------------------------------------------------------------------------------------------------------------
# the clique frontier: shares of independent cliques, up-solve plans over a resident store, mirror -> all-gather in place -> scatter
_FRONTIERS = [[["x0", "x1"], ["x3"], ["x5"]], [["x2"], ["x4"], ["x6", "l1"]]]


def _hex(R, N):
    fg = R.generateGraph_Hexagonal(N=N)
    R.dead_reckon_init(fg, seed=5)
    fg.initVariable("l1", np.array([[20.0], [0.0]]) + np.random.default_rng(1).standard_normal((2, N)))
    return fg


def _frontier_worker(rank, world, port, ret, N):
    torch, dist, R = _init(rank, world, port)
    try:
        from rome_jl_amd.distributed import FrontierShard
        from rome_jl_amd.clique import DeviceStore
        dev = torch.device("cuda", 0)
        fg = _hex(R, N)
        ctx = R.Context(0)
        store = DeviceStore(fg, ctx=ctx)
        comm = HostStagedComm(torch, dist, world, ctx)
        sh = FrontierShard(store, torch, dist, world, rank, device=dev, comm=comm)
        plans = [sh.plan(f, gibbsIters=2) for f in _FRONTIERS]
        for p in range(2):
            for k, pl in enumerate(plans):
                sh.step(pl, R.make_opts(N=N, seed=21 + p, stream_offset=(2 * p + k) << 40))
        torch.cuda.synchronize()
        ret[rank] = ({l: store.get(l).copy() for f in _FRONTIERS for c in f for l in c}, comm.calls)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])       # world 4: rank 3 has an empty share of a three-clique frontier
def test_frontier_shard_processes_equal_the_single_unsharded_plan_real_kernels(world):
    import torch
    import torch.multiprocessing as mp
    import rome_jl_amd as R
    from rome_jl_amd.clique import DeviceStore, UpsolvePlan
    N = 100
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_frontier_worker, args=(world, _free_port(), ret, N), nprocs=world, join=True)
    fg = _hex(R, N)
    store = DeviceStore(fg)
    plans = [UpsolvePlan(store, f, gibbsIters=2) for f in _FRONTIERS]     # ONE unsharded plan per frontier
    for p in range(2):
        for k, pl in enumerate(plans):
            pl.run(R.make_opts(N=N, seed=21 + p, stream_offset=(2 * p + k) << 40))
    torch.cuda.synchronize()
    ref = {l: store.get(l).copy() for f in _FRONTIERS for c in f for l in c}
    assert not np.array_equal(ref["x3"], fg.getVal("x3"))
    for r in range(world):
        got, calls = ret[r]
        assert calls == 4
        for l in ref:
            assert np.array_equal(got[l], ref[l]), (world, r, l)


# ---------------------------------------------------------------------------------------------------------------------------------
# The LEVELS of a Bayes tree sharded by clique with the REAL kernels (rome_jl_amd.tree.TreeSolver(shard=FrontierShard)): multi-frontal
# cliques, separator copies, store-resident and sampled-measurement messages; packed per-type exchange layout; block operations on
# every rank.  2 and 4 processes on one device == the unsharded device solve, bit for bit.
def _tree_fg(R, N):
    fg = R.initfg(N)
    rows, ids = [], set()
    for ln in open(os.path.join(ROOT, "tests", "golden", "manhattan.g2o")):
        t = ln.split()
        if t and t[0] == "EDGE_SE2" and int(t[1]) < 60 and int(t[2]) < 60:
            rows.append(t); ids.update((int(t[1]), int(t[2])))
    for k in sorted(ids):
        fg.addVariable("x%d" % k, R.Pose2)
    fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), np.diag([0.01, 0.01, 0.0025]))))
    for t in rows:
        u = [float(x) for x in t[6:12]]
        C = np.linalg.inv(np.array([[u[0], u[1], u[2]], [u[1], u[3], u[4]], [u[2], u[4], u[5]]]))
        fg.addFactor(["x%s" % t[1], "x%s" % t[2]], R.Pose2Pose2(R.MvNormal(np.array([float(x) for x in t[3:6]]), 0.5 * (C + C.T))))
    R.dead_reckon_init(fg, seed=3)
    return fg


def _tree_worker(rank, world, port, ret, N, messages):
    torch, dist, R = _init(rank, world, port)
    try:
        from rome_jl_amd.distributed import FrontierShard
        from rome_jl_amd.tree import TreeSolver
        dev = torch.device("cuda", 0)
        ctx = R.Context(0)
        fg = _tree_fg(R, N)
        box = {}

        def shard(store):
            box["comm"] = HostStagedComm(torch, dist, world, ctx)
            return FrontierShard(store, torch, dist, world, rank, device=dev, comm=box["comm"])
        if messages == "elimination":
            from rome_jl_amd.elimination import RelativeEliminationSolver
            ts = RelativeEliminationSolver(fg, ctx=ctx, shard=shard)
        else:
            ts = TreeSolver(fg, messages=messages, ctx=ctx, shard=shard)
        ts.upload(); ts.solve(R.make_opts(N=N, seed=13), passes=2)
        torch.cuda.synchronize()
        ret[rank] = ({l: ts.store.get(l).copy() for l in fg.variables}, box["comm"].calls)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,messages", [(2, "relative"), (4, "relative"), (2, "marginal"), (8, "relative"), (4, "elimination")])
def test_tree_levels_sharded_by_clique_processes_equal_the_unsharded_tree_solve_real_kernels(world, messages):
    import torch
    import torch.multiprocessing as mp
    import rome_jl_amd as R
    from rome_jl_amd.tree import TreeSolver
    N = 64
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_tree_worker, args=(world, _free_port(), ret, N, messages), nprocs=world, join=True)
    fg = _tree_fg(R, N)
    if messages == "elimination":
        from rome_jl_amd.elimination import RelativeEliminationSolver
        ts = RelativeEliminationSolver(fg)
    else:
        ts = TreeSolver(fg, messages=messages)
    ts.upload(); ts.solve(R.make_opts(N=N, seed=13), passes=2)
    torch.cuda.synchronize()
    if messages == "elimination":
        n_ex = ts.stats()["plan_steps"]
        assert ts.stats()["rounds"] >= 3
    else:
        n_ex = len(ts.tree.levels)
        assert len(ts.tree.levels) > 5 and (messages == "marginal" or ts.stats()["relative_messages"] > 5)
    for r in range(world):
        got, calls = ret[r]
        assert calls > n_ex                          # one exchange per level / plan step and pass
        for l in fg.variables:
            assert np.array_equal(got[l], ts.store.get(l)), (world, r, l)


# ---------------------------------------------------------------------------------------------------------------------------------
# bench.py under the launcher, exactly the driver's command line for N = 2 and N = 8, as N real processes (ROME_BENCH_SHARED_DEVICE=1: both on
# device 0, gloo process group, host-staged exchange): per-rank graph segments and tables, the pipeline, the barriers, the
# max-over-ranks reduction and rank 0's single JSON line all execute -- the flow an 8-GPU run takes, which no box available here can make.
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("scaling,world", [("weak", 2), ("strong", 2), ("weak", 8)])
def test_bench_under_torchrun_ranks_share_one_device(scaling, world):
    import json
    import subprocess
    env = dict(os.environ, ROME_BENCH_SHARED_DEVICE="1", ROME_BENCH_WATCHDOG_S="600")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "5",
           "--settle-launches", "50", "--scaling", scaling]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1100, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                       # ONE JSON line, printed by rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == scaling
    assert d["config"]["ranks_seen_by_rccl"] == world
    assert "NOT a measurement" in d["data"]
    assert d["value"] > 0 and d["ms_per_step"] > 0 and len(d["timed_blocks_ms_per_step"]) == 9
    assert "host-staged" in d["config"]["parallelism"]
    per_gpu = d["config"]["convolutions_per_step_per_gpu"]
    if scaling == "weak":      # whole-job units: both ranks' segments (10 907 + the two cut-edge rows each)
        assert per_gpu >= 10907
        assert abs(d["value"] - world * per_gpu * 20 / (d["ms_per_step"] * 20 * 1e-3)) <= 1e-6 * d["value"]
    else:                      # ONE graph: the rows are split between the ranks
        assert 0 < per_gpu < 10907


# ---------------------------------------------------------------------------------------------------------------------------------
# create_comms when the communicator set-up FAILS on a real RCCL: two ranks on one device is exactly a configuration RCCL rejects
# ("duplicate GPU").  Every rank must come back with None -- agreed through the process group, nobody left waiting inside the
# collective initialisation -- and stay usable for the torch.distributed fallback.
def _comms_fail_worker(rank, world, port, ret):
    os.environ["ROME_RCCL_INIT_TIMEOUT_S"] = "60"
    torch, dist, R = _init(rank, world, port)
    try:
        import time
        from rome_jl_amd import rccl
        dev = torch.device("cuda", 0)
        t0 = time.perf_counter()
        comms = rccl.create_comms(torch, dist, world, rank, dev, 2)
        dt = time.perf_counter() - t0
        # the process group still works afterwards (the fallback path needs it)
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        ret[rank] = (comms is None, dt, float(t.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_create_comms_failure_is_agreed_across_ranks_real_rccl():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_comms_fail_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        none, dt, s = ret[r]
        assert none, "rank %d got communicators for two ranks on one device" % r
        assert dt < 100.0
        assert s == 3.0


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] shape: the parametric solve with its linearisation rows sharded over the ranks (LinearizeShard), the per-rank
# evaluator being the real `rome_linearize` kernels: every rank takes the same Levenberg-Marquardt steps as the unsharded solve.
def _lin_worker(rank, world, port, ret):
    torch, dist, R = _init(rank, world, port)
    try:
        from rome_jl_amd.distributed import LinearizeShard
        fg = R.synth_helix3d(P=120, N=8, seed=4)            # SE(3) helix with loop closures between adjacent turns
        shard = LinearizeShard(torch, dist, world, rank, device="cpu")          # default kernel: api.linearize (the HIP entry point)
        x = R.solveGraphParametric(fg, shard=shard)
        ret[rank] = {l: np.asarray(v) for l, v in x.items()}
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_linearisation_processes_real_kernels(world):
    import torch.multiprocessing as mp
    import rome_jl_amd as R
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_lin_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    fg = R.synth_helix3d(P=120, N=8, seed=4)            # SE(3) helix with loop closures between adjacent turns
    ref = R.solveGraphParametric(fg)                                             # unsharded, same kernels
    for r in range(world):
        assert set(ret[r]) == set(ref)
        for l in ref:
            assert np.array_equal(ret[r][l], np.asarray(ref[l])), (r, l)        # every rank, bit for bit


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] sharded: the honeycomb lattice with multihypo re-sightings, one segment per rank, cut through the lattice -- SIX
# pose separators and FIVE landmark separators per rank, both bearing-range directions with hypothesis columns (the fused multihypo
# sweep kernel where the table shape allows), ONE all-gather per step carrying Pose2 and Point2 blocks.
_BH_POSES = (2, 5, 8, 11, 14, 17)
_BH_LMS = 5


def _beehive_segment(R, rank, N):
    rng = np.random.default_rng(90 + rank)
    fg = R.synth_beehive_mh(20, N=N)
    lms = [l for l, t in fg.variables.items() if t is R.Point2]
    leg = R.Pose2Pose2(R.MvNormal([10.0, 0.0, np.pi / 3], np.diag(np.square([0.1, 0.1, 0.1]))))
    sight = lambda: R.Pose2Point2BearingRange(R.Normal(0.0, 0.03), R.Normal(20.0, 0.5))   # noqa: E731
    for k, p in enumerate(_BH_POSES):
        fg.addVariable("gp%d" % k, R.Pose2)
        fg.addFactor(["gp%d" % k, "x%d" % p], leg)
    for k in range(_BH_LMS):
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


def _beehive_layout(R, pk, dg):
    rows_p = []
    for p in _BH_POSES:
        f = int(np.nonzero((pk.p2p2["var_from"] == pk.index["x%d" % (p - 1)]) & (pk.p2p2["var_to"] == pk.index["x%d" % p]))[0][0])
        rows_p.append(2 * f)
    r0 = dg.family_table("br0")["rows4"].cpu().numpy()
    lm_idx = [pk.index[l] for l in pk.labels[R.Point2][:_BH_LMS]]
    rows_l = [int(np.nonzero(r0[:, 3] == li)[0][0]) for li in lm_idx]
    return rows_p, rows_l


def _beehive_worker(rank, world, port, ret, N, steps):
    torch, dist, R = _init(rank, world, port)
    try:
        from rome_jl_amd.distributed import SeparatorPipeline
        dev = torch.device("cuda", 0)
        fg = _beehive_segment(R, rank, N)
        ctx = R.Context(0)
        dg = R.DeviceGraph(fg, device=dev, ctx=ctx); dg.upload_beliefs(fg)
        pk = dg.packed
        rows_p, rows_l = _beehive_layout(R, pk, dg)
        comms = [HostStagedComm(torch, dist, world, ctx) for _ in range(2)]
        pipe = SeparatorPipeline(dg, R.make_opts(N=N, seed=3, stream_offset=rank << 32), dist, world, rank,
                                 publish=[("p2p2", r) for r in rows_p] + [("br0", r) for r in rows_l],
                                 ghosts=[(R.Pose2, pk.index["gp%d" % k], rank - 1, k) for k in range(len(_BH_POSES))] +
                                        [(R.Point2, pk.index["gl%d" % k], rank - 1, k) for k in range(_BH_LMS)], rccl_comms=comms)
        hist = []
        for k in range(steps):
            pipe.step()
            torch.cuda.synchronize()
            hist.append({f: pipe.out[k % 2][f].cpu().numpy().copy() for f in pipe.families})
        pipe.drain()
        ret[rank] = hist
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_beehive_multihypo_lattice_cut_two_processes_real_kernels():
    import torch
    import torch.multiprocessing as mp
    import rome_jl_amd as R
    world, N, steps = 2, 100, 4
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_beehive_worker, args=(world, _free_port(), ret, N, steps), nprocs=world, join=True)
    dev = torch.device("cuda", 0)
    dgs = []
    for r in range(world):
        fg = _beehive_segment(R, r, N)
        dg = R.DeviceGraph(fg, device=dev); dg.upload_beliefs(fg)
        dgs.append(dg)
    lay = [_beehive_layout(R, d.packed, d) for d in dgs]
    assert any(d.family_table("br1")["alt"] is not None for d in dgs)
    hist = [[] for _ in range(world)]
    for k in range(steps):
        for r, d in enumerate(dgs):
            pk = d.packed
            bel = {vt: d.bel[vt].clone() for vt in (R.Pose2, R.Point2)}
            if k >= 2:   # step k reads what the previous rank published in step k - 2
                src = (r - 1) % world
                for s, row in enumerate(lay[src][0]):
                    bel[R.Pose2][pk.index["gp%d" % s]] = torch.as_tensor(hist[src][k - 2]["p2p2"][row], device=dev)
                for s, row in enumerate(lay[src][1]):
                    bel[R.Point2][pk.index["gl%d" % s]] = torch.as_tensor(hist[src][k - 2]["br0"][row], device=dev)
            outs = {}
            for f in d.families():
                tb = d.family_table(f)
                out = torch.zeros((tb["n"], tb["vt_target"].dim, N), dtype=torch.float64, device=dev)
                mh = {} if tb["alt"] is None else dict(alt_var=tb["alt"], hypo_w=tb["w"])
                d._plan(tb["fn"], R.make_opts(N=N, seed=3, stream_offset=r << 32), n_conv=tb["n"], dir_all=tb["dir_all"], rows4=tb["rows4"],
                        mu=tb["mu"], L=tb["L"], bel_fixed=bel[tb["vt_fixed"]], bel_target=bel[tb["vt_target"]], out=out, **mh)()
                torch.cuda.synchronize()
                outs[f] = out.cpu().numpy()
            hist[r].append(outs)
    for r in range(world):
        for k in range(steps):
            for f in ("p2p2", "br1", "br0"):
                assert np.array_equal(ret[r][k][f], hist[r][k][f]), (r, k, f)
    assert not np.array_equal(ret[0][2]["p2p2"], ret[0][0]["p2p2"]) and not np.array_equal(ret[1][2]["br1"], ret[1][0]["br1"])
