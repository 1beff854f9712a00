"""The Bayes-tree solve (rome_jl_amd.tree): ordering -> cliques -> levels -> up messages -> down pass, device-resident (one
rome_upsolve_plan per tree level, store-resident messages, sampled-measurement rows, block operations), against the oracle's
restatement of the same schedule (tests/dist_standin.py: OracleTreeBackend drives tests/solve_ref.py::upsolve_ref with the rows, groups,
stream ids and messages of every level)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

import rome_jl_amd as R   # noqa: E402
from rome_jl_amd.tree import BayesTree, TreeSolver, BlockOpPlan   # noqa: E402
from rome_jl_amd.clique import DeviceStore   # noqa: E402
from dist_standin import OracleTreeBackend, OracleTreeBlockOp, OracleTreeStore   # noqa: E402

G2O = os.path.join(ROOT, "tests", "golden", "manhattan.g2o")
HEX = {"x0": (0, 0), "x1": (10, 0), "x2": (15, 8.66), "x3": (10, 17.32), "x4": (0, 17.32), "x5": (-5, 8.66), "x6": (0, 0), "l1": (20, 0)}


def _wd(a, b):
    d = a - b
    if d.shape[0] == 3:
        d[2] = np.arctan2(np.sin(d[2]), np.cos(d[2]))
    return d


def test_block_operations_match_their_numpy_restatement():
    N = 100
    fg = R.initfg(N)
    rng = np.random.default_rng(3)
    for k in range(4):
        fg.addVariable("x%d" % k, R.Pose2)
    for k in range(3):
        fg.addVariable("l%d" % k, R.Point2)
    fg.addVariable("q0", R.Pose3); fg.addVariable("q1", R.Pose3)
    fg.initVariable("x0", np.array([[3.0], [-2.0], [3.1]]) + np.array([[0.5], [0.5], [0.2]]) * rng.standard_normal((3, N)))   # heading across the cut
    fg.initVariable("x1", np.array([[8.0], [1.0], [-3.0]]) + 0.3 * rng.standard_normal((3, N)))
    fg.initVariable("l0", np.array([[12.0], [5.0]]) + 0.4 * rng.standard_normal((2, N)))
    fg.initVariable("q0", rng.standard_normal((6, N)))
    dev, orc = DeviceStore(fg), OracleTreeStore(R, fg)
    orc.upload(fg)
    steps = [("anchor", [("x0", "x2"), ("l0", "l1"), ("q0", "q1")]), ("relative", [("x2", "x1", "x3"), ("x2", "l0", "l2")]), ("copy", [("x3", "x1"), ("l2", "l1")])]
    for op, ent in steps:
        BlockOpPlan(dev, op, ent).run(); OracleTreeBlockOp(orc, op, ent).run()
    for l in fg.variables:
        if l in orc.vals:
            assert np.abs(_wd(dev.get(l), orc.vals[l])).max() < 1e-12, l
    z = dev.get("x1")          # = the relative samples: composing them onto the anchor gives the particles of the source back
    ref = orc.vals["x2"][:, 0]
    c, s = np.cos(ref[2]), np.sin(ref[2])
    back = np.stack([ref[0] + c * z[0] - s * z[1], ref[1] + s * z[0] + c * z[1], ref[2] + z[2]])
    assert np.abs(_wd(back, fg.getVal("x1"))).max() < 1e-12


def _both(fg, messages, seed, passes=1, **kw):
    N = fg.N
    dev = TreeSolver(fg, messages=messages, **kw)
    orc = TreeSolver(fg, tree=dev.tree, messages=messages, backend=OracleTreeBackend(R), **kw)
    dev.upload(); orc.upload()
    out = []
    for ps in range(passes):
        o = R.make_opts(N=N, seed=seed + ps)
        dev.solve(o); orc.solve(o)
        fr, dm = [], []
        for l in fg.variables:
            d = _wd(dev.store.get(l), orc.store.get(l))
            fr.append(np.mean(np.abs(d) < 1e-6)); dm.append(np.abs(d.mean(axis=1)).max())
        out.append((float(np.mean(fr)), float(np.max(dm))))
    return dev, out


@pytest.mark.parametrize("messages", ["marginal", "relative"])
def test_hexagon_tree_solve_equals_the_oracle_tree_solve(messages):
    fg = R.generateGraph_Hexagonal(N=100)
    R.initAllOrdered(fg, seed=4)                                    # IIF: initAll! before solveTree!
    dev, worst = _both(fg, messages, 31, passes=2)
    assert dev.tree.cliques[0].parent == -1 and len(dev.tree.levels) >= 3
    for frac, dmean in worst:
        assert frac > 0.9 and dmean < 1e-3, worst                   # north_star tolerance on the belief means
    # the windows of test/testHexagonal2D_CliqByCliq.jl:37-79 on the device's posterior
    dev.download()
    for l, (x, y) in HEX.items():
        p = fg.getVal(l)
        if messages == "marginal":     # IIF's message form: the reference's own acceptance test
            assert np.mean((np.abs(p[0] - x) < 3.0) & (np.abs(p[1] - y) < 3.0)) > 0.35, (l, p[:2].mean(1))
        else:                          # beliefs conditional on one anchor chain are wider: the belief MEANS sit in the windows
            assert np.abs(p[:2].mean(1) - [x, y]).max() < 3.0, (l, p[:2].mean(1))


def manhattan_subgraph(P, N, tmpdir):
    """the Manhattan factors among the first P poses (odometry AND the loop closures between them)"""
    path = os.path.join(str(tmpdir), "manhattan_%d.g2o" % P)
    with open(path, "w") as f:
        for ln in open(G2O):
            t = ln.split()
            if t and t[0] == "EDGE_SE2" and int(t[1]) < P and int(t[2]) < P:
                f.write(ln)
    return R.loadG2o(path, N=N)


@pytest.mark.parametrize("messages", ["marginal", "relative"])
def test_manhattan_subgraph_tree_solve_equals_the_oracle_tree_solve(messages, tmp_path):
    """the first 150 poses of manhattan.g2o with their loop closures: multi-frontal cliques, several separators per clique, absolute
    and relative messages, sampled-measurement rows in the parents"""
    fg = manhattan_subgraph(150, 64, tmp_path)
    assert len(fg.factors) > 170
    R.initAllOrdered(fg, seed=2)
    dev, worst = _both(fg, messages, 41)
    st = dev.stats()
    assert st["levels"] > 5 and st["store_messages"] > 0 and (messages == "marginal" or st["relative_messages"] > 20), st
    for frac, dmean in worst:
        assert frac > 0.9 and dmean < 1e-3, worst


def test_manhattan_subgraph_message_trees_equal_the_oracle_tree_solve(tmp_path):
    """message_tree="hop": a spanning tree of relative messages over the separators (several anchored solves per clique)"""
    fg = manhattan_subgraph(150, 64, tmp_path)
    R.initAllOrdered(fg, seed=2)
    dev, worst = _both(fg, "relative", 41, message_tree="hop")
    assert dev.stats()["relative_messages"] > 20
    for frac, dmean in worst:
        assert frac > 0.9 and dmean < 1e-3, worst


@pytest.mark.timeout(1500)
def test_manhattan_1000_pose_prefix_tree_pass_equals_the_oracle_tree_pass(tmp_path):
    """Row h's numerics on the headline graph: ONE up + down pass over the first 1000 poses of manhattan.g2o with their loop closures
    (909 cliques, 45 levels, ~7500 rows, ~1700 relative messages over the message trees, two-stage products) at N = 100 -- device ==
    the oracle's restatement of the same schedule: > 90 % of the particles to 1e-6, every pose mean < 1e-3 (north_star's tolerance)."""
    import time
    fg = manhattan_subgraph(1000, 100, tmp_path)
    assert len(fg.variables) == 1000 and len(fg.factors) > 1300
    R.initAllOrdered(fg, seed=2)
    t0 = time.perf_counter()
    dev, worst = _both(fg, "relative", 61, message_tree="hop")
    st = dev.stats()
    assert st["levels"] > 30 and st["relative_messages"] > 1000 and st["cliques"] > 700, st
    (frac, dmean), = worst
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_tree_parity_1000.txt"), "w") as f:
        f.write("Manhattan first 1000 poses, N=100, one up+down tree pass (message_tree=hop), device vs oracle restatement: %.4f of the particles "
                "within 1e-6, worst |mean difference| %.3e; %s; %.1f s incl. the oracle pass\n" % (frac, dmean, st, time.perf_counter() - t0))
    assert frac > 0.9 and dmean < 1e-3, worst


def test_beehive_multihypo_tree_solve_runs_with_landmark_separators():
    """BASELINE configs[3]: landmarks among the separators (relative messages pose -> landmark as sampled bearing-range rows), multihypo
    sightings inside the cliques"""
    fg = R.synth_beehive_mh(20, N=100)
    R.initAllOrdered(fg, seed=2)
    ts = TreeSolver(fg, messages="relative")
    assert any(fg.variables[s] is R.Point2 for c in ts.tree.cliques for s in c.separators)
    ts.upload(); ts.solve(R.make_opts(N=100, seed=3)); ts.download()
    sim = fg._sim
    err = [np.hypot(*(fg.getVal(l)[:2].mean(1) - np.asarray(sim[l])[:2])) for l in fg.variables if fg.variables[l] is R.Pose2]
    assert np.isfinite(err).all() and np.median(err) < 3.0, np.median(err)


@pytest.mark.parametrize("messages", ["marginal", "relative"])
def test_pose3_helix_tree_solve_equals_the_oracle_tree_solve(messages):
    """SE(3): a 40-pose helix with loop closures between adjacent turns and its PriorPose3.  The relative form has no Pose2 anchor on a
    Pose3 graph: its cliques send the marginals of the separators their priors / children informed (the documented fall-back)."""
    fg = R.synth_helix3d(P=40, N=32, seed=4)
    R.dead_reckon_init_pose3(fg, seed=2)
    dev, worst = _both(fg, messages, 51, gibbsIters=2)
    st = dev.stats()
    assert st["levels"] >= 3 and st["relative_messages"] == 0 and st["store_messages"] > 0, st
    for frac, dmean in worst:
        assert frac > 0.9 and dmean < 1e-3, worst


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_tree_levels_through_frontier_shard_one_rank_rccl():
    """TreeSolver(shard=FrontierShard) with the DIRECT RCCL binding (one rank, collective forced): every level's written blocks go through
    the packed exchange buffer (mirror stride N), ONE in-place ncclAllGather on the context's stream, ONE scatter -- same beliefs as the
    unsharded device solve, bit for bit; then Manhattan-3500: seconds per pass with the exchange in the loop -> gpurun_out/."""
    import time
    import torch
    import torch.distributed as dist
    from rome_jl_amd.distributed import FrontierShard
    from rome_jl_amd import rccl
    N = 64
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        comms = rccl.create_comms(torch, dist, 1, 0, dev, 1)
        comm = comms[0] if comms else None
        mk = lambda store: FrontierShard(store, torch, dist, 1, 0, device=dev, comm=comm, always_collective=True)   # noqa: E731
        fg_a, fg_b = R.generateGraph_Hexagonal(N=N), R.generateGraph_Hexagonal(N=N)
        for fg in (fg_a, fg_b):
            R.dead_reckon_init(fg, seed=5)
            fg.initVariable("l1", np.array([[20.0], [0.0]]) + np.random.default_rng(1).standard_normal((2, N)))
        ref = TreeSolver(fg_a, messages="relative"); sh = TreeSolver(fg_b, messages="relative", shard=mk)
        ref.upload(); sh.upload()
        for ps in range(2):
            o = R.make_opts(N=N, seed=7 + ps)
            ref.solve(o); sh.solve(o)
        torch.cuda.synchronize()
        for l in fg_a.variables:
            assert np.array_equal(sh.store.get(l), ref.store.get(l)), l
        # ---- Manhattan-3500 with the exchange in the loop
        fg = R.loadG2o(G2O, N=100)
        R.initAllOrdered(fg, seed=1)
        ts = TreeSolver(fg, messages="relative", shard=mk)
        ts.upload()
        ts.solve(R.make_opts(N=100, seed=1)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ps in range(3):
            ts.solve(R.make_opts(N=100, seed=2 + ps))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        assert dt < 2.0
        nbytes = sum(p["send"].numel() * 8 for plans in (ts.up_plans, ts.down_plans) for p in plans if p is not None)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r05_tree_shard_one_rank.txt"), "w") as f:
            f.write("TreeSolver(shard=FrontierShard) on Manhattan-3500, one rank, %s, collective forced (always_collective): every level = share up-solve (all "
                    "cliques) + in-place all-gather of the level's written blocks (packed: Pose2 = 3 slots of N doubles) + scatter:\n"
                    "  %.3f s per pass (0.104 s without the exchange path), %.1f MB through the exchange buffers per pass\n"
                    % ("direct ncclAllGather" if comm is not None else "torch.distributed all_gather", dt, nbytes / 1e6))
    finally:
        dist.destroy_process_group()


def test_solve_tree_host_entry_like_the_reference_example():
    """R.solveTree(fg) = the reference's `solveTree!(fg)` call of examples/Hexagonal2D_SLAM.jl:24 on a graph WITHOUT beliefs (initAll! inside),
    and a second call with the returned tree re-solves from the current beliefs (examples/ManhattanDatasetIncremental.jl:107)"""
    fg = R.generateGraph_Hexagonal(N=100)
    assert not fg.vals
    ts = R.solveTree(fg, messages="marginal", seed=3)
    assert all(fg.isInitialized(l) for l in fg.variables) and hasattr(fg, "ppes")
    for l, (x, y) in HEX.items():
        p = fg.getVal(l)
        assert np.mean((np.abs(p[0] - x) < 3.0) & (np.abs(p[1] - y) < 3.0)) > 0.35, (l, p[:2].mean(1))
    before = {l: fg.getVal(l).copy() for l in fg.variables}
    ts2 = R.solveTree(fg, tree=ts, seed=4)
    assert ts2 is ts and any(not np.array_equal(before[l], fg.getVal(l)) for l in fg.variables)
    with pytest.raises(ValueError):
        R.solveTree(R.generateGraph_Hexagonal(N=100), tree=ts)


def test_staged_products_agree_with_the_single_product_in_law():
    """A hub pose seen from 14 spokes (each spoke has its own prior and one Pose2Pose2 to the hub): the hub's product takes 14 proposals.
    TreeSolver(max_product=8) takes it in two stages (partial products of <= 8, identity rows), max_product=0 in ONE multiscale Gibbs product;
    both recover the information-weighted mean of the proposals (Gaussian case: known in closed form) within the sampling error, with
    comparable spread."""
    N = 100
    rng = np.random.default_rng(4)
    truth = np.array([12.0, -5.0, 0.7])
    res = {}
    for mp_ in (0, 8):
        fg = R.initfg(N)
        fg.addVariable("h", R.Pose2)
        for k in range(14):
            a = 2 * np.pi * k / 14
            spoke = np.array([truth[0] + 6 * np.cos(a), truth[1] + 6 * np.sin(a), a])
            c, s = np.cos(spoke[2]), np.sin(spoke[2])
            z = np.array([c * (truth[0] - spoke[0]) + s * (truth[1] - spoke[1]), -s * (truth[0] - spoke[0]) + c * (truth[1] - spoke[1]), truth[2] - spoke[2]])
            fg.addVariable("s%d" % k, R.Pose2)
            fg.addFactor(["s%d" % k], R.PriorPose2(R.MvNormal(spoke, np.diag([0.01, 0.01, 0.0004]))))
            fg.addFactor(["s%d" % k, "h"], R.Pose2Pose2(R.MvNormal(z, np.diag([0.04, 0.04, 0.0025]))))
        R.initAllOrdered(fg, seed=1)
        ts = TreeSolver(fg, messages="relative", max_product=mp_)
        assert max(len(s_.pairs) for s_ in ts.up_specs + ts.down_specs) > 0
        ts.upload(); ts.solve(R.make_opts(N=N, seed=9)); ts.download()
        h = fg.getVal("h")
        res[mp_] = (h[:2].mean(1), h[:2].std(1), np.arctan2(np.sin(h[2]).mean(), np.cos(h[2]).mean()))
        staged = any("^" in l for s_ in ts.up_specs + ts.down_specs for l in s_.order)
        assert staged == (mp_ == 8)
    for mp_, (m, sd, th) in res.items():
        assert np.abs(m - truth[:2]).max() < 0.15 and abs(th - truth[2]) < 0.05, (mp_, m, th)      # 14 proposals of sigma ~0.25: posterior sigma ~0.07
        assert 0.02 < sd.max() < 0.3, (mp_, sd)
    assert np.abs(res[0][0] - res[8][0]).max() < 0.15
