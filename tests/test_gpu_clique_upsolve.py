"""rome_clique_upsolve (IIF upGibbsCliqueDensity, device-resident) against the oracle's restatement of the same loop, clique by
clique on the hexagon (BASELINE configs[0]; windows of test/testHexagonal2D_CliqByCliq.jl:37-79)."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

import rome_jl_amd as R   # noqa: E402
import oracle as ro       # noqa: E402
from solve_ref import upsolve_ref   # noqa: E402

CLIQUES = [["x0", "x1"], ["x2"], ["x3"], ["x4"], ["x5"], ["x6", "l1"]]


def _hex(N=100, seed=5):
    fg = R.generateGraph_Hexagonal(N=N)
    R.dead_reckon_init(fg, seed=seed)
    fg.initVariable("l1", np.array([[20.0], [0.0]]) + np.random.default_rng(1).standard_normal((2, N)))
    return fg


def _wrapdiff(a, b, dim):
    d = a - b
    if dim == 3:
        d[2] = np.arctan2(np.sin(d[2]), np.cos(d[2]))
    return d


@pytest.mark.parametrize("schedule", ["sequential", "jacobi"])
def test_hexagon_clique_by_clique_equals_the_oracle_loop(schedule):
    N = 100
    fg_d, fg_o = _hex(N), _hex(N)
    for p in range(2):   # two upward passes over the cliques
        for ci, fr in enumerate(CLIQUES):
            seed = 1000 + 17 * p + ci
            res = R.upGibbsCliqueDensity(fg_d, fr, gibbsIters=3, schedule=schedule, seed=seed)
            ref = upsolve_ref(R, fg_o, fr, N, seed=seed, gibbs_iters=3, schedule=schedule)
            for l in fr:
                pts, bw = res[l]
                dim = pts.shape[0]
                d = _wrapdiff(pts.copy(), ref[l], dim)
                assert np.mean(np.abs(d) < 1e-6) > 0.9, (p, fr, l, np.mean(np.abs(d) < 1e-6))
                md = np.array([np.mean(d[k]) for k in range(dim)])
                assert np.abs(md).max() < 1e-3, (p, fr, l, md)          # north_star tolerance on the belief means
                bo = ro.kde_bandwidths(ref[l][None], 0b100 if dim == 3 else 0)[0]
                assert np.allclose(bw, bo, rtol=2e-2), (l, bw, bo)      # the manikde! bandwidth setValKDE! stores
                fg_o.initVariable(l, ref[l])
    # statistical windows of the reference's clique-by-clique hexagon test (test/testHexagonal2D_CliqByCliq.jl:37-79)
    want = {"x0": (0, 0), "x1": (10, 0), "x2": (15, 8.66), "x3": (10, 17.32), "x4": (0, 17.32), "x5": (-5, 8.66), "x6": (0, 0), "l1": (20, 0)}
    for l, (x, y) in want.items():
        pts = fg_d.getVal(l)
        inside = np.mean((np.abs(pts[0] - x) < 3.0) & (np.abs(pts[1] - y) < 3.0))
        assert inside > 0.55, (l, inside)


def test_upsolve_with_more_than_128_particles_equals_the_oracle_loop():
    """N = 150 (test/testPoint2Point2Init.jl:12 runs that many) and N = 256: the 256-slot instantiation of the tree build / sampler and the
    general-N bandwidth kernel inside rome_clique_upsolve, against the oracle's loop; N = 257 is refused before anything is launched."""
    for N, fr in ((150, ["x0", "x1"]), (256, ["x6", "l1"])):
        fg_d, fg_o = _hex(N), _hex(N)
        res = R.upGibbsCliqueDensity(fg_d, fr, gibbsIters=2, seed=77)
        ref = upsolve_ref(R, fg_o, fr, N, seed=77, gibbs_iters=2)
        for l in fr:
            pts, bw = res[l]
            d = _wrapdiff(pts.copy(), ref[l], pts.shape[0])
            assert np.mean(np.abs(d) < 1e-6) > 0.9 and np.abs(d.mean(1)).max() < 1e-3, (N, l, np.mean(np.abs(d) < 1e-6))
            assert (bw > 0).all()
    with pytest.raises(R.RomeError) as e:
        R.upGibbsCliqueDensity(_hex(257), ["x1"], seed=1)
    assert e.value.code == R._lib.ERR_UNSUPPORTED_N


def test_upsolve_messages_and_layouts_and_errors():
    N = 100
    fg = _hex(N)
    # an upward message on x1 (a tight density at a shifted position) pulls the product of x1 towards it
    msg = fg.getVal("x1").copy(); msg[0] += 1.0; msg[:2] = msg[:2].mean(1, keepdims=True) + 0.05 * (msg[:2] - msg[:2].mean(1, keepdims=True))
    a = R.upGibbsCliqueDensity(fg, ["x1"], gibbsIters=2, seed=9, setvals=False)["x1"][0]
    b = R.upGibbsCliqueDensity(fg, ["x1"], gibbsIters=2, seed=9, setvals=False, messages={"x1": [msg]})["x1"][0]
    ref = upsolve_ref(R, fg, ["x1"], N, seed=9, gibbs_iters=2, messages={"x1": [msg]})["x1"]
    assert np.mean(np.abs(_wrapdiff(b.copy(), ref, 3)) < 1e-6) > 0.9
    assert abs(b[0].mean() - msg[0].mean()) < abs(a[0].mean() - msg[0].mean())
    # rows not grouped by target in update order / a row targeting a variable that is not updated -> ROME_ERR_INVALID_ARG
    from rome_jl_amd.clique import CliqueBatch
    pairs = [(fl, d) for d in ("x1", "x2") for fl, labels, _ in fg.factors if d in labels]
    batch = CliqueBatch(fg, pairs)
    o = R.make_opts(N=N, seed=3)
    with pytest.raises(R.RomeError):
        batch.upsolve(o, ["x2", "x1"])        # rows are grouped x1-first
    with pytest.raises(R.RomeError):
        batch.upsolve(o, ["x1"])              # rows targeting x2 have no updated variable
    ok = batch.upsolve(o, ["x1", "x2"])
    assert set(ok) == {"x1", "x2"} and np.isfinite(ok["x1"][0]).all() and (ok["x1"][1] > 0).all()
    with pytest.raises(R.RomeError):
        batch.upsolve(R.make_opts(N=N, seed=3), ["x1", "x1"])


def test_upsolve_pcie_inclusive_time_per_clique():
    """ms per clique, host beliefs in -> new host beliefs out (PCIe and Python included); written to gpurun_out/ for profiles/."""
    N = 100
    fg = _hex(N)
    R.upGibbsCliqueDensity(fg, ["x0", "x1"], seed=1, setvals=False)
    ts = []
    for rep in range(20):
        t0 = time.perf_counter()
        for fr in CLIQUES:
            R.upGibbsCliqueDensity(fg, fr, gibbsIters=3, seed=rep, setvals=False)
        ts.append((time.perf_counter() - t0) / len(CLIQUES))
    ms = 1e3 * float(np.median(ts))
    tj = []
    for rep in range(20):
        t0 = time.perf_counter()
        for fr in CLIQUES:
            R.upGibbsCliqueDensity(fg, fr, gibbsIters=3, seed=rep, setvals=False, schedule="jacobi")
        tj.append((time.perf_counter() - t0) / len(CLIQUES))
    msj = 1e3 * float(np.median(tj))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r03_clique_upsolve.txt"), "w") as f:
        f.write("rome_clique_upsolve, hexagon cliques %s, N=100, gibbsIters=3, host beliefs in -> host beliefs out (PCIe + Python included):\n"
                "  sequential schedule %.3f ms per clique (median of 20 passes)\n  jacobi schedule     %.3f ms per clique\n" % (CLIQUES, ms, msj))
    assert ms < 50.0


def test_frontier_of_independent_cliques_in_one_call_equals_the_oracle():
    """cliques on the current frontier are independent -> ONE rome_clique_upsolve call with update groups (group g = the
    g-th frontal of every clique), against the oracle's restatement with the same groups."""
    N = 100
    fg_d, fg_o = _hex(N), _hex(N)
    frontier = [["x0", "x1"], ["x3"], ["x5"]]       # x2, x4, x6, l1 only appear as fixed separators
    res = R.upGibbsCliqueFrontier(fg_d, frontier, gibbsIters=3, seed=41)
    order = ["x0", "x3", "x5", "x1"]; groups = [0, 0, 0, 1]
    ref = upsolve_ref(R, fg_o, order, N, seed=41, gibbs_iters=3, groups=groups)
    assert set(res) == set(order)
    for l in order:
        d = _wrapdiff(res[l][0].copy(), ref[l], 3)
        assert np.mean(np.abs(d) < 1e-6) > 0.9 and np.abs(d.mean(1)).max() < 1e-3, (l, np.mean(np.abs(d) < 1e-6))
    with pytest.raises(ValueError):
        R.upGibbsCliqueFrontier(_hex(N), [["x0"], ["x1"]])          # x0 - x1 share a factor: not independent
    with pytest.raises(ValueError):
        R.upGibbsCliqueFrontier(_hex(N), [["x0"], ["x0", "x3"]])


def test_manhattan_frontier_throughput():
    """A frontier of ~1000 single-frontal cliques of the M3500 graph (an independent set of poses) through ONE call, PCIe and Python
    included: ms per clique -> gpurun_out/ for profiles/."""
    import time
    N = 100
    fg = R.loadG2o(os.path.join(ROOT, "tests", "golden", "manhattan.g2o"), N=N)
    R.dead_reckon_init(fg, seed=11)
    nbr = {l: set() for l in fg.variables}
    for _, labels, _ in fg.factors:
        for a in labels:
            nbr[a].update(b for b in labels if b != a)
    chosen, blocked = [], set()
    for l in fg.variables:                         # greedy independent set
        if l not in blocked:
            chosen.append(l); blocked.add(l); blocked.update(nbr[l])
    assert len(chosen) > 900
    frontier = [[l] for l in chosen]
    R.upGibbsCliqueFrontier(fg, frontier[:10], gibbsIters=3, seed=1, setvals=False)
    t0 = time.perf_counter()
    res = R.upGibbsCliqueFrontier(fg, frontier, gibbsIters=3, seed=2, setvals=False)
    dt = time.perf_counter() - t0
    assert len(res) == len(chosen) and all(np.isfinite(p).all() and (bw > 0).all() for p, bw in res.values())
    moved = np.mean([np.abs(res[l][0][:2].mean(1) - fg.getVal(l)[:2].mean(1)).max() for l in chosen[:200]])
    assert moved > 1e-4
    with open(os.path.join(ROOT, "gpurun_out", "r03_clique_frontier.txt"), "w") as f:
        f.write("rome_clique_upsolve with update groups: a frontier of %d independent single-frontal cliques of the M3500 graph (N=100, gibbsIters=3) in ONE call,\n"
                "host beliefs in -> host beliefs out (PCIe + Python table building included): %.1f ms total = %.4f ms per clique\n" % (len(chosen), 1e3 * dt, 1e3 * dt / len(chosen)))


def test_frontier_shard_one_rank_rccl_device_resident():
    """rome_jl_amd.distributed.FrontierShard, device-resident: the share's up-solve plan writes the new frontals in place and into the
    exchange buffer, ONE ncclAllGather (direct binding, one rank, collective forced), ONE scatter launch -- same beliefs as the single
    unsharded plan and as the one-shot host call; the multi-rank form (world 2 / 8) runs over gloo in tests/test_distributed_gloo.py.
    Then the same on a ~1500-clique frontier of Manhattan-3500: ms per frontier step with NO belief crossing PCIe -> gpurun_out/."""
    import time
    import torch
    import torch.distributed as dist
    from rome_jl_amd.distributed import FrontierShard
    from rome_jl_amd.clique import DeviceStore, UpsolvePlan
    from rome_jl_amd import rccl
    N = 100
    frontiers = [[["x0", "x1"], ["x3"], ["x5"]], [["x2"], ["x4"], ["x6", "l1"]]]
    fg_a, fg_b, fg_c = _hex(N), _hex(N), _hex(N)
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = str(__import__("portutil").free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        comms = rccl.create_comms(torch, dist, 1, 0, dev, 1)
        s_ref = DeviceStore(fg_a)
        ref_plans = [UpsolvePlan(s_ref, f, gibbsIters=2) for f in frontiers]
        s_sh = DeviceStore(fg_b)
        sh = FrontierShard(s_sh, torch, dist, 1, 0, device=dev, comm=comms[0] if comms else None, always_collective=True)
        plans = [sh.plan(f, gibbsIters=2) for f in frontiers]
        for p in range(2):
            for k in range(2):
                o = R.make_opts(N=N, seed=5 + p, stream_offset=(2 * p + k) << 40)
                ref_plans[k].run(o)
                sh.step(plans[k], o)
        torch.cuda.synchronize()
        for f in frontiers:
            for c in f:
                for l in c:
                    assert np.array_equal(s_sh.get(l), s_ref.get(l)), l
        # the first frontier of the first pass == the one-shot host call (default stream ids = positions in the frontier's tables)
        s_one = DeviceStore(fg_c)
        UpsolvePlan(s_one, frontiers[0], gibbsIters=2).run(R.make_opts(N=N, seed=5))
        direct = R.upGibbsCliqueFrontier(fg_c, frontiers[0], gibbsIters=2, seed=5, setvals=False)
        for l in direct:
            assert np.array_equal(s_one.get(l), direct[l][0]), l
        # ---- Manhattan-3500: a frontier of ~1500 independent single-frontal cliques, device-resident step time
        fg = R.loadG2o(os.path.join(ROOT, "tests", "golden", "manhattan.g2o"), N=N)
        R.dead_reckon_init(fg, seed=11)
        nbr = {l: set() for l in fg.variables}
        for _, labels, _ in fg.factors:
            for a in labels:
                nbr[a].update(b for b in labels if b != a)
        chosen, blocked = [], set()
        for l in fg.variables:
            if l not in blocked:
                chosen.append(l); blocked.add(l); blocked.update(nbr[l])
        store = DeviceStore(fg)
        shm = FrontierShard(store, torch, dist, 1, 0, device=dev, comm=comms[0] if comms else None, always_collective=True)
        t0 = time.perf_counter()
        pl = shm.plan([[l] for l in chosen], gibbsIters=3)
        t_plan = time.perf_counter() - t0
        for w in range(2):
            shm.step(pl, R.make_opts(N=N, seed=w))
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            shm.step(pl, R.make_opts(N=N, seed=10 + rep))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ms = 1e3 * float(np.median(ts))
        moved = np.mean([np.abs(store.get(l)[:2].mean(1) - fg.getVal(l)[:2].mean(1)).max() for l in chosen[:50]])
        assert moved > 1e-4 and np.isfinite(store.get(chosen[-1])).all()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r04_frontier_shard.txt"), "w") as f:
            f.write("FrontierShard, device-resident (rome_store + rome_upsolve_plan + rome_scatter_plan), one rank, collective forced (%s):\n"
                    "a frontier of %d independent single-frontal cliques of Manhattan-3500, N=100, gibbsIters=3:\n"
                    "  plan built once (host tables, Python): %.1f ms\n"
                    "  step = up-solve plan run + ONE all-gather of %d blocks x %d B + ONE scatter launch: %.3f ms (median of 5), %.2f us per clique,\n"
                    "  PCIe crossings per step: 0 (round 3: host beliefs in -> out, 20.1 ms per frontier)\n"
                    % ("direct ncclAllGather" if comms else "torch.distributed", len(chosen), 1e3 * t_plan, len(chosen), 6 * N * 8, ms, 1e3 * ms / len(chosen)))
        if comms:
            for cm in comms:
                cm.close()
    finally:
        dist.destroy_process_group()


def test_pose3_clique_upsolve_equals_the_oracle_loop():
    """rome_clique_upsolve on Pose3 variables (Pose3Pose3 + PriorPose3 rows, dim-6 product in the chart of each proposal, rotation-aware
    bandwidth rule): a small helix, two cliques, against the oracle's restatement."""
    from scipy.spatial.transform import Rotation as Rot
    N = 64
    mk = lambda: (lambda fg: (R.dead_reckon_init_pose3(fg, seed=2, sigma=(0.2, 0.2, 0.2, 0.02, 0.02, 0.02)), fg)[1])(R.synth_helix3d(P=8, N=N, seed=3))
    fg_d, fg_o = mk(), mk()
    for ci, fr in enumerate([["x0", "x1"], ["x4"]]):
        res = R.upGibbsCliqueDensity(fg_d, fr, gibbsIters=2, seed=70 + ci)
        ref = upsolve_ref(R, fg_o, fr, N, seed=70 + ci, gibbs_iters=2)
        for l in fr:
            pts = res[l][0]
            dt = np.abs(pts[:3] - ref[l][:3])
            ang = np.array([np.linalg.norm((Rot.from_rotvec(a).inv() * Rot.from_rotvec(b)).as_rotvec()) for a, b in zip(pts[3:].T, ref[l][3:].T)])
            assert np.mean(dt < 1e-6) > 0.9 and np.mean(ang < 1e-6) > 0.9, (l, np.mean(dt < 1e-6), np.mean(ang < 1e-6))
            assert np.abs(pts[:3].mean(1) - ref[l][:3].mean(1)).max() < 1e-3
            assert (res[l][1] > 0).all()
            fg_o.initVariable(l, ref[l])


def test_landmark_prior_rows_in_the_clique_entries():
    """PriorPoint2 rows in rome_clique_proposals / rome_clique_upsolve: the prior's samples are a proposal of its landmark (family
    offset 7 << 28), behind the bearing-range -> landmark rows in the product; clique call == per-factor call, up-solve == oracle."""
    from rome_jl_amd.clique import FAMILY_STREAM
    N = 100
    def graph():
        fg = _hex(N)
        fg.addFactor(["l1"], R.PriorPoint2(R.MvNormal([20.0, 0.0], np.diag([0.3, 0.3]) ** 2)))
        return fg
    fg_d, fg_o = graph(), graph()
    props, batch = R.proposalbeliefs(fg_d, "l1", seed=9, stream_offset=100)
    assert len(props) == 3 and ("l1f1", "l1") in props
    fam, r = batch.rows[("l1f1", "l1")]
    assert fam == "prpt2"
    assert np.array_equal(props[("l1f1", "l1")], R.approxConv(fg_d, "l1f1", "l1", seed=9, stream_offset=100 + FAMILY_STREAM["prpt2"] + r))
    res = R.upGibbsCliqueDensity(fg_d, ["x6", "l1"], gibbsIters=3, seed=61)
    ref = upsolve_ref(R, fg_o, ["x6", "l1"], N, seed=61, gibbs_iters=3)
    for l, dim in (("x6", 3), ("l1", 2)):
        d = _wrapdiff(res[l][0].copy(), ref[l], dim)
        assert np.mean(np.abs(d) < 1e-6) > 0.9 and np.abs(d.mean(1)).max() < 1e-3, (l, np.mean(np.abs(d) < 1e-6))
    assert np.abs(res["l1"][0].mean(1) - [20.0, 0.0]).max() < 0.5
