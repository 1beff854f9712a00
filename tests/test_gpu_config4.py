"""BASELINE configs[3] on ONE graph: the honeycomb / double-hex lattice (src/canonical/GenerateHoneycomb.jl:59-100,
test/testBeehive2D_DoubleHexInit.jl:7-66) WITH `multihypo=[1, .5, .5]` sightings (test/testMultimodalRangeBearing.jl:53):
table sweep == per-factor path bit for bit, device solve == the oracle's loop, the reference's windows, and the same graph cut in two
through SeparatorPipeline with a one-rank RCCL exchange (the multi-rank form runs over gloo in tests/test_distributed_gloo.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

import rome_jl_amd as R   # noqa: E402
import oracle as ro       # noqa: E402
from solve_ref import solve_ref   # noqa: E402


def _wd(a, b):
    d = a - b
    d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    return d


def _graph(P=20, N=100, seed=4):
    fg = R.synth_beehive_mh(P, N=N)
    R.dead_reckon_init(fg, seed=seed)
    rng = np.random.default_rng(seed)
    sim = fg._sim
    for l, t in fg.variables.items():   # landmarks: around their simulated position (what the reference's graphinit leaves behind)
        if t is R.Point2:
            fg.initVariable(l, np.asarray(sim[l])[:, None] + 0.5 * rng.standard_normal((2, N)))
    return fg


def test_generator_is_the_honeycomb_with_ambiguous_resightings():
    fg = R.synth_beehive_mh(36)
    plain = R.generateGraph_Honeycomb(36)
    assert list(fg.variables) == list(plain.variables) and len(fg.factors) == len(plain.factors)
    assert len(fg.multihypo) >= 5 and all(w == (0.5, 0.5) for w in fg.multihypo.values())
    for fl, labels, f in fg.factors:
        if fl in fg.multihypo:
            assert isinstance(f, R.Pose2Point2BearingRange) and len(labels) == 3 and labels[1] != labels[2]
            assert (f.bearing.mu, f.bearing.sigma, f.range.mu, f.range.sigma) == (0.0, 0.03, 20.0, 0.5)
    legs = [f for _, _, f in fg.factors if isinstance(f, R.Pose2Pose2)]
    assert all(f.Z.mu[0] == 10.0 and abs(abs(f.Z.mu[2]) - np.pi / 3) < 1e-12 for f in legs)


@pytest.mark.parametrize("solver", [R.SOLVER_NEWTON, R.SOLVER_GAUSS_NEWTON])
def test_table_sweep_equals_per_factor_path_bit_for_bit(solver):
    N = 100
    fg = _graph(20, N)
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    pk = dg.packed
    o = R.make_opts(N=N, solver=solver, seed=31)
    dg.conv_step(o, 0)
    p2 = dg.prop[R.Pose2].cpu().numpy(); pl = dg.prop[R.Point2].cpu().numpy()
    F = pk.p2p2["F"]; C2 = dg.tab["p2p2"]["C"]
    # Pose2Pose2 rows 2f (-> second pose), 2f+1 (-> first pose)
    for f, fl in enumerate(pk.p2p2["labels"]):
        _, labels, _ = fg.getFactor(fl)
        for d, tgt in ((0, labels[1]), (1, labels[0])):
            ref = R.approxConv(fg, fl, tgt, solver=solver, seed=31, stream_offset=dg.STREAM_P2P2 + 2 * f + d)
            assert np.array_equal(p2[2 * f + d], ref), (fl, d)
    # bearing-range -> pose rows (one per factor; multihypo: the landmark is drawn per particle)
    for k, fl in enumerate(pk.br["labels"]):
        _, labels, _ = fg.getFactor(fl)
        ref = R.approxConv(fg, fl, labels[0], solver=solver, seed=31, stream_offset=dg.STREAM_BR1 + k)
        assert np.array_equal(p2[C2 + k], ref), fl
    # bearing-range -> landmark rows (one per (factor, candidate))
    r0 = pk.br["rows0"]
    n_mh = 0
    for r in range(len(r0["factor"])):
        fl = pk.br["labels"][int(r0["factor"][r])]
        tgt = pk.labels[R.Point2][int(r0["point"][r])]
        ref = R.approxConv(fg, fl, tgt, solver=solver, seed=31, stream_offset=dg.STREAM_BR0 + r)
        assert np.array_equal(pl[r], ref), (fl, tgt)
        n_mh += int(r0["alt"][r] >= 0)
    assert n_mh >= 4


def test_device_solve_equals_the_oracle_loop_and_reference_windows():
    """double hexagon (x0..x13, l-lattice) with ambiguous re-sightings: DeviceGraph.solve against solve_ref under the shared RNG, then
    the boxes of test/testBeehive2D_DoubleHexInit.jl (> 50 of 100 particles; the reference asserts the same boxes on its solve)"""
    N, S = 100, 4
    fg = _graph(13, N)
    assert len(fg.multihypo) >= 1
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    dg.solve(R.make_opts(N=N, solver=1, seed=77), n_sweeps=S, bandwidth="lcv", product="gibbs")
    b2, bl = solve_ref(R, fg, S, N, seed=77, bandwidth="lcv", product="gibbs")
    g2 = dg.bel[R.Pose2].cpu().numpy(); gl = dg.bel[R.Point2].cpu().numpy()
    assert np.mean(np.abs(_wd(g2, b2)) < 1e-6) > 0.9 and np.mean(np.abs(gl - bl) < 1e-6) > 0.9
    m2, _ = R.belief_stats(g2); mo, _ = R.belief_stats(b2)
    dm = m2 - mo; dm[:, 2] = np.arctan2(np.sin(dm[:, 2]), np.cos(dm[:, 2]))
    assert np.abs(dm).max() < 1e-3                                     # north_star tolerance on the pose means
    dg.solve(R.make_opts(N=N, solver=1, seed=78), n_sweeps=10, bandwidth="lcv", product="gibbs")
    b = dg.bel[R.Pose2].cpu().numpy()
    idx = {lb: k for k, lb in enumerate(dg.packed.labels[R.Pose2])}
    win = {"x0": ((-3, 3), (-3, 3)), "x1": ((7, 13), (-3, 3)), "x2": ((12, 18), (6, 11)), "x3": ((7, 13), (15, 20)),
           "x4": ((-4, 4), (15, 20)), "x5": ((-8, -2), (6, 11)), "x6": ((-3, 3), (-3, 3))}
    for lb, w in win.items():
        p = b[idx[lb]]
        for d in range(2):
            assert 50 < ((w[d][0] < p[d]) & (p[d] < w[d][1])).sum(), (lb, d, p[d].mean())


def test_cut_lattice_through_separator_pipeline_one_rank_rccl():
    """The graph cut in two segments with MORE than four published separator rows per family (rome_conv_dev.mirror_map): segment 0
    run through SeparatorPipeline with the one-rank RCCL exchange; its tables equal the plain device-graph sweeps of the same
    segment, and the published blocks land in the receive buffer."""
    import torch
    import torch.distributed as dist
    from rome_jl_amd.distributed import SeparatorPipeline
    N = 100
    fg = _graph(20, N)
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    pk = dg.packed
    o = R.make_opts(N=N, solver=1, seed=5)
    # publish the proposals of 6 Pose2Pose2 rows, 3 bearing-range -> pose rows and 5 bearing-range -> landmark rows
    publish = [("p2p2", r) for r in (1, 4, 7, 10, 13, 16)] + [("br1", r) for r in (0, 2, 5)] + [("br0", r) for r in (0, 1, 3, 6, 8)]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = str(__import__("portutil").free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        pipe = SeparatorPipeline(dg, o, dist, 1, 0, publish, [], always_collective=True, depth=2)
        pipe.step(); pipe.drain(); torch.cuda.synchronize()
        outs = {f: pipe.out[0][f].cpu().numpy() for f in pipe.families}
        recv = pipe.recv[0].cpu().numpy()
    finally:
        dist.destroy_process_group()
    dg2 = R.DeviceGraph(fg); dg2.upload_beliefs(fg)
    dg2.conv_step(o, 0)
    C2 = dg2.tab["p2p2"]["C"]; Fb = pk.br["F"]
    p2 = dg2.prop[R.Pose2].cpu().numpy(); pl = dg2.prop[R.Point2].cpu().numpy()
    # (the pipeline draws stream_offset + row for every family: compare with per-family sweeps made with the same offsets)
    ref = {"p2p2": dg2.sweep_pose2pose2(o).cpu().numpy(), "br1": dg2.sweep_bearingrange(o, 1).cpu().numpy(), "br0": dg2.sweep_bearingrange(o, 0).cpu().numpy()}
    for f in ("p2p2", "br1", "br0"):
        assert np.array_equal(outs[f], ref[f][:outs[f].shape[0]]), f
    # Pose2 section of the payload: 6 + 3 blocks of 3N doubles in publish order; then the Point2 section (5 blocks of 2N)
    pose_blocks = [outs["p2p2"][r] for r in (1, 4, 7, 10, 13, 16)] + [outs["br1"][r] for r in (0, 2, 5)]
    got = recv[:9 * 3 * N].reshape(9, 3, N)
    assert np.array_equal(got, np.stack(pose_blocks))
    U = 6 * N
    off = -(-(9 * 3 * N) // U) * U
    gotl = recv[off:off + 5 * 2 * N].reshape(5, 2, N)
    assert np.array_equal(gotl, np.stack([outs["br0"][r] for r in (0, 1, 3, 6, 8)]))


@pytest.mark.parametrize("N", [100, 128, 66, 65, 64, 200])
def test_fused_graph_sweep_equals_per_family_launches_bit_for_bit(N):
    """rome_sweep_pose2_dev: the three families of a Pose2 / Point2 graph (plain tables: one fused launch, k_sweep_fused) against the
    per-family entry points with the same Philox streams -- identical bits; a table with multihypo columns takes the per-family
    launches inside the library and agrees as well.  N = 65: odd (scalar accesses in the fused kernel); N = 64 / 200: outside the fused
    kernel's range (per-family launches inside the library)."""
    fg = R.synth_mit_br(P=300, n_landmarks=60, N=N)
    R.dead_reckon_init(fg, seed=4)
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    assert not dg.tab["br"]["mh"]
    for solver in (R.SOLVER_NEWTON, R.SOLVER_CLOSED_FORM, R.SOLVER_GAUSS_NEWTON):   # (GAUSS_NEWTON: k_sweep_fused<., kSolverGaussNewton>, round 5)
        o = R.make_opts(N=N, solver=solver, seed=12, stream_offset=5 << 32)
        C2, Fb, Fb0 = dg.tab["p2p2"]["C"], dg.tab["br"]["F"], dg.tab["br"]["F0"]
        a2 = torch_empty(dg, (C2, 3, N)); a1 = torch_empty(dg, (Fb, 3, N)); a0 = torch_empty(dg, (Fb0, 2, N))
        dg.sweep_graph_pose2(o, a2, a1, a0)
        b2 = dg.sweep_pose2pose2(dg._opts_at(o, dg.STREAM_P2P2)); b1 = dg.sweep_bearingrange(dg._opts_at(o, dg.STREAM_BR1), 1)
        b0 = dg.sweep_bearingrange(dg._opts_at(o, dg.STREAM_BR0), 0)
        for x, y in ((a2, b2), (a1, b1), (a0, b0)):
            assert np.array_equal(x.cpu().numpy(), y.cpu().numpy())
    fg2 = _graph(20, 100)                                      # beehive with multihypo sightings: ONE fused launch too (k_sweep_fused_mh)
    dg2 = R.DeviceGraph(fg2); dg2.upload_beliefs(fg2)
    assert dg2.tab["br"]["mh"]
    o = R.make_opts(N=100, solver=1, seed=12)
    C2, Fb, Fb0 = dg2.tab["p2p2"]["C"], dg2.tab["br"]["F"], dg2.tab["br"]["F0"]
    a2 = torch_empty(dg2, (C2, 3, 100)); a1 = torch_empty(dg2, (Fb, 3, 100)); a0 = torch_empty(dg2, (Fb0, 2, 100))
    dg2.sweep_graph_pose2(o, a2, a1, a0)
    assert np.array_equal(a1.cpu().numpy(), dg2.sweep_bearingrange(dg2._opts_at(o, dg2.STREAM_BR1), 1).cpu().numpy())
    assert np.array_equal(a0.cpu().numpy(), dg2.sweep_bearingrange(dg2._opts_at(o, dg2.STREAM_BR0), 0).cpu().numpy())
    assert np.array_equal(a2.cpu().numpy(), dg2.sweep_pose2pose2(dg2._opts_at(o, dg2.STREAM_P2P2)).cpu().numpy())
    if N == 100:   # event-timed: the beehive sweep as three per-family launches against the one fused launch -> gpurun_out/
        import torch

        def timeit(fn, reps=300):
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return 1e3 * e0.elapsed_time(e1) / reps

        def three():
            dg2.sweep_bearingrange(dg2._opts_at(o, dg2.STREAM_BR1), 1, out=a1); dg2.sweep_pose2pose2(dg2._opts_at(o, dg2.STREAM_P2P2), out=a2)
            dg2.sweep_bearingrange(dg2._opts_at(o, dg2.STREAM_BR0), 0, out=a0)
        t3, t1 = timeit(three), timeit(lambda: dg2.sweep_graph_pose2(o, a2, a1, a0))
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r04_beehive_fused_sweep.txt"), "w") as f:
            f.write("beehive + multihypo sightings (synth_beehive_mh(20): %d p2p2 + %d br->pose + %d br->landmark rows, N=100), one convolution sweep, HIP events:\n"
                    "  three per-family launches %.2f us | ONE fused launch (k_sweep_fused_mh: wave bodies with multihypo columns + packed odometry) %.2f us\n" % (C2, Fb, Fb0, t3, t1))


def torch_empty(dg, shape):
    return dg.torch.zeros(shape, dtype=dg.torch.float64, device=dg.device)
