"""Clique-level batching behind the plugin surface (rome_clique_proposals / R.proposalbeliefs): every proposal of a variable -- or
of a whole clique -- from HOST beliefs in one library call, against the per-factor path (`approxConv`, one call per convolution,
what the first Julia shim did) and against the device-resident graph sweep, bit for bit; and its PCIe-inclusive rate."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _pkg():
    global R
    import rome_jl_amd
    R = rome_jl_amd
    R.default_context()
    yield


def _hexagon(N=100):
    fg = R.generateGraph_Hexagonal(N=N)
    R.dead_reckon_init(fg, seed=5)
    fg.initVariable("l1", np.array([[20.0], [0.0]]) + np.random.default_rng(0).standard_normal((2, N)))
    return fg


@pytest.mark.parametrize("solver", [0, 1, 2])
def test_proposalbeliefs_equals_per_factor_approxconv_hexagon(solver):
    from rome_jl_amd.clique import FAMILY_STREAM
    fg = _hexagon()
    # one call per destination variable (what a Gibbs step on that variable needs) ...
    for dest in ("x0", "x3", "x6", "l1"):
        props, batch = R.proposalbeliefs(fg, dest, solver=solver, seed=9, stream_offset=1000)
        assert len(props) == sum(1 for _, labels, _ in fg.factors if dest in labels)
        for (flabel, d), got in props.items():
            fam, r = batch.rows[(flabel, d)]
            ref = R.approxConv(fg, flabel, d, solver=solver, seed=9, stream_offset=1000 + FAMILY_STREAM[fam] + r)
            assert np.array_equal(got, ref), (dest, flabel, np.abs(got - ref).max())
    # ... and the whole clique (all variables, all factors) in one call
    props, batch = R.proposalbeliefs(fg, list(fg.variables), solver=solver, seed=9)
    assert len(props) == sum(len(labels) for _, labels, _ in fg.factors)
    for (flabel, d), got in props.items():
        fam, r = batch.rows[(flabel, d)]
        ref = R.approxConv(fg, flabel, d, solver=solver, seed=9, stream_offset=FAMILY_STREAM[fam] + r)
        assert np.array_equal(got, ref), (flabel, d)


def _manhattan500(N=100):
    import os
    fg = R.loadG2o(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manhattan.g2o"), N=N, max_edges=500)
    R.dead_reckon_init(fg, seed=3)
    return fg


def test_clique_call_equals_device_graph_sweep_manhattan500_and_rate():
    """The first 500 Manhattan edges (the graph of the reference's examples/fg-after-solve.tar.gz) as ONE clique: the host-belief
    call reproduces the device-resident sweep of the same table bit for bit, and from host memory to host memory it delivers
    more than 1e6 convolutions/s (the per-factor call: ~2e4)."""
    import torch
    fg = _manhattan500()
    pairs = []
    for flabel, labels, f in fg.factors:
        if isinstance(f, R.Pose2Pose2):
            pairs += [(flabel, labels[1]), (flabel, labels[0])]          # row 2f: solve the 2nd variable, 2f+1: the 1st
    pairs += [(flabel, labels[0]) for flabel, labels, f in fg.factors if isinstance(f, R.PriorPose2)]
    batch = R.CliqueBatch(fg, pairs)
    opts = R.make_opts(N=fg.N, solver=1, seed=21)
    got = batch.run(opts)["p2p2"]
    dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
    # (variables are numbered in first-use order by the batch and in insertion order by the device graph: the tables differ,
    #  the proposals must not)
    ref = dg.sweep_pose2pose2(opts).cpu().numpy()
    torch.cuda.synchronize()
    assert got.shape == ref.shape == (1001, 3, 100)
    assert np.array_equal(got, ref)
    # per-factor path on a sample of rows
    from rome_jl_amd.clique import FAMILY_STREAM
    for r in (0, 1, 500, 999, 1000):
        flabel, d = pairs[r]
        assert np.array_equal(got[r], R.approxConv(fg, flabel, d, solver=1, seed=21, stream_offset=FAMILY_STREAM["p2p2"] + r))
    # PCIe-inclusive rate, host beliefs -> host proposals
    batch.run(opts)
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        batch.run(opts)
    dt = (time.perf_counter() - t) / reps
    rate = len(pairs) / dt
    print("clique call: %d convolutions in %.1f us -> %.3g conv/s (host beliefs -> host proposals)" % (len(pairs), dt * 1e6, rate))
    assert rate > 1e6, rate
    # a small clique (one variable's Gibbs step): still a single call
    t = time.perf_counter()
    for _ in range(reps):
        R.proposalbeliefs(fg, "x10", seed=21)
    print("proposalbeliefs(x10): %.1f us per call" % ((time.perf_counter() - t) / reps * 1e6))


def test_clique_layouts_and_errors():
    """AoS coordinates and the reference's native point containers through the same entry (the Julia shim passes `pointer(vals)`)."""
    import ctypes as C
    from rome_jl_amd import _lib
    from rome_jl_amd.clique import CliqueHost
    fg = _hexagon(N=64)
    pairs = [(fl, lb[1]) for fl, lb, f in fg.factors if isinstance(f, R.Pose2Pose2)]
    batch = R.CliqueBatch(fg, pairs)
    soa = batch.run(R.make_opts(N=64, solver=1, seed=4))["p2p2"]
    ctx = R.default_context()
    bel = batch.beliefs(R.Pose2)
    for layout in (_lib.LAYOUT_AOS, _lib.LAYOUT_AOS_POINTS):
        q = CliqueHost()
        b = np.ascontiguousarray(bel.transpose(0, 2, 1))                                # [V][N][3]
        if layout == _lib.LAYOUT_AOS_POINTS:
            b = R.coords_to_points(3, b.reshape(-1, 3)).reshape(bel.shape[0], 64, 6)
        rows = np.array(batch.fam_rows["p2p2"], dtype=np.int32)
        mu = np.array(batch.tabs["p2p2"]["mu"]); cov = np.array(batch.tabs["p2p2"]["spread"])
        width = 6 if layout == _lib.LAYOUT_AOS_POINTS else 3
        out = np.zeros((len(rows), 64, width))
        q.n_pose2 = bel.shape[0]; q.bel_pose2 = b.ctypes.data_as(C.c_void_p)
        q.n_p2p2, q.f_p2p2 = len(rows), len(mu)
        q.p2p2_rows4 = rows.ctypes.data_as(C.c_void_p); q.p2p2_mu = mu.ctypes.data_as(C.c_void_p); q.p2p2_cov = cov.ctypes.data_as(C.c_void_p)
        q.out_p2p2 = out.ctypes.data_as(C.c_void_p)
        o = R.make_opts(N=64, solver=1, seed=4, layout=layout)
        _lib.check(_lib.load().rome_clique_proposals(ctx.handle, C.byref(o), C.byref(q)), ctx.handle)
        c = R.points_to_coords(3, out.reshape(-1, 6)).reshape(len(rows), 64, 3) if layout == _lib.LAYOUT_AOS_POINTS else out
        d = c.transpose(0, 2, 1) - soa
        d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
        assert np.abs(d).max() < (1e-12 if layout == _lib.LAYOUT_AOS else 1e-9)
    # a row that points outside the clique's arrays is rejected, not read
    q = CliqueHost()
    rows = np.array([[0, 0, 0, 99]], dtype=np.int32)
    q.n_pose2 = bel.shape[0]; q.bel_pose2 = bel.ctypes.data_as(C.c_void_p)
    q.n_p2p2, q.f_p2p2 = 1, 1
    mu = np.zeros(3); cov = np.eye(3); out = np.zeros((1, 3, 64))
    q.p2p2_rows4 = rows.ctypes.data_as(C.c_void_p); q.p2p2_mu = mu.ctypes.data_as(C.c_void_p); q.p2p2_cov = cov.ctypes.data_as(C.c_void_p)
    q.out_p2p2 = out.ctypes.data_as(C.c_void_p)
    assert _lib.load().rome_clique_proposals(ctx.handle, C.byref(R.make_opts(N=64)), C.byref(q)) == _lib.ERR_INVALID_ARG


def test_predictbelief_multiplies_all_proposals_of_a_variable():
    """IIF predictbelief = proposalbeliefs + manifoldProduct, two library calls: x3 of the hexagon has two odometry neighbours,
    the predicted belief is tighter than either proposal and sits between / on them; solveGraph runs with the reference's product."""
    fg = _hexagon()
    props, _ = R.proposalbeliefs(fg, "x3", seed=4)
    assert len(props) == 2
    pb = R.predictbelief(fg, "x3", seed=4)
    assert pb.shape == (3, fg.N) and np.isfinite(pb).all()
    sd = np.array([p[:2].std(axis=1) for p in props.values()])
    assert (pb[:2].std(axis=1) < sd.max(axis=0)).all()                       # the product is tighter than the wider proposal
    m = np.array([p[:2].mean(axis=1) for p in props.values()])
    assert np.linalg.norm(pb[:2].mean(axis=1) - m.mean(axis=0)) < 2.0
    one = R.predictbelief(fg, "x0", factor_labels=[fg.factors[0][0]], seed=4)   # the prior alone: its samples
    assert one.shape == (3, fg.N)
    fg2 = R.generateGraph_Hexagonal(N=100)
    dg = R.solveGraph(fg2, n_sweeps=6, product="gibbs")
    x6 = fg2.getVal("x6")
    assert np.hypot(x6[0].mean(), x6[1].mean()) < 3.0                          # the loop closes on the origin
