"""The reference's solved factor graph, read from the data artefact itself (tests/golden/fg-after-solve.tar.gz: byte copy of
examples/fg-after-solve.tar.gz, the DistributedFactorGraphs save of the first 500 Manhattan edges after a reference `solveTree!`)
through this package's `loadDFG` -- full double precision, unlike the float32 particles of manhattan500_reference_solve.npz.

CPU part: file format + oracle restatements against what the reference stored next to its particles (bandwidths, point estimates).
GPU part (marked): the same through the C ABI."""
import os

import numpy as np
import pytest

import oracle as ro

HERE = os.path.dirname(os.path.abspath(__file__))
ART = os.path.join(HERE, "golden", "fg-after-solve.tar.gz")
_cache = {}


def _load():
    if "fg" not in _cache:
        import rome_jl_amd as R
        fg = R.loadDFG(ART)
        V = len(fg.ls())
        _cache["fg"] = fg
        _cache["bel"] = np.stack([fg.getVal("x%d" % i) for i in range(V)])                       # [361, 3, 100]
        _cache["bw"] = np.stack([fg.bws["x%d" % i] for i in range(V)])
        _cache["ppe"] = {k: np.stack([fg.ppes["x%d" % i]["default"][k] for i in range(V)]) for k in ("suggested", "max", "mean")}
    return _cache


def test_loaddfg_reads_the_early_2020_layout_and_agrees_with_the_extracted_fixture():
    import rome_jl_amd as R
    c = _load()
    fg = c["fg"]
    assert len(fg.ls()) == 361 and fg.N == 100 and all(fg.variables[l] is R.Pose2 for l in fg.ls())
    kinds = [type(f).__name__ for _, _, f in fg.factors]
    assert kinds.count("Pose2Pose2") == 500 and kinds.count("PriorPose2") == 1 and len(kinds) == 501
    d = np.load(os.path.join(HERE, "golden", "manhattan500_reference_solve.npz"))
    assert np.abs(c["bel"] - d["particles"].transpose(0, 2, 1)).max() < 2e-6            # the .npz keeps float32 particles
    assert np.array_equal(c["bw"], d["bandwidth"]) and np.array_equal(c["ppe"]["max"], d["ppe"][:, 1])
    table = {(int(a[1:]), int(b[1:])): f for _, (a, *rest), f in [(x, y, z) for x, y, z in fg.factors if len(y) == 2] for b in rest}
    for (i, j), mu, cov in zip(d["edges"], d["mu"], d["cov"]):
        f = table[(int(i), int(j))]
        assert np.array_equal(f.Z.mu, mu) and np.array_equal(f.Z.cov, cov)
    prior = [f for _, ls, f in fg.factors if len(ls) == 1][0]
    assert np.array_equal(prior.Z.mu, d["prior_mu"]) and np.array_equal(prior.Z.cov, d["prior_cov"])


def test_oracle_point_estimates_and_bandwidths_at_full_precision():
    c = _load()
    bel, bw, ppe = c["bel"], c["bw"], c["ppe"]
    # getKDEMax: every stored coordinate, to rounding
    assert np.abs(ro.kde_max(bel, bw) - ppe["max"]).max() < 1e-12
    # mean: arithmetic mean of the coordinates (x, y exactly; heading wherever the samples do not wrap)
    assert np.abs(bel[:, :2].mean(2) - ppe["mean"][:, :2]).max() < 1e-12
    tight = np.ptp(bel[:, 2], axis=1) < 3.0
    assert tight.sum() > 300 and np.abs(bel[tight, 2].mean(1) - ppe["mean"][tight, 2]).max() < 1e-12
    # suggested = (mean x, mean y, max-density heading)
    assert np.array_equal(ppe["suggested"][:, :2], ppe["mean"][:, :2]) and np.array_equal(ppe["suggested"][:, 2], ppe["max"][:, 2])
    # bandwidths: headings to 1e-6 (the reference's circular search runs to Optim's tolerance), x / y inside its 1 % stopping rule
    h = ro.kde_bandwidths(bel, 0b100, 1e-2, 1e-8)
    assert np.abs(h[:, 2] / bw[:, 2] - 1).max() < 1e-6, np.abs(h[:, 2] / bw[:, 2] - 1).max()      # measured 4.0e-7
    assert np.abs(h[:, :2] / bw[:, :2] - 1).max() < 8e-3


@pytest.mark.gpu
def test_device_point_estimates_and_bandwidths_at_full_precision():
    import rome_jl_amd as R
    c = _load()
    bel, bw, ppe = c["bel"], c["bw"], c["ppe"]
    assert np.abs(R.kde_max(bel, bw) - ppe["max"]).max() < 1e-12
    mean, _ = R.belief_stats(bel)
    assert np.abs(mean[:, :2] - ppe["mean"][:, :2]).max() < 1e-12
    tight = np.ptp(bel[:, 2], axis=1) < 3.0
    assert np.abs(np.arctan2(np.sin(mean[tight, 2] - ppe["mean"][tight, 2]), np.cos(mean[tight, 2] - ppe["mean"][tight, 2]))).max() < 1e-12
    h = R.kde_bandwidth(bel, 0b100, 1e-2, 1e-8)
    # headings: the device returns the stationary point of the leave-one-out likelihood; the reference's GoldenSection result is
    # defined to the rounding noise of its likelihood values (flat likelihoods: ~1e-5), the two agree within that
    assert np.abs(h[:, 2] / bw[:, 2] - 1).max() < 5e-5 and np.median(np.abs(h[:, 2] / bw[:, 2] - 1)) < 2e-6
    assert np.abs(h[:, :2] / bw[:, :2] - 1).max() < 8e-3
    est = R.calcPPE(bel, bw)
    assert np.abs(est["suggested"] - ppe["suggested"])[:, :2].max() < 1e-12 and np.abs(est["suggested"][:, 2] - ppe["suggested"][:, 2]).max() < 1e-12
