"""Canonical graph generators and the g2o exporter (host side).

Own counterparts of the reference's simulated-graph builders -- the data either side of the hot path:
  generateGraph_ZeroPose / buildGraphChain / generateGraph_TwoPoseOdo   src/canonical/GenerateCommon.jl:15-207
  generateGraph_Helix2D / _Helix2DSlew / _Helix2DSpiral                 src/canonical/GenerateHelix.jl:17-146
  generateGraph_Boxes2D                                                 src/canonical/GenerateBox.jl:12-137
  generateGraph_Beehive / generateGraph_Honeycomb                       src/canonical/GenerateBeehive.jl:15-75,
                                                                        src/canonical/GenerateHoneycomb.jl:55-232
  exportG2o                                                             src/services/g2oParser.jl:176-396
Pins: test/testGenerateHelix.jl:18-26,59-64,85-116, test/testG2oParser.jl:27-50, test/testBeehiveGrow.jl:44-48
(tests/test_canonical.py).

Every generator records the noise-free pose of each variable it adds (the reference's `:simulated` PPE) in
`fg.simulated[label]`, and can extend a graph it made earlier (`fg=`), like the reference's `dfg=` keyword.
The helix curve (`calcHelix_T`) lives in the unvendored IncrementalInference package; it is restated here from
the generator's documented behaviour and the test pins: p(t) = radius·(cis(π − 2πt) + 1 + spine(t)), t in turns,
heading = direction of dp/dt.
"""
import re

import numpy as np

from .factors import (MvNormal, Normal, Pose2, Point2, Pose2Pose2, PriorPose2, Pose2Point2BearingRange, PriorPoint2,
                      Point2Point2, Pose3Pose3)
from .graph import FactorGraph, initfg, se2_compose, se2_between


def _wrap(a):
    return float(np.arctan2(np.sin(a), np.cos(a)))


def _sim(fg):
    if not hasattr(fg, "simulated"):
        fg.simulated = {}
    return fg.simulated


def _pose_labels(fg, prefix="x"):
    pat = re.compile(r"^%s\d+$" % re.escape(prefix))
    return sorted((l for l in fg.ls() if pat.match(l)), key=lambda s: int(s[len(prefix):]))


def getPPE(fg, label, key="simulated"):
    """`getPPE(fg, label, key).suggested`: the `:simulated` reference estimate of the canonical generators, or -- for any other
    solveKey -- the point estimate stored by `setPPE` / `loadDFG` (fg.ppes)."""
    if key == "simulated":
        return _sim(fg)[label]
    try:
        return np.asarray(getattr(fg, "ppes", {})[label][key]["suggested"], dtype=float)
    except KeyError:
        raise KeyError("no %s point estimate for %s: call setPPE(fg) after a solve" % (key, label)) from None


def setPPE(fg, labels=None, solveKey="default"):
    """DFG `setPPE!(fg, label, solveKey)` for all (or the given) initialised variables: mean, max and suggested estimates of the
    current beliefs, computed on the device (rome_belief_stats, rome_kde_bandwidth, rome_kde_max; api.calcPPE) and stored in fg.ppes."""
    from . import api
    fg.ppes = getattr(fg, "ppes", None) or {}
    todo = [l for l in (labels or fg.ls()) if fg.isInitialized(l)]
    for vt in {fg.variables[l] for l in todo}:
        ls = [l for l in todo if fg.variables[l] is vt]
        est = api.calcPPE(np.stack([fg.getVal(l) for l in ls]))
        for i, l in enumerate(ls):
            fg.ppes.setdefault(l, {})[solveKey] = {k: est[k][i].copy() for k in est}
    return fg


def _predict(factor, prev):
    """Noise-free value of the new variable a relative factor implies from `prev` (what IIF's
    `_checkVariableByReference` computes for the `:simulated` estimate)."""
    if isinstance(factor, Pose2Pose2):
        q = se2_compose(prev, factor.Z.mu)
        q[2] = _wrap(q[2])
        return q
    if isinstance(factor, Pose2Point2BearingRange):
        b, r = factor.bearing.mu, factor.range.mu
        return np.array([prev[0] + r * np.cos(prev[2] + b), prev[1] + r * np.sin(prev[2] + b)])
    if isinstance(factor, Point2Point2):
        return np.asarray(prev[:2], dtype=float) + factor.Z.mu
    raise TypeError("no simulated prediction for %s" % type(factor).__name__)


def accumulateFactorMeans(fg, flabels):
    """Compose the measurement means of a prior followed by a chain of relative factors (IIF `accumulateFactorMeans`,
    pinned by test/testAccumulateFactors.jl:13-33 and test/testParametricSimulated.jl:62-65,149-152)."""
    flabels = list(flabels)
    _, _, f0 = fg.getFactor(flabels[0])
    if not f0.is_prior:
        raise ValueError("accumulateFactorMeans: the first factor must be a prior")
    val = np.asarray(f0.Z.mu, dtype=float).copy()
    for fl in flabels[1:]:
        val = _predict(fg.getFactor(fl)[2], val)
    return val


def _add_pose_canonical(fg, prev, label, factor, vartype=Pose2, override=None, postpose_cb=None):
    """One new variable + the factor that introduces it (+ its simulated estimate)."""
    if factor.is_prior:
        sim = np.asarray(factor.Z.mu, dtype=float).copy() if override is None else np.asarray(override, dtype=float)
        fg.addVariable(label, vartype)
        fg.addFactor([label], factor)
    else:
        sim = _predict(factor, _sim(fg)[prev]) if override is None else np.asarray(override, dtype=float)
        fg.addVariable(label, vartype)
        fg.addFactor([prev, label], factor)
    _sim(fg)[label] = sim
    if postpose_cb is not None:
        postpose_cb(fg, label)
    return label


def generateGraph_ZeroPose(fg=None, N=100, varType=Pose2, label="x0", mu0=None, cov0=None, postpose_cb=None):
    """`label` with a prior MvNormal(μ0, Σ0 = 0.01·I) (GenerateCommon.jl:61-92)."""
    fg = fg if fg is not None else initfg(N)
    d = varType.dim
    mu0 = np.zeros(d) if mu0 is None else np.asarray(mu0, dtype=float)
    cov0 = 0.01 * np.eye(d) if cov0 is None else np.asarray(cov0, dtype=float)
    if fg.exists(label):
        return fg
    prior = {Pose2: PriorPose2, Point2: PriorPoint2}[varType](MvNormal(mu0, cov0))
    _add_pose_canonical(fg, None, label, prior, vartype=varType, postpose_cb=postpose_cb)
    return fg


def buildGraphChain(fctData=None, fctType=Pose2Pose2, fg=None, N=100, stopAfter=None, postpose_cb=None):
    """Chain of variables joined by binary factors built from `fctData` (GenerateCommon.jl:107-159)."""
    if fctData is None:
        fctData = [MvNormal([10.0, 0.0, 0.0], 0.1 * np.eye(3)) for _ in range(3)]
    vt = fctType.variable_types[1]
    fg = fg if fg is not None else generateGraph_ZeroPose(N=N, varType=vt, postpose_cb=postpose_cb)
    last = _pose_labels(fg)[-1]
    count = int(last[1:])
    for i, dat in enumerate(fctData):
        if stopAfter is not None and i >= stopAfter:
            break
        count += 1
        last = _add_pose_canonical(fg, last, "x%d" % count, fctType(dat), vartype=vt, postpose_cb=postpose_cb)
    return fg


def generateGraph_TwoPoseOdo(N=100, addlandmark=True):
    """x0 (prior), x1 (odometry (10,0,0)), landmark l1 sighted from x1 (GenerateCommon.jl:173-197)."""
    fg = buildGraphChain([MvNormal([10.0, 0.0, 0.0], np.diag([1.0, 1.0, 0.01]))], N=N)
    if addlandmark:
        fg.addVariable("l1", Point2)
        f = Pose2Point2BearingRange(Normal(0.0, 0.01), Normal(20.0, 1.0))
        fg.addFactor(["x1", "l1"], f)
        _sim(fg)["l1"] = _predict(f, _sim(fg)["x1"])
    return fg


# ------------------------------------------------------------------------------------------ helix family
def calcHelix_T(start=0.0, stop=1.0, pointsperturn=20, radius=0.5, direction=-1, spine_t=None, xr_t=None, yr_t=None, h=1e-6):
    """Points of a planar helix ("slinky seen from above"): t [n] in turns, xy [n,2], heading [n].
    p(t) = radius·(cis(π + direction·2πt) + 1 + xr(t) + i·yr(t)); heading = arg(dp/dt) by central differences."""
    spine_t = spine_t if spine_t is not None else (lambda t: 0.0 + 0.0j)
    xr = xr_t if xr_t is not None else (lambda t: complex(spine_t(t)).real)
    yr = yr_t if yr_t is not None else (lambda t: complex(spine_t(t)).imag)

    def p(t):
        return radius * (np.exp(1j * (np.pi + direction * 2 * np.pi * t)) + 1.0 + xr(t) + 1j * yr(t))

    n = int(round((stop - start) * pointsperturn)) + 1
    ts = start + np.arange(n) / float(pointsperturn)
    # central differences; second-order one-sided at the start, where a spine like the spiral's t^0.4 has no left side
    pts = np.array([p(t) for t in ts])
    grad = np.array([(p(t + h) - p(t - h)) / (2 * h) if t - h >= 0.0 else (-3 * p(t) + 4 * p(t + h) - p(t + 2 * h)) / (2 * h)
                     for t in ts])
    return ts, np.stack([pts.real, pts.imag], 1), np.angle(grad)


def generateGraph_Helix2D(numposes=40, posesperturn=20, radius=10.0, spine_t=None, xr_t=None, yr_t=None, mu0=(0.0, 0.0, np.pi / 2),
                          Qd=None, fg=None, N=100, postpose_cb=None):
    """Poses x0..x{numposes-1} along a planar helix, exact relative-pose odometry Pose2Pose2(MvNormal(Δ, Qd)),
    prior on x0 at μ0 (GenerateHelix.jl:17-101).  Calling it again with a larger `numposes` and `fg=` extends the
    same graph; with a smaller or equal one nothing is added."""
    Qd = np.diag(np.square([0.1, 0.1, 0.05])) if Qd is None else np.asarray(Qd, dtype=float)
    mu0 = np.asarray(mu0, dtype=float)
    fg = fg if fg is not None else initfg(N)
    if not fg.exists("x0"):
        generateGraph_ZeroPose(fg=fg, mu0=mu0, postpose_cb=postpose_cb)
    have = len(_pose_labels(fg))
    if numposes <= have:
        return fg
    turns = numposes / float(posesperturn)
    _, xy, th = calcHelix_T(0.0, np.ceil(turns), posesperturn, radius=radius, spine_t=spine_t, xr_t=xr_t, yr_t=yr_t)
    frame = np.array([mu0[0], mu0[1], mu0[2] - np.pi / 2])   # the curve starts heading +y; μ0 re-bases it
    pose = lambda k: se2_compose(frame, np.array([xy[k, 0], xy[k, 1], th[k]]))
    for k in range(have, numposes):
        old, new = pose(k - 1), pose(k)
        delta = se2_between(old, new)
        delta[2] = _wrap(delta[2])
        new[2] = _wrap(new[2])
        _add_pose_canonical(fg, "x%d" % (k - 1), "x%d" % k, Pose2Pose2(MvNormal(delta, Qd)), override=new, postpose_cb=postpose_cb)
    return fg


def generateGraph_Helix2DSlew(numposes=40, slew_x=2.0 / 3.0, slew_y=0.0, **kw):
    """Helix pulled out at a constant rate along x (and y): a flattened slinky (GenerateHelix.jl:104-122)."""
    return generateGraph_Helix2D(numposes, spine_t=lambda t: slew_x * t + 1j * slew_y * t, **kw)


def generateGraph_Helix2DSpiral(numposes=100, rate_r=0.6, rate_a=6.0, **kw):
    """Helix whose spine follows a spiral, like flower petals (GenerateHelix.jl:125-146)."""
    return generateGraph_Helix2D(numposes, spine_t=lambda t: rate_r * (t ** 0.5) * np.exp(1j * rate_a * (t ** 0.4)), **kw)


# ------------------------------------------------------------------------------------------ boxes
def generateGraph_Boxes2D(numposes=16, length_x=15.0, length_y=None, slew_x=2.0 / 3.0, fg=None, N=100, postpose_cb=None):
    """Point2 "poses" driven clockwise around boxes that advance along x with 2/3 overlap: legs (+Lx, 0), (0, +Ly),
    (−slew·Lx, 0), (0, −Ly) joined by Point2Point2(MvNormal(leg, I)); whole boxes only (GenerateBox.jl:12-137)."""
    length_y = length_x if length_y is None else length_y
    fg = generateGraph_ZeroPose(fg=fg, N=N, varType=Point2, postpose_cb=postpose_cb)
    legs = [np.array([length_x, 0.0]), np.array([0.0, length_y]), np.array([-slew_x * length_x, 0.0]), np.array([0.0, -length_y])]
    for _ in range(int(np.ceil(numposes / 4.0))):
        for leg in legs:
            last = _pose_labels(fg)[-1]
            _add_pose_canonical(fg, last, "x%d" % (int(last[1:]) + 1), Point2Point2(MvNormal(leg, np.eye(2))), vartype=Point2,
                                postpose_cb=postpose_cb)
    return fg


# ------------------------------------------------------------------------------------------ beehive / honeycomb
def _add_landmark_beehive(fg, pose, atol=1.0):
    """Sight a landmark straight ahead at 20 m: Pose2Point2BearingRange(Normal(0, 0.03), Normal(20, 0.5)).  A landmark
    whose simulated position is within `atol` is re-sighted (perfect data association), otherwise l<pose number> is new
    (GenerateHoneycomb.jl:55-101; the reference's honeycomb recipe table is this rule evaluated ahead of time)."""
    f = Pose2Point2BearingRange(Normal(0.0, 0.03), Normal(20.0, 0.5))
    where = _predict(f, _sim(fg)[pose])
    for l in fg.ls():
        if fg.variables[l] is Point2 and l.startswith("l") and np.hypot(*(_sim(fg)[l] - where)) < atol:
            fg.addFactor([pose, l], f)
            return l
    label = "l%d" % int(pose[1:])
    fg.addVariable(label, Point2)
    fg.addFactor([pose, label], f)
    _sim(fg)[label] = where
    return label


def _hex_leg(fg, count, turn, addLandmarks, atol, postpose_cb):
    f = Pose2Pose2(MvNormal([10.0, 0.0, turn], np.diag(np.square([0.1, 0.1, 0.1]))))
    lbl = _add_pose_canonical(fg, "x%d" % count, "x%d" % (count + 1), f, postpose_cb=postpose_cb)
    if addLandmarks:
        _add_landmark_beehive(fg, lbl, atol)
    return count + 1


def _beehive_start(fg, N, mu0, addLandmarks, atol, postpose_cb):
    fg = fg if fg is not None else initfg(N)
    if fg.exists("x0"):
        return fg, int(_pose_labels(fg)[-1][1:])
    generateGraph_ZeroPose(fg=fg, mu0=mu0, postpose_cb=postpose_cb)
    if addLandmarks:
        _add_landmark_beehive(fg, "x0", atol)
    return fg, 0


def generateGraph_Honeycomb(poseCountTarget=36, direction="right", left_after=(41, 63, 78), addLandmarks=True, atol=1.0, fg=None,
                            N=100, postpose_cb=None):
    """Predetermined honeycomb: repeat {six legs turning +π/3 (one hexagon), one offset leg turning −π/3 ("right")}
    until x{poseCountTarget}; after poses `left_after` an extra leg turning +π/3 keeps the comb compact
    (GenerateHoneycomb.jl:3-51,186-232).  Growing an existing graph continues where it stopped."""
    fg, count = _beehive_start(fg, N, np.zeros(3), addLandmarks, atol, postpose_cb)
    sign = {"right": -1.0, "left": +1.0}
    while count < poseCountTarget:
        for _ in range(6):
            if count >= poseCountTarget:
                break
            count = _hex_leg(fg, count, np.pi / 3, addLandmarks, atol, postpose_cb)
        if count in left_after and count < poseCountTarget:
            count = _hex_leg(fg, count, sign["left"] * np.pi / 3, addLandmarks, atol, postpose_cb)
        if count < poseCountTarget:
            count = _hex_leg(fg, count, sign[direction] * np.pi / 3, addLandmarks, atol, postpose_cb)
    return fg


def generateGraph_Beehive(poseCountTarget=10, locality=1.0, yaw0=None, seed=None, addLandmarks=True, atol=1.0, fg=None, N=100,
                          postpose_cb=None):
    """A bee walking the edges of a honeycomb lattice: every step is a 10 m leg turning ±π/3, the turn direction
    flips with probability 1/(1+locality) (GenerateBeehive.jl:15-75)."""
    rng = np.random.default_rng(seed)
    yaw0 = float(rng.choice([0.0, -2 * np.pi / 3, 2 * np.pi / 3])) if yaw0 is None else float(yaw0)
    fg, count = _beehive_start(fg, N, np.array([0.0, 0.0, yaw0]), addLandmarks, atol, postpose_cb)
    left = bool(rng.integers(2))
    flip = 1.0 / (1.0 + locality)
    while count < poseCountTarget:
        if rng.uniform() < flip:
            left = not left
        count = _hex_leg(fg, count, (np.pi / 3) if left else (-np.pi / 3), addLandmarks, atol, postpose_cb)
    return fg


def synth_beehive_mh(poseCountTarget=36, ambiguous_every=2, weights=(0.5, 0.5), N=100, **kw):
    """BASELINE configs[3] in one graph: the honeycomb of `generateGraph_Honeycomb` (legs μ = (10, 0, ±π/3), Σ = diag(0.1²);
    landmarks sighted straight ahead at 20 m, GenerateHoneycomb.jl:59-100) in which every `ambiguous_every`-th RE-sighting of a
    known landmark is an uncertain data association between that landmark and the nearest other one:
    `addFactor!(fg, [pose; l_a; l_b], p2br, multihypo=[1.0; 0.5; 0.5])` as in test/testMultimodalRangeBearing.jl:53.
    -> fg (simulated ground truth kept under fg._sim as for the other generators)"""
    base = generateGraph_Honeycomb(poseCountTarget=poseCountTarget, N=N, **kw)
    sim = _sim(base)
    fg = initfg(N)
    fg._sim = dict(sim)
    for l, t in base.variables.items():
        fg.addVariable(l, t)
    lms = [l for l, t in base.variables.items() if t is Point2]
    seen, resight = set(), 0
    for flabel, labels, f in base.factors:
        if isinstance(f, Pose2Point2BearingRange):
            pose, lm = labels
            if lm in seen and len(lms) > 1:
                resight += 1
                if resight % ambiguous_every == 0:
                    other = min((l for l in lms if l != lm), key=lambda l: np.hypot(*(sim[l] - sim[lm])))
                    fg.addFactor([pose, lm, other], f, multihypo=[1.0, weights[0], weights[1]])
                    continue
            seen.add(lm)
        fg.addFactor(labels, f)
    return fg


# ------------------------------------------------------------------------------------------ g2o export
def _jl(x):
    """Shortest round-trip decimal of a double, written the way Julia's `string(::Float64)` writes it."""
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Inf" if x > 0 else "-Inf"
    r = repr(x)
    mant, _, ex = r.partition("e")
    if ex:                                   # python: 1e-05 / 1.5e+20  -> julia: 1.0e-5 / 1.5e20
        if "." not in mant:
            mant += ".0"
        return "%se%d" % (mant, int(ex))
    a = abs(x)
    if a != 0.0 and (a < 1e-4 or a >= 1e6):    # python keeps fixed notation longer than julia does: re-write with an exponent
        for p in range(0, 17):                 # shortest mantissa that round-trips
            cand = "%.*e" % (p, x)
            if float(cand) == x:
                break
        m, e = cand.split("e")
        if "." not in m:
            m += ".0"
        return "%se%d" % (m, int(e))
    return r


def _invcov(cov):
    """Information matrix the way `invcov(::MvNormal)` gets it: Cholesky Σ = UᵀU, X = U⁻¹ (column by column, LAPACK
    trti2 order), Λ = X Xᵀ (lauu2 order).  Spelled out because the order fixes the signed zeros the reference's exporter
    prints for diagonal Σ ("100.0 0.0 -0.0 100.0 -0.0 100.0", test/testG2oParser.jl:29)."""
    A = np.asarray(cov, dtype=float)
    n = A.shape[0]
    U = np.linalg.cholesky(A).T.copy()
    X = np.zeros((n, n))
    for j in range(n):
        X[j, j] = 1.0 / U[j, j]
        col = U[:j, j].copy()                 # col <- X[:j,:j] (upper triangular) · U[:j, j], in place (trmv order)
        for k in range(j):
            if col[k] != 0.0:
                t = col[k]
                for i in range(k):
                    col[i] += t * X[i, k]
                col[k] = t * X[k, k]
        for i in range(j):
            X[i, j] = col[i] * (-X[j, j])
    L = np.zeros((n, n))
    for i in range(n):
        for j in range(i, n):
            s = None
            for k in range(j, n):
                term = X[i, k] * X[j, k]
                s = term if s is None else s + term
            L[i, j] = L[j, i] = s
    return L


class _VarIds:
    def __init__(self):
        self.ids = {}

    def __call__(self, label):
        if label not in self.ids:
            self.ids[label] = len(self.ids)
        return self.ids[label]


def stringG2o(fg, flabel, ids):
    """One g2o record for a factor (g2oParser.jl:190-273)."""
    _, labels, f = fg.getFactor(flabel)
    v = [ids(l) for l in labels]
    if isinstance(f, Pose2Pose2):
        I = _invcov(f.Z.cov)
        nums = [f.Z.mu[0], f.Z.mu[1], f.Z.mu[2], I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]]
        return "EDGE_SE2 %d %d " % (v[0], v[1]) + " ".join(_jl(x) for x in nums)
    if isinstance(f, Pose2Point2BearingRange):
        nums = [f.bearing.mu, f.range.mu, 1.0 / f.bearing.sigma ** 2, 0.0, 1.0 / f.range.sigma ** 2]
        return "LANDMARK %d %d " % (v[0], v[1]) + " ".join(_jl(x) for x in nums)
    if isinstance(f, Pose3Pose3):
        from .factors import getPoint, Pose3
        I = _invcov(f.Z.cov)
        R = np.asarray(getPoint(Pose3, f.Z.mu))[3:].reshape(3, 3, order="F")   # 12 doubles [t, R column-major]
        q = _quat_from_R(R)
        nums = list(f.Z.mu[:3]) + [q[1], q[2], q[3], q[0]] + [I[i, j] for i in range(6) for j in range(i, 6)]
        return "EDGE_SE3:QUAT %d %d " % (v[0], v[1]) + " ".join(_jl(x) for x in nums)
    raise TypeError("unknown factor type %s" % type(f).__name__)


def _quat_from_R(R):
    """(w, x, y, z) of a rotation matrix, w ≥ 0."""
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    if w > 1e-6:
        return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(max(0.0, 1.0 + R[i, i] - R[j, j] - R[k, k])) * 2.0
    q = np.zeros(4)
    q[0] = (R[k, j] - R[j, k]) / s
    q[1 + i] = s / 4.0
    q[1 + j] = (R[j, i] + R[i, j]) / s
    q[1 + k] = (R[k, i] + R[i, k]) / s
    return q if q[0] >= 0 else -q


def exportG2o(fg, filename="/tmp/test.txt", ignorePriors=True, posePrefix="x", estimates=None, varIntLabel=None):
    """Write the graph as g2o records: pose variables in label order, each contributing the not-yet-written factors
    attached to it in creation order; variables are numbered from 0 in order of first appearance, or as `varIntLabel`
    (label -> int) says.  With `estimates` (label -> coordinates; the reference's `solveKey=` PPEs) VERTEX_SE2 /
    VERTEX_SE3:QUAT records of the numbered variables come first (g2oParser.jl:298-396, test/testG2oExportSE3.jl)."""
    ids = _VarIds()
    if varIntLabel:
        ids.ids.update(varIntLabel)
    done = set()
    lines = []
    if isinstance(estimates, str):     # a solveKey: the stored point estimates (reference: exportG2o(fg; solveKey=:parametric))
        estimates = {l: getPPE(fg, l, estimates) for l in (varIntLabel or {})}
    if estimates is not None:
        if not varIntLabel:
            raise ValueError("exportG2o: vertex records need varIntLabel (as the reference's exporter does)")
        for label, i in varIntLabel.items():
            c = np.asarray(estimates[label], dtype=float)
            if fg.variables[label] is Pose2:
                lines.append("VERTEX_SE2 %d %s" % (i, " ".join(_jl(x) for x in c)))
            else:
                from .factors import getPoint, Pose3
                if fg.variables[label] is not Pose3:
                    raise TypeError("exportG2o does not support %s vertices" % fg.variables[label])
                q = _quat_from_R(np.asarray(getPoint(Pose3, c))[3:].reshape(3, 3, order="F"))
                lines.append("VERTEX_SE3:QUAT %d %s" % (i, " ".join(_jl(x) for x in (c[0], c[1], c[2], q[1], q[2], q[3], q[0]))))
    for pose in _pose_labels(fg, posePrefix):
        for flabel, labels, f in fg.factors:
            if pose not in labels or flabel in done:
                continue
            done.add(flabel)
            if f.is_prior:
                if ignorePriors:
                    continue
                raise TypeError("exportG2o: prior factors have no g2o record here")
            lines.append(stringG2o(fg, flabel, ids))
    with open(filename, "w") as fh:
        for ln in lines:
            fh.write(ln + "\n")
    return filename
