"""Factor-graph container, g2o import and synthetic g2o-shaped generators (host side).

Mirrors the slice of the DistributedFactorGraphs / RoME API the hot path is driven through:
  initfg / addVariable! / addFactor! / initVariable! / getVal      (used all over test/*.jl)
  importG2o, parseG2oInstruction!                                  src/services/g2oParser.jl:39-171
  generateGraph_Hexagonal / generateGraph_Circle                   src/canonical/GenerateCircular.jl:31-94
and packs a graph into the flat factor / convolution tables the device sweep consumes.
"""
import contextlib as _contextlib
import gc as _gc
from collections import OrderedDict

import numpy as np

from .factors import (MvNormal, Normal, Uniform, Pose2, Point2, Pose3, Pose2Pose2, PriorPose2, Pose2Point2BearingRange,
                      Pose3Pose3, PriorPose3, PriorPoint2)


@_contextlib.contextmanager
def gc_paused():
    """the host-side plan builders (elimination structure, Bayes-tree levels, initAll! rounds) allocate ~1e5 - 1e6 small objects that all stay
    alive: the cyclic collector's generation scans find nothing to free and cost a third of such a build (Manhattan-3500 elimination
    structure 0.33 -> 0.23 s) -- paused for the duration, restored afterwards"""
    on = _gc.isenabled()
    _gc.disable()
    try:
        yield
    finally:
        if on:
            _gc.enable()


class FactorGraph:
    def __init__(self, N=100):
        self.N = int(N)                      # getSolverParams(fg).N
        self.variables = OrderedDict()       # label -> vartype
        self.factors = []                    # (label, [var labels], factor)
        self.vals = {}                       # label -> (dim, N) coordinates (belief particles)
        self.multihypo = {}                  # factor label -> (w1, w2)
        self.nullhypo = {}                   # factor label -> p (IIF addFactor!(...; nullhypo=p), test/testPose3Pose3NH.jl:118)
        self._findex = {}                    # factor label -> (label, [var labels], factor)   (getFactor in O(1))

    # -- DFG-style API --
    def addVariable(self, label, vartype):
        if label in self.variables:
            raise KeyError("variable %s already exists" % label)
        self.variables[label] = vartype
        return label

    def exists(self, label):
        return label in self.variables

    def ls(self):
        return list(self.variables)

    def addFactor(self, labels, factor, multihypo=None, nullhypo=None):
        """nullhypo=p (IIF kwarg, test/testPose3Pose3NH.jl:118): with probability p the factor does not apply to a particle.
        multihypo=[1.0, w1, w2] (IIF kwarg, test/testMultimodalRangeBearing.jl:53): the SECOND variable of a two-variable factor is
        labels[1] with probability w1 or labels[2] with probability w2 -- a Pose2Point2BearingRange over [pose, l1, l2] (every use in
        the reference) or a Pose2Pose2 over [a, b1, b2] (IIF accepts the keyword on any factor)."""
        labels = list(labels)
        if multihypo is not None:
            if not isinstance(factor, (Pose2Point2BearingRange, Pose2Pose2)) or len(labels) != 3 or len(multihypo) != 3:
                raise ValueError("multihypo is supported for Pose2Point2BearingRange over [pose, l1, l2] and Pose2Pose2 over [a, b1, b2]")
            w = [float(x) for x in multihypo]
            if w[0] != 1.0 or abs(w[1] + w[2] - 1.0) > 1e-12 or min(w[1:]) < 0:
                raise ValueError("multihypo must be [1.0, w1, w2] with w1 + w2 = 1")
            second = factor.variable_types[1]
            if self.variables.get(labels[2]) is not second:
                raise TypeError("multihypo: %s must be a %s variable" % (labels[2], second.__name__ if hasattr(second, "__name__") else second))
            extra, labels_chk = labels[2], labels[:2]
        else:
            extra, labels_chk = None, labels
        for l, t in zip(labels_chk, factor.variable_types):
            if l not in self.variables:
                raise KeyError("addFactor: unknown variable %s" % l)
            if self.variables[l] is not t:
                raise TypeError("addFactor: %s expects %s for %s" % (type(factor).__name__, t, l))
        if len(labels_chk) != len(factor.variable_types):
            raise ValueError("addFactor: wrong number of variables")
        # DFG numbers the factors of one variable list f1, f2, ...; after deleteFactor a plain count would hand out a label that is
        # still in use: take the smallest free suffix instead
        used = {f[0] for f in self.factors}
        k = 1
        while "".join(labels) + "f%d" % k in used:
            k += 1
        flabel = "".join(labels) + "f%d" % k
        self.factors.append((flabel, labels, factor))
        self._findex[flabel] = self.factors[-1]
        if extra is not None:
            self.multihypo[flabel] = (w[1], w[2])
        if nullhypo:
            if not 0.0 <= float(nullhypo) <= 1.0:
                raise ValueError("nullhypo must be a probability")
            self.nullhypo[flabel] = float(nullhypo)
        return flabel

    def deleteFactor(self, flabel):
        """DFG `deleteFactor!(fg, label)`"""
        k = [i for i, f in enumerate(self.factors) if f[0] == flabel]
        if not k:
            raise KeyError(flabel)
        self.factors.pop(k[0])
        self._findex.pop(flabel, None)
        self.multihypo.pop(flabel, None)
        self.nullhypo.pop(flabel, None)

    def getFactor(self, flabel):
        f = getattr(self, "_findex", {}).get(flabel)
        if f is not None:
            return f
        for f in self.factors:               # (graphs whose factor list was filled directly, e.g. by a loader)
            if f[0] == flabel:
                return f
        raise KeyError(flabel)

    def initVariable(self, label, coords):
        """coords: (dim, N) particle coordinates."""
        t = self.variables[label]
        a = np.ascontiguousarray(coords, dtype=np.float64)
        if a.shape != (t.dim, self.N):
            raise ValueError("initVariable(%s): expected %s, got %s" % (label, (t.dim, self.N), a.shape))
        self.vals[label] = a

    def getVal(self, label):
        return self.vals[label]

    def isInitialized(self, label):
        return label in self.vals


def initfg(N=100):
    return FactorGraph(N)


def fifoFreeze(fg, qfl=None):
    """IIF `fifoFreeze!(fg)` with `getSolverParams(fg).qfl` (test/testFixedLagFG.jl:33-35,89): all but the newest `qfl` variables, in the
    order they were added, become marginalized; returns their labels (and records them in fg.marginalized)."""
    if qfl is None:
        qfl = getattr(fg, "qfl", None)
    if qfl is None or qfl < 0:
        raise ValueError("fifoFreeze needs the fixed-lag window length qfl")
    order = fg.ls()
    frozen = order[:max(len(order) - int(qfl), 0)]
    fg.marginalized = set(getattr(fg, "marginalized", set())) | set(frozen)
    return sorted(fg.marginalized, key=order.index)


def isMarginalized(fg, label):
    return label in getattr(fg, "marginalized", set())


# ------------------------------------------------------------------------------------------ g2o
def importG2o(path):
    """Every line split on blanks (src/services/g2oParser.jl:39-49)."""
    out = []
    with open(path) as f:
        for ln in f:
            pieces = ln.split()
            if pieces:
                out.append(pieces)
    return out


def parseG2oInstruction(fg, ins):
    """EDGE_SE2 / VERTEX_SE2 semantics of src/services/g2oParser.jl:62-122:
    μ = fields 4-6, Λ from the upper triangle (fields 7-12), Σ = inv(Λ) symmetrised."""
    if ins[0] == "VERTEX_SE2":
        lbl = "x" + ins[1]
        if not fg.exists(lbl):
            fg.addVariable(lbl, Pose2)
        fg.vertex_init = getattr(fg, "vertex_init", {})
        fg.vertex_init[lbl] = np.array([float(ins[2]), float(ins[3]), float(ins[4])])
    elif ins[0] == "EDGE_SE2":
        a, b = "x" + ins[1], "x" + ins[2]
        v = [float(x) for x in ins[3:12]]
        mu = np.array(v[0:3])
        info = np.array([[v[3], v[4], v[5]], [v[4], v[6], v[7]], [v[5], v[7], v[8]]])
        cov = np.linalg.inv(info)
        cov = (cov + cov.T) / 2.0
        for l in (a, b):
            if not fg.exists(l):
                fg.addVariable(l, Pose2)
        fg.addFactor([a, b], Pose2Pose2(MvNormal(mu, cov)))
    elif ins[0] == "VERTEX_SE3:QUAT":      # id x y z qx qy qz qw  (g2oParser.jl:76-92)
        from scipy.spatial.transform import Rotation as Rot
        lbl = "x" + ins[1]
        v = [float(x) for x in ins[2:9]]
        if not fg.exists(lbl):
            fg.addVariable(lbl, Pose3)
        fg.vertex_init = getattr(fg, "vertex_init", {})
        fg.vertex_init[lbl] = np.concatenate([v[0:3], Rot.from_quat(v[3:7]).as_rotvec()])
    elif ins[0] == "EDGE_SE3:QUAT":        # i j x y z qx qy qz qw + 21 upper-triangle information entries (g2oParser.jl:124-168)
        from scipy.spatial.transform import Rotation as Rot
        a, b = "x" + ins[1], "x" + ins[2]
        v = [float(x) for x in ins[3:31]]
        mu = np.concatenate([v[0:3], Rot.from_quat(v[3:7]).as_rotvec()])   # coordinates [t; vee(Log(dR))]
        info = np.zeros((6, 6))
        info[np.triu_indices(6)] = v[7:28]
        info = info + np.triu(info, 1).T
        cov = np.linalg.inv(info)
        cov = (cov + cov.T) / 2.0
        for l in (a, b):
            if not fg.exists(l):
                fg.addVariable(l, Pose3)
        fg.addFactor([a, b], Pose3Pose3(MvNormal(mu, cov)))
    return fg


def loadG2o(path, N=100, prior_sigma=(0.1, 0.1, 0.05), max_edges=None):
    """Manhattan-style batch graph: :x0 + PriorPose2(N(0, diag(σ²))) then every g2o line
    (examples/ManhattanDatasetBatch.jl:28-37)."""
    fg = initfg(N)
    fg.addVariable("x0", Pose2)
    fg.addFactor(["x0"], PriorPose2(MvNormal(np.zeros(3), np.diag(np.square(prior_sigma)))))
    n_edges = 0
    for ins in importG2o(path):
        is_edge = ins[0].startswith("EDGE")
        if max_edges is not None and is_edge and n_edges >= max_edges:
            break
        parseG2oInstruction(fg, ins)
        n_edges += int(is_edge)     # max_edges counts EDGE_* records only (VERTEX_* lines are not edges)
    return fg


def stringG2oEdgeSE2(i, j, mu, info):
    u = [info[0, 0], info[0, 1], info[0, 2], info[1, 1], info[1, 2], info[2, 2]]
    return "EDGE_SE2 %d %d %.6f %.6f %.6f " % (i, j, mu[0], mu[1], mu[2]) + " ".join("%.6f" % x for x in u)


# ------------------------------------------------------------------------------------------ generators
def se2_compose(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([a[0] + c * b[0] - s * b[1], a[1] + s * b[0] + c * b[1], a[2] + b[2]])


def se2_between(a, b):
    """a⁻¹ ∘ b"""
    c, s = np.cos(a[2]), np.sin(a[2])
    dx, dy = b[0] - a[0], b[1] - a[1]
    th = b[2] - a[2]
    return np.array([c * dx + s * dy, -s * dx + c * dy, np.arctan2(np.sin(th), np.cos(th))])


def synth_manhattan_edges(P=3500, loops=1954, seed=0x524F4D45, grid=24):
    """Deterministic g2o-shaped stand-in for examples/manhattan.g2o: unit steps
    on an integer grid with turns in {0, ±π/2}, `loops` closures between poses ≤ 1 cell apart,
    information ≈ the file's (odometry diag ≈ (44.6, 399, 9591) with Λ12 ≠ 0; closures ≈ (175, 416, 1504)).
    -> (edges [(i, j, μ(3), Λ(3,3))...], ground-truth poses (P,3))"""
    rng = np.random.default_rng(seed)
    gt = np.zeros((P, 3))
    pos = np.array([0, 0]); hd = 0  # heading index 0..3
    dirs = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]])
    cells = {}
    for k in range(P):
        gt[k] = [pos[0], pos[1], hd * np.pi / 2]
        cells.setdefault((int(pos[0]), int(pos[1])), []).append(k)
        if k == P - 1:
            break
        turn = rng.choice([0, 1, -1], p=[0.7, 0.15, 0.15])
        nh = (hd + turn) % 4
        nxt = pos + dirs[nh]
        tries = 0
        while (abs(nxt[0]) > grid or abs(nxt[1]) > grid) and tries < 8:
            turn = rng.choice([1, -1]); nh = (hd + turn) % 4; nxt = pos + dirs[nh]; tries += 1
        if abs(nxt[0]) > grid or abs(nxt[1]) > grid:
            nh = (hd + 2) % 4; nxt = pos + dirs[nh]
        # the pose first turns in place then steps: relative motion = R(turn) then 1 forward
        hd = nh; pos = nxt
    gt[:, 2] = np.arctan2(np.sin(gt[:, 2]), np.cos(gt[:, 2]))

    def noisy(i, j, diag, off):
        rel = se2_between(gt[i], gt[j])
        d = diag * rng.uniform(0.9, 1.1, 3)
        info = np.diag(d)
        info[0, 1] = info[1, 0] = off * rng.uniform(-1, 1) * np.sqrt(d[0] * d[1])
        cov = np.linalg.inv(info)
        mu = rel + np.linalg.cholesky(cov) @ rng.standard_normal(3)
        return (i, j, mu, info)

    edges = [noisy(k, k + 1, np.array([44.6, 399.0, 9591.0]), 0.12) for k in range(P - 1)]
    cand = []
    for (cx, cy), ks in cells.items():
        near = list(ks)
        for dxy in ((1, 0), (0, 1)):
            near += cells.get((cx + dxy[0], cy + dxy[1]), [])
        near = sorted(set(near))
        for a in ks:
            for b in near:
                if b - a >= 4:
                    cand.append((a, b))
    cand = sorted(set(cand))
    if len(cand) < loops:
        raise RuntimeError("synthetic Manhattan walk produced only %d loop-closure candidates" % len(cand))
    pick = rng.choice(len(cand), size=loops, replace=False)
    for idx in sorted(pick):
        a, b = cand[idx]
        edges.append(noisy(a, b, np.array([175.0, 416.0, 1504.0]), 0.1))
    return edges, gt


def synth_manhattan(P=3500, loops=1954, seed=0x524F4D45, N=100, prior_sigma=(0.1, 0.1, 0.05)):
    edges, gt = synth_manhattan_edges(P, loops, seed)
    fg = initfg(N)
    fg.addVariable("x0", Pose2)
    fg.addFactor(["x0"], PriorPose2(MvNormal(np.zeros(3), np.diag(np.square(prior_sigma)))))
    for i, j, mu, info in edges:
        cov = np.linalg.inv(info); cov = (cov + cov.T) / 2
        for l in ("x%d" % i, "x%d" % j):
            if not fg.exists(l):
                fg.addVariable(l, Pose2)
        fg.addFactor(["x%d" % i, "x%d" % j], Pose2Pose2(MvNormal(mu, cov)))
    fg.ground_truth = {"x%d" % k: gt[k] for k in range(P)}
    return fg


def generateGraph_Circle(poses=6, N=100, landmark=True, loopClosure=True, kappaOdo=1.0, biasTurn=0.0):
    """src/canonical/GenerateCircular.jl:31-94"""
    fg = initfg(N)
    fg.addVariable("x0", Pose2)
    fg.addFactor(["x0"], PriorPose2(MvNormal(np.zeros(3), 0.01 * np.eye(3))))
    for i in range(poses):
        fg.addVariable("x%d" % (i + 1), Pose2)
        pp = Pose2Pose2(MvNormal([10.0, 0.0, 2 * np.pi / poses + biasTurn], np.diag((kappaOdo * np.array([0.1, 0.1, 0.1])) ** 2)))
        fg.addFactor(["x%d" % i, "x%d" % (i + 1)], pp)
    if landmark:
        fg.addVariable("l1", Point2)
        fg.addFactor(["x0", "l1"], Pose2Point2BearingRange(Normal(0, 0.1), Normal(20.0, 1.0)))
        if loopClosure:
            fg.addFactor(["x%d" % poses, "l1"], Pose2Point2BearingRange(Normal(0, 0.1), Normal(20.0, 1.0)))
    return fg


def generateGraph_Hexagonal(N=100, **kw):
    """src/canonical/GenerateHexagonal.jl:27-42"""
    return generateGraph_Circle(6, N=N, **kw)


def synth_helix3d(P=10000, N=100, seed=0x524F4D45, radius=10.0, per_turn=20, pitch=1.0):
    """Synthetic SE(3) helix (BASELINE.json configs[4]): P Pose3 on a helix, Pose3Pose3 odometry with
    Σ = diag(0.1²x3, 0.01²x3) (test/testPose3.jl:35) + closures between adjacent turns every 5th pose."""
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(seed)
    fg = initfg(N)
    T = []
    for k in range(P):
        a = 2 * np.pi * k / per_turn
        t = np.array([radius * np.cos(a), radius * np.sin(a), pitch * k / per_turn])
        R = Rot.from_euler("z", a + np.pi / 2).as_matrix()
        T.append((t, R))
    cov = np.diag([0.1 ** 2] * 3 + [0.01 ** 2] * 3)
    Lc = np.linalg.cholesky(cov)
    fg.addVariable("x0", Pose3)
    fg.addFactor(["x0"], PriorPose3(MvNormal(np.concatenate([T[0][0], Rot.from_matrix(T[0][1]).as_rotvec()]), cov)))

    def rel(i, j):
        ti, Ri = T[i]; tj, Rj = T[j]
        return np.concatenate([Ri.T @ (tj - ti), Rot.from_matrix(Ri.T @ Rj).as_rotvec()])

    for k in range(1, P):
        fg.addVariable("x%d" % k, Pose3)
        fg.addFactor(["x%d" % (k - 1), "x%d" % k], Pose3Pose3(MvNormal(rel(k - 1, k) + Lc @ rng.standard_normal(6), cov)))
    for k in range(per_turn, P, 5):
        fg.addFactor(["x%d" % (k - per_turn), "x%d" % k], Pose3Pose3(MvNormal(rel(k - per_turn, k) + Lc @ rng.standard_normal(6), cov)))
    fg.ground_truth = {"x%d" % k: np.concatenate([T[k][0], Rot.from_matrix(T[k][1]).as_rotvec()]) for k in range(P)}
    return fg


def add_synthetic_landmarks(fg, poses, n_landmarks, rng, sigma_b=0.03, sigma_r=0.5, min_range=3.0):
    """Point2 landmarks l0.. scattered around the given pose estimates (label -> (x, y, θ)), each sighted by the 2-4 nearest
    poses at least `min_range` away with Pose2Point2BearingRange(Normal(b, σ_b), Normal(ρ, σ_ρ)) (σ as in
    src/canonical/GenerateHoneycomb.jl:69; sightings ≥ 6 σ_ρ away so that a sampled range stays positive).
    Used to put landmarks on pose-only datasets such as the reference's examples/MIT.g2o (BASELINE configs[2]).
    -> {landmark label: true position}"""
    if isinstance(rng, (int, np.integer)):
        rng = np.random.default_rng(int(rng))
    labels = [l for l in poses if l in fg.variables and fg.variables[l] is Pose2]
    gt = np.array([poses[l] for l in labels], dtype=float)
    P = len(labels)
    lm = gt[rng.choice(P, n_landmarks, replace=False), :2] + rng.uniform(-8, 8, (n_landmarks, 2))
    truth = {}
    for j in range(n_landmarks):
        d = np.hypot(gt[:, 0] - lm[j, 0], gt[:, 1] - lm[j, 1])
        d[d < min_range] = np.inf
        near = np.argsort(d)[:rng.integers(2, 5)]
        fg.addVariable("l%d" % j, Point2)
        truth["l%d" % j] = lm[j]
        for k in sorted(near):
            dx, dy = lm[j] - gt[k, :2]
            b = np.arctan2(dy, dx) - gt[k, 2] + sigma_b * rng.standard_normal()
            r = np.hypot(dx, dy) + sigma_r * rng.standard_normal()
            fg.addFactor([labels[k], "l%d" % j],
                         Pose2Point2BearingRange(Normal(np.arctan2(np.sin(b), np.cos(b)), sigma_b), Normal(max(r, 0.1), sigma_r)))
    return truth


def synth_mit_br(P=808, n_landmarks=120, N=100, seed=0x524F4D45, step=2.0):
    """BASELINE.json configs[2] stand-in (the shipped MIT.g2o has no landmarks): a P-pose planar
    random walk (steps ≈ 2 m, MIT.g2o odometry statistics σ ≈ (0.75, 0.61, 0.053) scaled down 5x so the walk
    stays informative) plus landmarks, each sighted by the 2-4 nearest poses with Pose2Point2BearingRange
    (σ_b = 0.03, σ_ρ = 0.5 as in src/canonical/GenerateHoneycomb.jl:69)."""
    rng = np.random.default_rng(seed)
    gt = np.zeros((P, 3))
    for k in range(1, P):
        dth = rng.choice([0.0, 0.0, 0.0, np.pi / 2, -np.pi / 2]) + 0.05 * rng.standard_normal()
        gt[k] = se2_compose(gt[k - 1], np.array([step, 0.0, dth]))
        if np.hypot(gt[k, 0], gt[k, 1]) > 60:   # steer back towards the origin
            gt[k, 2] = np.arctan2(-gt[k, 1], -gt[k, 0]) + 0.3 * rng.standard_normal()
    gt[:, 2] = np.arctan2(np.sin(gt[:, 2]), np.cos(gt[:, 2]))
    fg = initfg(N)
    fg.addVariable("x0", Pose2)
    fg.addFactor(["x0"], PriorPose2(MvNormal(gt[0], np.diag(np.square([0.1, 0.1, 0.05])))))
    cov = np.diag(np.square([0.15, 0.12, 0.0106]))
    Lc = np.linalg.cholesky(cov)
    for k in range(1, P):
        fg.addVariable("x%d" % k, Pose2)
        fg.addFactor(["x%d" % (k - 1), "x%d" % k], Pose2Pose2(MvNormal(se2_between(gt[k - 1], gt[k]) + Lc @ rng.standard_normal(3), cov)))
    truth = {"x%d" % k: gt[k] for k in range(P)}
    truth.update(add_synthetic_landmarks(fg, truth, n_landmarks, rng))
    fg.ground_truth = truth
    return fg


def dead_reckon_init_pose3(fg, seed=1, sigma=(0.1, 0.1, 0.1, 0.01, 0.01, 0.01)):
    """Pose3 counterpart of dead_reckon_init: prior mean composed along the first incoming Pose3Pose3."""
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(seed)
    mean = {}
    for _, labels, f in fg.factors:
        if isinstance(f, PriorPose3):
            mean[labels[0]] = f.Z.mu.copy()
    for _, labels, f in fg.factors:
        if isinstance(f, Pose3Pose3) and labels[0] in mean and labels[1] not in mean:
            a = mean[labels[0]]
            Ra = Rot.from_rotvec(a[3:])
            mean[labels[1]] = np.concatenate([a[:3] + Ra.apply(f.Z.mu[:3]), (Ra * Rot.from_rotvec(f.Z.mu[3:])).as_rotvec()])
    for l, t in fg.variables.items():
        if t is Pose3:
            m = mean.get(l, np.zeros(6))
            fg.initVariable(l, m[:, None] + np.asarray(sigma)[:, None] * rng.standard_normal((6, fg.N)))
    return fg


# ------------------------------------------------------------------------------------------ packing
class PackedGraph:
    """Flat tables for the device sweep.  Variables are numbered per type in insertion order."""

    def __init__(self, fg):
        self.N = fg.N
        self.labels = {Pose2: [], Point2: [], Pose3: []}
        self.index = {}
        for l, t in fg.variables.items():
            self.index[l] = len(self.labels[t])
            self.labels[t].append(l)
        p2, br, p3, pr2, pr3, prpt = [], [], [], [], [], []
        nullh = getattr(fg, "nullhypo", {})
        for flabel, labels, f in fg.factors:
            ids = [self.index[l] for l in labels]
            if isinstance(f, Pose2Pose2): p2.append((ids, f, flabel, fg.multihypo.get(flabel)))
            elif isinstance(f, Pose2Point2BearingRange): br.append((ids, f, flabel, fg.multihypo.get(flabel)))
            elif isinstance(f, Pose3Pose3): p3.append((ids, f, flabel, None))
            elif isinstance(f, PriorPose2): pr2.append((ids, f, flabel))
            elif isinstance(f, PriorPose3): pr3.append((ids, f, flabel))
            elif isinstance(f, PriorPoint2): prpt.append((ids, f, flabel))
            else: raise TypeError("factor type %s is outside the hot path" % type(f).__name__)

        def rel_tables(items, d):
            F = len(items)
            mu = np.zeros((F, d)); cov = np.zeros((F, d, d))
            vfrom = np.zeros(F, dtype=np.int32); vto = np.zeros(F, dtype=np.int32)
            alt = np.full(F, -1, dtype=np.int32); w = np.ones(F); w2 = np.zeros(F); nh = np.zeros(F)
            for k, (ids, f, fl, mh) in enumerate(items):
                mu[k] = f.Z.mu; cov[k] = f.Z.cov; vfrom[k], vto[k] = ids[:2]
                nh[k] = nullh.get(fl, 0.0)      # IIF nullhypo= of the factor: every row of the factor carries it
                if mh is not None:   # multihypo over the second pose: var_to with probability w, else alt
                    alt[k] = ids[2]; w[k], w2[k] = mh
            return dict(F=F, mu=mu, cov=cov, var_from=vfrom, var_to=vto, alt=alt, w=w, w2=w2, nh=nh, labels=[it[2] for it in items])

        self.p2p2 = rel_tables(p2, 3)
        self.p3p3 = rel_tables(p3, 6)
        Fb = len(br)
        # dir 1 (landmark -> pose): one row per factor, `point` = primary landmark, `alt` = other candidate (-1), `w` = P(primary)
        # dir 0 (pose -> landmark): `rows0`, one row per (factor, candidate landmark)
        r0 = dict(factor=[], pose=[], point=[], alt=[], w=[])
        for k, (ids, f, _, mh) in enumerate(br):
            cands = [(ids[1], -1, 1.0)] if mh is None else [(ids[1], ids[2], mh[0]), (ids[2], ids[1], mh[1])]
            for pt, al, w in cands:
                r0["factor"].append(k); r0["pose"].append(ids[0]); r0["point"].append(pt); r0["alt"].append(al); r0["w"].append(w)
        self.br = dict(F=Fb, mu=np.array([[f.bearing.mu, f.range.mu] for _, f, _, _ in br]).reshape(Fb, 2),
                       sigma=np.array([[f.bearing.sigma, f.range.sigma] for _, f, _, _ in br]).reshape(Fb, 2),
                       pose=np.array([ids[0] for ids, _, _, _ in br], dtype=np.int32),
                       point=np.array([ids[1] for ids, _, _, _ in br], dtype=np.int32),
                       alt=np.array([(-1 if mh is None else ids[2]) for ids, _, _, mh in br], dtype=np.int32),
                       w=np.array([(1.0 if mh is None else mh[0]) for _, _, _, mh in br], dtype=np.float64),
                       nh=np.array([nullh.get(fl, 0.0) for _, _, fl, _ in br], dtype=np.float64),
                       rows0={k: np.asarray(v, dtype=(np.float64 if k == "w" else np.int32)) for k, v in r0.items()},
                       labels=[it[2] for it in br])

        def prior_tables(items, d):
            F = len(items)
            return dict(F=F, mu=np.array([f.Z.mu for _, f, _ in items]).reshape(F, d),
                        cov=np.array([f.Z.cov for _, f, _ in items]).reshape(F, d, d),
                        var=np.array([ids[0] for ids, _, _ in items], dtype=np.int32),
                        labels=[it[2] for it in items])

        self.prior2 = prior_tables(pr2, 3)
        self.prior3 = prior_tables(pr3, 6)
        self.priorpt2 = prior_tables(prpt, 2)   # landmark priors: parametric rows AND one proposal row each in the solve loop

    @classmethod
    def from_pose2_tables(cls, N, n_poses, mu, cov, var_from, var_to, prior_mu=None, prior_cov=None, prior_var=None):
        """Packed graph of a Pose2 / Pose2Pose2 / PriorPose2 problem straight from arrays (no per-factor Python
        objects): what a bulk loader hands over for the 2^16 / 2^20-factor scaling sets of SURVEY §8(d)-6."""
        self = cls.__new__(cls)
        self.N = int(N)
        self.labels = {Pose2: ["x%d" % k for k in range(n_poses)], Point2: [], Pose3: []}
        self.index = {}   # labels -> index is the identity here; not materialised for large graphs
        F = len(var_from)
        self.p2p2 = dict(F=F, mu=np.asarray(mu, dtype=np.float64).reshape(F, 3), cov=np.asarray(cov, dtype=np.float64).reshape(F, 3, 3),
                         var_from=np.asarray(var_from, dtype=np.int32), var_to=np.asarray(var_to, dtype=np.int32), labels=[])
        e3 = dict(F=0, mu=np.zeros((0, 6)), cov=np.zeros((0, 6, 6)), var_from=np.zeros(0, np.int32), var_to=np.zeros(0, np.int32), labels=[])
        self.p3p3 = e3
        z32 = np.zeros(0, np.int32)
        self.br = dict(F=0, mu=np.zeros((0, 2)), sigma=np.zeros((0, 2)), pose=z32, point=z32, alt=z32, w=np.zeros(0),
                       rows0=dict(factor=z32, pose=z32, point=z32, alt=z32, w=np.zeros(0)), labels=[])
        P = 0 if prior_var is None else len(prior_var)
        self.prior2 = dict(F=P, mu=np.asarray(prior_mu if P else np.zeros((0, 3)), dtype=np.float64).reshape(P, 3),
                           cov=np.asarray(prior_cov if P else np.zeros((0, 3, 3)), dtype=np.float64).reshape(P, 3, 3),
                           var=np.asarray(prior_var if P else np.zeros(0), dtype=np.int32), labels=[])
        self.prior3 = dict(F=0, mu=np.zeros((0, 6)), cov=np.zeros((0, 6, 6)), var=np.zeros(0, np.int32), labels=[])
        self.priorpt2 = dict(F=0, mu=np.zeros((0, 2)), cov=np.zeros((0, 2, 2)), var=np.zeros(0, np.int32), labels=[])
        return self

    @staticmethod
    def conv_table(tab):
        """Both directions of every relative factor, interleaved in factor order:
        conv 2f = (f, dir 0: fixed=from, target=to), conv 2f+1 = (f, dir 1: fixed=to, target=from)."""
        F = tab["F"]
        factor = np.repeat(np.arange(F, dtype=np.int32), 2)
        d = np.tile(np.array([0, 1], dtype=np.int32), F)
        fixed = np.empty(2 * F, dtype=np.int32); target = np.empty(2 * F, dtype=np.int32)
        fixed[0::2] = tab["var_from"]; target[0::2] = tab["var_to"]
        fixed[1::2] = tab["var_to"]; target[1::2] = tab["var_from"]
        return factor, d, fixed, target

    @staticmethod
    def conv_hypotheses(tab):
        """multihypo columns of a relative-pose table, or None when no factor carries hypotheses:
        (alt, w) for the 2F interleaved rows of conv_table -- both rows of a multihypo factor name the other candidate and the
        probability of their own -- and `extra`: one more row per such factor, (f, dir 0, fixed = from, target = the second candidate)
        with alt = the first candidate and w = its own probability (the second candidate receives a proposal too)."""
        mh = np.nonzero(tab["alt"] >= 0)[0].astype(np.int32)
        if mh.size == 0:
            return None
        alt = np.repeat(tab["alt"], 2).astype(np.int32); w = np.repeat(tab["w"], 2).astype(np.float64)
        extra = dict(factor=mh, dir=np.zeros(mh.size, np.int32), fixed=tab["var_from"][mh], target=tab["alt"][mh],
                     alt=tab["var_to"][mh], w=tab["w2"][mh].astype(np.float64))
        return alt, w, extra

    def beliefs(self, fg, vartype):
        """(V, dim, N) SoA blocks from fg.vals (all variables of the type must be initialised)."""
        ls = self.labels[vartype]
        out = np.zeros((len(ls), vartype.dim, self.N))
        for k, l in enumerate(ls):
            out[k] = fg.vals[l]
        return out


def synth_pose2_tables(n_factors, seed=0x524F4D45, loop_fraction=0.358, N=100):
    """Vectorised g2o-shaped Pose2Pose2 problem with `n_factors` edges in Manhattan proportions (SURVEY Appendix D:
    64 % odometry, 36 % closures with spans ~ median 64): -> (PackedGraph, initial beliefs [P,3,N]).  Geometry is a
    unit-step walk; the tables are what the scaling runs F ∈ {2^16, 2^20} sweep over."""
    rng = np.random.default_rng(seed)
    n_loops = int(round(n_factors * loop_fraction))
    P = n_factors - n_loops + 1
    turn = rng.choice([0.0, np.pi / 2, -np.pi / 2], size=P, p=[0.7, 0.15, 0.15]); turn[0] = 0.0
    th = np.cumsum(turn)
    pos = np.concatenate([[[0.0, 0.0]], np.cumsum(np.stack([np.cos(th[1:]), np.sin(th[1:])], 1), axis=0)])
    gt = np.column_stack([pos, np.arctan2(np.sin(th), np.cos(th))])
    i_od = np.arange(P - 1); j_od = i_od + 1
    span = np.minimum(np.maximum(4, rng.lognormal(np.log(64), 1.0, n_loops).astype(np.int64)), P - 1)
    i_lc = rng.integers(0, P - span); j_lc = i_lc + span
    vf = np.concatenate([i_od, i_lc]).astype(np.int32); vt = np.concatenate([j_od, j_lc]).astype(np.int32)
    c, s_ = np.cos(gt[vf, 2]), np.sin(gt[vf, 2])
    d = gt[vt, :2] - gt[vf, :2]
    rel = np.column_stack([c * d[:, 0] + s_ * d[:, 1], -s_ * d[:, 0] + c * d[:, 1], np.arctan2(np.sin(gt[vt, 2] - gt[vf, 2]), np.cos(gt[vt, 2] - gt[vf, 2]))])
    sig = np.where((np.arange(n_factors) < P - 1)[:, None], [[0.15, 0.05, 0.0102]], [[0.076, 0.049, 0.0258]])
    mu = rel + sig * rng.standard_normal((n_factors, 3))
    cov = np.zeros((n_factors, 3, 3)); cov[:, 0, 0] = sig[:, 0] ** 2; cov[:, 1, 1] = sig[:, 1] ** 2; cov[:, 2, 2] = sig[:, 2] ** 2
    cov[:, 0, 1] = cov[:, 1, 0] = 0.1 * sig[:, 0] * sig[:, 1]
    pk = PackedGraph.from_pose2_tables(N, P, mu, cov, vf, vt, prior_mu=[[0, 0, 0]], prior_cov=[np.diag([0.01, 0.01, 0.0025])], prior_var=[0])
    bel = gt[:, :, None] + np.array([0.1, 0.1, 0.05])[None, :, None] * rng.standard_normal((P, 3, N))
    return pk, bel


def dead_reckon_init(fg, seed=1, sigma=(0.1, 0.1, 0.05)):
    """Initialise every Pose2 belief from the prior mean composed along the first incoming odometry
    factor, plus N(0, diag σ²) per particle (bench/test input preparation)."""
    rng = np.random.default_rng(seed)
    mean = {}
    for _, labels, f in fg.factors:
        if isinstance(f, PriorPose2):
            mean[labels[0]] = f.Z.mu.copy()
    if not mean:
        first = next(l for l, t in fg.variables.items() if t is Pose2)
        mean[first] = np.zeros(3)
    pending = [(labels, f) for _, labels, f in fg.factors if isinstance(f, Pose2Pose2)]
    progress = True
    while pending and progress:
        progress = False
        rest = []
        for labels, f in pending:
            a, b = labels[:2]   # (a multihypo factor dead-reckons through its first candidate)
            if a in mean and b not in mean:
                mean[b] = se2_compose(mean[a], f.Z.mu); progress = True
            elif b in mean and a not in mean:
                inv = se2_between(f.Z.mu, np.zeros(3))
                mean[a] = se2_compose(mean[b], inv); progress = True
            elif a not in mean:
                rest.append((labels, f))
        pending = rest
    for l, t in fg.variables.items():
        if t is Pose2:
            m = mean.get(l, np.zeros(3))
            pts = m[:, None] + np.asarray(sigma)[:, None] * rng.standard_normal((3, fg.N))
            fg.initVariable(l, pts)
    # landmarks: first sighting, particle by particle (pose particle ∘ measurement mean)
    for _, labels, f in fg.factors:
        if isinstance(f, Pose2Point2BearingRange) and not fg.isInitialized(labels[1]) and fg.isInitialized(labels[0]):
            p = fg.getVal(labels[0])
            a = p[2] + f.bearing.mu
            fg.initVariable(labels[1], np.stack([p[0] + f.range.mu * np.cos(a), p[1] + f.range.mu * np.sin(a)]))
    return fg
