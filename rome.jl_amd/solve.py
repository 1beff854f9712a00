"""Convenience drivers around the hot path (host control flow only; every computation is a device launch).

`initAll` is the counterpart of IIF `initAll!` / `doautoinit!` (graphinit): variables are initialised in graph order by
convolving a factor whose other variables already have beliefs (priors first).  `solveGraph` strings the pieces a `solveTree!`
user calls implicitly: initialisation, (optionally) the parametric solve as a starting point, the device-resident non-parametric
sweeps of DeviceGraph.solve -- the per-variable operations of the reference (convolutions, manikde! bandwidths, manifoldProduct)
on a whole-graph Jacobi schedule instead of the Bayes tree --, download and point estimates."""
import numpy as np

from .convolution import approxConv
from .factors import _PriorFactor


def initAll(fg, seed=1, solver=None):
    """Initialise every variable that can be reached: priors are sampled, then each uninitialised variable takes the proposal of
    the FIRST factor (in insertion order) that links it to initialised variables -- IIF multiplies the proposals of all such
    factors (`predictbelief`); with one odometry chain per variable, the usual case at graph-init time, the two coincide.
    Returns the labels that are still uninitialised (disconnected from every prior)."""
    kw = {} if solver is None else {"solver": solver}
    k = 0
    changed = True
    while changed:
        changed = False
        for flabel, labels, f in fg.factors:
            if isinstance(f, _PriorFactor):
                if not fg.isInitialized(labels[0]):
                    fg.initVariable(labels[0], approxConv(fg, flabel, labels[0], seed=seed + k, **kw)); k += 1; changed = True
                continue
            mh = fg.multihypo.get(flabel)
            cand = labels[:2] if mh is None else labels[:1]       # multihypo factors only initialise their pose
            for t in cand:
                others = [l for l in labels if l != t]
                if not fg.isInitialized(t) and all(fg.isInitialized(o) for o in (others if mh is None else others[:1])):
                    if mh is not None and not all(fg.isInitialized(o) for o in others):
                        continue
                    fg.initVariable(t, approxConv(fg, flabel, t, seed=seed + k, **kw)); k += 1; changed = True
    return [l for l in fg.ls() if not fg.isInitialized(l)]


def initAllOrdered(fg, seed=1, ctx=None, sweeps=0, kind="colour", solver=None):
    """IIF `initAll!` on the device, with IIF's semantics: every variable WITHOUT a belief gets the `manifoldProduct` of the proposals of
    ALL factors whose other variables already have one (`doautoinit!` -> `predictbelief`), in rounds outward from the priors and from
    the variables that are initialised already (`schedule.OrderedSolve`: beliefs resident in a `DeviceStore`, one up-solve plan per
    independent group); then `sweeps` ordered Gauss-Seidel sweeps over the whole graph.  Beliefs are written back to `fg`.
    -> the OrderedSolve (its store keeps the beliefs on the device for further sweeps)."""
    from .api import make_opts
    from .clique import DeviceStore
    from .schedule import OrderedSolve
    from .graph import gc_paused
    keep = [l for l in fg.variables if fg.isInitialized(l)]
    store = DeviceStore(fg, ctx=ctx)
    kw = {} if solver is None else {"solver": solver}
    with gc_paused():     # the rounds' plans: live objects only
        osv = OrderedSolve(store, kind=kind, keep=keep)
        osv.init(make_opts(N=fg.N, seed=seed, **kw))
    if sweeps:
        osv.sweep(make_opts(N=fg.N, seed=seed + 1, **kw), sweeps)
    store.download(fg)
    return osv


def solveGraph(fg, n_sweeps=10, seed=0x524F4D45, init="graph", bandwidth="silverman", frozen=None, opts=None, product="importance",
               gibbs_iters=1):
    """initialise -> device sweeps -> download -> setPPE.  init: "graph" (initAll for whatever has no belief yet), "parametric"
    (beliefs around solveGraphParametric's solution, like IIF's initParametricFrom!), "ordered" (initAllOrdered: IIF's initAll! order with
    the product of ALL usable factors per variable, device-resident) or None (beliefs must exist).
    product: "gibbs" = the reference's `manifoldProduct` (multiscale Gibbs product on `manikde!` bandwidths) per variable,
    "importance" = the round-1 stand-in; gibbs_iters = AMP's `Niter` (1 at the reference's default; the relative weight of the modes of
    a multimodal product needs ~3, tests/test_gpu_gibbs.py).  Returns the DeviceGraph (beliefs stay resident for further sweeps)."""
    from .api import make_opts
    from .canonical import setPPE
    from .device import DeviceGraph
    if init not in ("graph", "parametric", "ordered", None):
        raise ValueError("init must be 'graph', 'parametric', 'ordered' or None")
    if init == "ordered":
        initAllOrdered(fg, seed=seed & 0xFFFF)
    elif init is not None:
        left = initAll(fg, seed=seed & 0xFFFF)
        if left:
            raise ValueError("variables without a path to a prior: %s" % left[:5])
    dg = DeviceGraph(fg)
    if init == "parametric":
        from .parametric import solveGraphParametric
        dg.init_from_means(solveGraphParametric(fg))
    else:
        dg.upload_beliefs(fg)
    if frozen:
        dg.set_frozen(frozen)
    dg.solve(opts if opts is not None else make_opts(N=fg.N, seed=seed), n_sweeps=n_sweeps, bandwidth=bandwidth, product=product,
             gibbs_iters=gibbs_iters)
    dg.download_beliefs(fg)
    setPPE(fg)
    return dg


def solveTree(fg, tree=None, messages="auto", passes=1, seed=0x524F4D45, ctx=None, order="mmd", **kw):
    """IIF `solveTree!(fg [, tree])` (examples/Hexagonal2D_SLAM.jl:24, examples/ManhattanDatasetBatch.jl:43, the incremental re-solves of
    examples/ManhattanDatasetIncremental.jl:107): `initAll!` for whatever has no belief yet (`initAllOrdered`), Bayes tree (built here, or the
    `tree.TreeSolver` of a previous call to re-solve from the current beliefs: plans are reused), up pass + down pass on the device,
    beliefs written back, PPEs set.  messages: "auto" (default) = "elimination" where it applies, else "marginal"; "marginal" = IIF's
    per-variable separator beliefs (the reference's semantics; hexagon windows, landmarks, multihypo, Pose3), "relative" (tree.py: relative
    messages between the separators of a clique), or "elimination" (elimination.py; Pose2 graphs of Pose2Pose2 / PriorPose2 factors:
    variable elimination in relative-factor algebra -- the form that SOLVES a large single-prior pose graph: Manhattan-3500 to 0.4 - 1.2 m
    of the MAP in one pass from the factors alone, no init pass).
    -> the TreeSolver (its store keeps the beliefs on the device; pass it back as `tree=` after adding nothing to the graph)."""
    from .api import make_opts
    from .canonical import setPPE
    from .tree import TreeSolver
    from .tree import BayesTree
    sig = (tuple(fg.variables), tuple(fl for fl, _, _ in fg.factors))
    if messages == "auto":    # a pose graph the elimination form covers -> "elimination"; anything else -> IIF's own message form
        from .elimination import RelativeEliminationSolver
        messages = getattr(tree, "messages", None) or ("elimination" if RelativeEliminationSolver.covers(fg) else "marginal")
    if messages == "elimination" or getattr(tree, "messages", None) == "elimination":
        # variable elimination in relative-factor algebra (elimination.py): no init pass, no starting beliefs -- the factors alone
        from .elimination import RelativeEliminationSolver
        es = tree if isinstance(tree, RelativeEliminationSolver) and tree.fg is fg and getattr(tree, "graph_signature", None) == sig else None
        if es is None:
            es = RelativeEliminationSolver(fg, ctx=ctx, **kw)
        es.graph_signature = sig
        es.solve(make_opts(N=fg.N, seed=seed), passes=passes)
        es.download(fg)
        setPPE(fg)
        return es
    if any(not fg.isInitialized(l) for l in fg.variables):
        initAllOrdered(fg, seed=seed & 0xFFFF, ctx=ctx)
    if isinstance(tree, TreeSolver):
        ts = tree
        if ts.fg is not fg:
            raise ValueError("solveTree: the TreeSolver passed as `tree` belongs to another graph")
        if getattr(ts, "graph_signature", sig) != sig:      # variables / factors were added since: the plans cover a stale tree
            ts = TreeSolver(fg, messages=ts.messages, ctx=ctx, order=order, **kw)
    else:   # (a BayesTree -- what the reference's solveTree!(fg, tree) takes -- is solved as given; None: built here)
        ts = TreeSolver(fg, tree=tree if isinstance(tree, BayesTree) else None, messages=messages, ctx=ctx, order=order, **kw)
    ts.graph_signature = sig
    ts.upload(fg)
    ts.solve(make_opts(N=fg.N, seed=seed), passes=passes)
    ts.download(fg)
    setPPE(fg)
    return ts
