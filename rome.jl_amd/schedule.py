"""Ordered up-solve schedules over the device-resident clique entry (SURVEY §8(f)-4 "Gibbs sweep scheduler").

What `initAll!` -> `solveTree!` do for a user of the reference (src/services/AdditionalUtils.jl:18-19, examples/ManhattanDatasetBatch.jl:43)
is an ORDERED visit of the variables: IIF `initAll!` gives every variable the product of the proposals of the factors whose other
variables already have beliefs, outward from the priors; the tree solve then visits cliques leaves-to-root and back.  This module
provides the orderings and drives them through `UpsolvePlan`s over a `DeviceStore` -- beliefs stay in HBM, a step is kernel launches
only, every group of a step is an independent set of variables (single-frontal cliques: one `rome_upsolve_plan` per group):

  * `init_rounds`       IIF's initialisation order: round r = the variables that have a factor whose other variables are all initialised
  * `greedy_colouring`  colour classes of the variable graph (a class is an independent set: one launch covers the whole class)
  * `OrderedSolve.init` the initialisation pass: level by level outward, a variable's product takes the factors whose other variables
                        are ALREADY initialised (earlier levels / earlier groups) -- beliefs never come from dead reckoning
  * `OrderedSolve.sweep` Gauss-Seidel over the whole graph: colour class after colour class ("colour"), or the level groups outward and
                        back ("levels"); every update sees the beliefs the previous groups of the same sweep wrote.

No Bayes tree is built (IIF's: elimination order, clique formation and the cavity messages between cliques stay there): a sweep
multiplies BELIEFS of neighbours, not messages, so repeated sweeps over-count evidence (DESIGN.md §11 has the measured trace)."""


def adjacency(fg):
    nb = {l: set() for l in fg.variables}
    for _, labels, _ in fg.factors:
        for a in labels:
            nb[a].update(b for b in labels if b != a)
    return nb


def greedy_colouring(fg, labels=None, nb=None):
    """colour classes (lists of labels, graph order inside a class) of the variables `labels` (default all): no factor links two
    members of a class"""
    nb = nb or adjacency(fg)
    labels = list(fg.variables) if labels is None else list(labels)
    col = {}
    for l in labels:
        used = {col[o] for o in nb[l] if o in col}
        c = 0
        while c in used:
            c += 1
        col[l] = c
    classes = [[] for _ in range(max(col.values()) + 1)] if col else []
    for l in labels:
        classes[col[l]].append(l)
    return classes


def init_rounds(fg, initialised=()):
    """IIF `initAll!` (initSolvableAll! / doautoinit!) as rounds: round r = the variables that have, before the round, at least one
    factor whose OTHER variables are all initialised (a prior has none, so the prior-carrying variables form round 0); repeated until
    nothing more can be initialised.  On a pose chain this is the hop distance from the priors.  `initialised`: variables that already
    have a belief -- like IIF, the pass leaves them alone and starts from them.  -> (rounds, unreachable labels)"""
    by_var = {l: [] for l in fg.variables}
    mh = getattr(fg, "multihypo", {})
    for flabel, labels, _ in fg.factors:
        for l in labels:
            others = [o for o in labels if o != l]
            if flabel in mh and l != labels[0]:
                # a candidate of a multihypo=[1, w, 1-w] factor needs the CERTAIN variable only: particles drawn for the other candidate
                # keep their value (IIF initialises a fractional variable from the certain one) -- two candidates that are reachable
                # through such sightings alone must not wait for each other
                others = [labels[0]]
            by_var[l].append(others)
    done, rounds = set(initialised), []
    pending = [l for l in fg.variables if l not in done]
    while pending:
        cand = [l for l in pending if any(all(o in done for o in others) for others in by_var[l])]
        if not cand:
            break
        rounds.append(cand)
        done.update(cand)
        pending = [l for l in pending if l not in done]
    return rounds, pending


class OrderedSolve:
    """init pass + ordered Gauss-Seidel sweeps of a whole graph, device-resident.
    store: DeviceStore (or a stand-in with the same interface: the CPU tests inject oracle-backed ones through plan_cls).
    kind:  "colour" (sweep = the colour classes in order) or "levels" (sweep = the groups of the init rounds outward and back)."""

    def __init__(self, store, kind="colour", gibbsIters=1, Niter=1, plan_cls=None, keep=()):
        """keep: labels whose beliefs in the store are to be KEPT by the init pass (IIF initAll! only touches uninitialised variables).
        kind="colour" sweeps update them like every other variable; kind="levels" sweeps visit the groups of the init pass only, so
        the kept variables stay as they are there too."""
        if kind not in ("colour", "levels"):
            raise ValueError("kind must be 'colour' or 'levels'")
        if not 1 <= int(gibbsIters) <= 16:      # Philox stream of run k = k << 36: (iteration << 32) + family + row must stay below it
            raise ValueError("gibbsIters must be in 1..16")
        if plan_cls is None:
            from .clique import UpsolvePlan
            plan_cls = UpsolvePlan
        self.store, self.kind = store, kind
        fg = store.fg
        nb = adjacency(fg)
        self.levels, self.unreachable = init_rounds(fg, keep)
        if self.unreachable:
            raise ValueError("variables without a path to a prior: %s" % self.unreachable[:5])
        # ---- init pass: level by level, each level split into independent sets; usable = what has been initialised before the group
        self.init_groups, self.init_plans = [], []
        done = set(keep)
        for lv in self.levels:
            for grp in greedy_colouring(fg, lv, nb):
                snapshot = frozenset(done)
                self.init_groups.append(grp)
                self.init_plans.append(plan_cls(store, [[l] for l in grp], gibbsIters=gibbsIters, Niter=Niter, usable=snapshot.__contains__))
                done.update(grp)
        # ---- sweeps: every factor is usable.  The groups are fixed here; their plans are built on the first sweep (an init pass that is
        #      followed by a tree solve never sweeps: 0.1 s of host time on Manhattan-3500)
        if kind == "colour":
            self.sweep_groups = greedy_colouring(fg, None, nb)
        else:
            self.sweep_groups = self.init_groups + self.init_groups[-2::-1]     # outward, then back to the roots
        self._sweep_plans = None
        self._plan_args = (plan_cls, gibbsIters, Niter)
        self.runs = 0

    @property
    def sweep_plans(self):
        if self._sweep_plans is None:
            plan_cls, gibbsIters, Niter = self._plan_args
            everything = frozenset(self.store.fg.variables)
            cache = {}
            self._sweep_plans = []
            for grp in self.sweep_groups:
                key = tuple(grp)
                if key not in cache:
                    cache[key] = plan_cls(self.store, [[l] for l in grp], gibbsIters=gibbsIters, Niter=Niter, usable=everything.__contains__)
                self._sweep_plans.append(cache[key])
        return self._sweep_plans

    def _run(self, plans, opts):
        for p in plans:
            o = type(opts).from_buffer_copy(opts)
            o.stream_offset = opts.stream_offset + (self.runs << 36)     # (it << 32) + family / product offsets stay below 2^36
            p.run(o)
            self.runs += 1

    def init(self, opts):
        """IIF initAll!-style: every variable <- product of the proposals of the factors whose other variables are already initialised"""
        self._run(self.init_plans, opts)

    def sweep(self, opts, n=1):
        for _ in range(n):
            self._run(self.sweep_plans, opts)

    def stats(self):
        return dict(levels=len(self.levels), init_steps=len(self.init_plans), sweep_steps=len(self.sweep_groups),
                    largest_group=max(len(g) for g in self.sweep_groups), colours=len(self.sweep_groups) if self.kind == "colour" else None)
