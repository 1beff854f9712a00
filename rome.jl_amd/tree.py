"""Bayes tree of a factor graph and the tree-shaped solve over it: `solveTree!` (examples/ManhattanDatasetBatch.jl:43,
src/services/AdditionalUtils.jl:18-19) as a minimal HOST driver over the device-resident up-solve plans.

    variable ordering -> symbolic elimination -> cliques (frontals | separators) -> levels
    up pass, level by level from the leaves:   every clique of the level solves its own sub-graph (its factors + the messages of its
        children) and hands a message over its separators to its parent;
    down pass, level by level from the root:   every clique re-solves its frontals with the separators fixed at the posteriors the
        levels above have written (IIF solveDown: same inner path, one sweep).

Two message forms (`TreeSolver(messages=...)`):
  "marginal"  IIF's: the clique solves its frontals AND its own copies of its separators from their current beliefs (upGibbsCliqueDensity
              on the sub-graph: gibbsIters x products of factor proposals and child messages); the message is the belief of every
              separator copy (IIF TreeBelief: N points + manikde! bandwidths per separator VARIABLE).  A pose graph with one prior has
              no absolute information below the prior's clique: such messages only restate the beliefs the init pass left, and the
              solve stays where `initAll!` put it (measured: DESIGN.md, scripts/tree_surrogate.py).
  "relative"  (opt-in) the message keeps what a prior-free sub-tree DOES know -- where its separators are relative to each other: the clique
              conditions on its ANCHOR separator (the first Pose2 one) being exactly at its current mean, solves outward from it ONCE
              (every variable takes the product of the proposals from already-solved neighbours; no belief is used before the clique's
              own potential has informed it), and sends, per other separator s, the N samples of anchor^-1 * s -- a Pose2Pose2 /
              bearing-range factor with a SAMPLED measurement distribution between anchor and s in the parent's sub-graph (IIF accepts
              any SamplableBelief as Z) -- plus, when the sub-tree holds a prior, the marginal of the anchor alone (solved outward from
              the priors).  The joint over the separators is approximated by p(anchor) * prod_s p(s | anchor).

What is NOT here, on purpose: IIF's clique state machine, task-per-clique concurrency, recycling / incremental updates, fixed-lag
marginalisation, `multiproc`.  A level (all cliques of equal height: mutually independent given their children) is ONE
`rome_upsolve_plan` -- every launch covers every clique of the level, and a level is what `distributed.FrontierShard` deals to the
ranks.  The tree is built on the host once per graph; a solve is plan runs only (beliefs never leave HBM).
"""
import heapq

import numpy as np


class Clique:
    __slots__ = ("id", "frontals", "separators", "parent", "children", "factors", "level")

    def __init__(self, cid, frontals, separators, parent):
        self.id, self.frontals, self.separators, self.parent = cid, list(frontals), list(separators), parent
        self.children, self.factors, self.level = [], [], 0

    def __repr__(self):
        return "Clique(%d: %s | %s)" % (self.id, ",".join(map(str, self.frontals)), ",".join(map(str, self.separators)))


def min_degree_order(labels, nb, last=()):
    """greedy minimum-degree elimination order on the variable graph (ties: graph order); `last`: variables constrained to the end of
    the order (IIF passes such constraints to CCOLAMD), i.e. into the root."""
    pos = {l: k for k, l in enumerate(labels)}
    last = set(last)
    adj = {l: set(nb[l]) for l in labels}
    heap = [((l in last), len(adj[l]), pos[l], l) for l in labels]
    heapq.heapify(heap)
    done, order = set(), []
    while heap:
        pen, deg, _, v = heapq.heappop(heap)
        if v in done or deg != len(adj[v]):
            continue                              # stale entry
        done.add(v); order.append(v)
        S = adj.pop(v)
        for u in S:
            a = adj[u]
            a.discard(v)
            a |= S
            a.discard(u)
        for u in S:
            heapq.heappush(heap, ((u in last), len(adj[u]), pos[u], u))
    return order


class BayesTree:
    """cliques[c].frontals / .separators / .parent / .children / .factors (ids as given) / .level; levels[h] = cliques of height h
    (leaves = 0); order = the elimination order; clique_of[v] = the clique where v is frontal."""

    def __init__(self):
        self.cliques, self.levels, self.order, self.clique_of = [], [], [], {}

    @classmethod
    def build(cls, labels, factors, order="mmd", last=()):
        """labels: variables in graph order; factors: [(id, (variables...))]; order: "mmd" (minimum degree), "natural" (graph
        order) or an explicit list."""
        labels = list(labels)
        nb = {l: set() for l in labels}
        for _, vs in factors:
            for a in vs:
                nb[a].update(b for b in vs if b != a)
        if isinstance(order, str):
            if order == "mmd":
                order = min_degree_order(labels, nb, last)
            elif order == "natural":
                order = [l for l in labels if l not in set(last)] + [l for l in labels if l in set(last)]
            else:
                raise ValueError("order must be 'mmd', 'natural' or a list of variables")
        order = list(order)
        if sorted(map(str, order)) != sorted(map(str, labels)):
            raise ValueError("the elimination order must be a permutation of the variables")
        pos = {v: k for k, v in enumerate(order)}
        # ---- symbolic elimination: separator of v = its neighbours in the elimination graph at the time it is eliminated
        adj = {l: set(nb[l]) for l in labels}
        sep = {}
        for v in order:
            S = adj.pop(v)
            sep[v] = S
            for u in S:
                a = adj[u]
                a.discard(v); a |= S; a.discard(u)
        # ---- Bayes net -> Bayes tree (reverse elimination order; Kaess et al., "The Bayes tree", alg. 2)
        t = cls()
        t.order = order
        members = []                                # per clique: set(frontals + separators)
        for v in reversed(order):
            S = sep[v]
            if S:
                p = t.clique_of[min(S, key=pos.__getitem__)]
                if len(S) == len(members[p]) and S == members[p]:
                    t.cliques[p].frontals.insert(0, v); members[p].add(v); t.clique_of[v] = p
                    continue
            else:
                p = -1
            c = Clique(len(t.cliques), [v], sorted(S, key=pos.__getitem__), p)
            t.cliques.append(c); members.append(set(S) | {v}); t.clique_of[v] = c.id
            if p >= 0:
                t.cliques[p].children.append(c.id)
        # ---- factors: to the clique whose frontals hold the factor's first-eliminated variable
        for fid, vs in factors:
            t.cliques[t.clique_of[min(vs, key=pos.__getitem__)]].factors.append(fid)
        # ---- levels by height (children before parents: cliques were created parents first)
        for c in reversed(t.cliques):
            c.level = 1 + max((t.cliques[k].level for k in c.children), default=-1)
        t.levels = [[] for _ in range(1 + max((c.level for c in t.cliques), default=-1))]
        for c in t.cliques:
            t.levels[c.level].append(c.id)
        return t

    def summary(self):
        fr = np.array([len(c.frontals) for c in self.cliques]); sp = np.array([len(c.separators) for c in self.cliques])
        w = np.array([len(l) for l in self.levels])
        return ("BayesTree: %d cliques, %d levels; frontals/clique max %d mean %.2f; separators max %d mean %.2f; clique size max %d; "
                "level width max %d median %d, %d levels of width 1"
                % (len(self.cliques), len(self.levels), fr.max(), fr.mean(), sp.max(), sp.mean(), (fr + sp).max(), w.max(), int(np.median(w)),
                   int((w == 1).sum())))


# ------------------------------------------------------------------------------------------------------------------ the solve
ZERO = "0~"          # a Pose2 block that stays zero: the samples of an identity row (tree.TreeSolver max_product)


def _colour(order, nbr):
    """greedy colouring of `order` (list) under adjacency nbr(v) -> iterable; -> {v: colour}"""
    col = {}
    for v in order:
        used = {col[o] for o in nbr(v) if o in col}
        c = 0
        while c in used:
            c += 1
        col[v] = c
    return col


def _outward(targets, informed, pairwise, unary):
    """The solve order of a clique's sub-graph outward from what is known: round r holds the targets that have, before the round, a
    unary source or a factor whose OTHER variables are all informed; a variable is solved once, from every such source.
    pairwise: [(factor id, variables)], unary: [(source id, variable)] -> ([[(variable, [factor ids], [source ids])] per round], unreached)"""
    informed = set(informed)
    by_var = {v: [] for v in targets}
    for fid, vs in pairwise:
        for v in vs:
            if v in by_var:
                by_var[v].append((fid, [o for o in vs if o != v]))
    un = {v: [] for v in targets}
    for sid, v in unary:
        if v in un:
            un[v].append(sid)
    pending, rounds = [v for v in targets if v not in informed], []
    while pending:
        now = []
        for v in pending:
            fs = [fid for fid, others in by_var[v] if all(o in informed for o in others)]
            if fs or un[v]:
                now.append((v, fs, un[v]))
        if not now:
            break
        rounds.append(now)
        informed.update(v for v, _, _ in now)
        pending = [v for v in pending if v not in informed]
    return rounds, pending


class LevelSpec:
    """One tree level as ONE up-solve description over lifted labels:
      fg        FactorGraph holding the level's variables and (relabelled) factors -- what CliqueBatch / the oracle restatement consume
      cliques   per clique of the level: (update labels, their group numbers)
      order / groups / owner   the level's update list: group g of every clique together, groups in order; owner[k] = clique (position
                in the level) of entry k
      pairs     (factor, destination) rows in update order
      smsgs     (source label, destination label): store-resident messages, in destination order
      copies / anchors   block operations BEFORE the run: (source, destination) copies; (belief, destination) anchors (N copies of the mean)
      relatives          AFTER the run: (anchor block, separator block, destination): samples of anchor^-1 * separator"""

    def __init__(self, fg, cliques, pairs_of, smsgs, gibbs_iters, copies=(), anchors=(), relatives=()):
        self.fg, self.cliques, self.gibbs_iters = fg, cliques, gibbs_iters
        self.copies, self.anchors, self.relatives = list(copies), list(anchors), list(relatives)
        self.order, self.groups, self.owner = [], [], []
        for g in sorted({g for _, gs in cliques for g in gs}):
            for k, (upd, gs) in enumerate(cliques):
                for l, gl in zip(upd, gs):
                    if gl == g:
                        self.order.append(l); self.groups.append(g); self.owner.append(k)
        self.pairs = [(fl, l) for l in self.order for fl in pairs_of.get(l, ())]
        by_dest = {}
        for src, dst in smsgs:
            by_dest.setdefault(dst, []).append(src)
        self.smsgs = [(src, l) for l in self.order for src in by_dest.get(l, ())]


class TreeSolver:
    """`solveTree!` over a BayesTree: builds, once, the lifted variable universe (home blocks + the cliques' private copies), one up and
    one down LevelSpec per tree level and their plans through `backend`; `solve(opts)` = plan runs and block operations only.

    backend: object with Store(universe_fg) -> store (`.index`, `.upload(fg)`, `.download(fg, labels)`), Plan(store, spec, share=None)
    -> `.run(opts)`, BlockOp(store, op, entries) -> `.run()` (op "copy" / "anchor" / "relative").  Default: the device
    (`DeviceBackend`); the CPU tests inject an oracle-backed one.
    messages: "marginal" (default: IIF's form -- the reference's semantics, and the more robust one on small or multimodal graphs: hexagon
    windows, beehive, the reference's Manhattan-500 graph) or "relative" (what moves a LARGE pose graph with a single prior off its init
    pass: Manhattan-3500 5.3 m -> metres; module docstring).  gibbsIters / downIters: iterations of the up / down clique solves in
    "marginal" form (IIF: 3 / 1).  The "relative" form solves every variable ONCE outward (up and down); rootIters / refineIters add
    Gibbs sweeps over the frontals of the root / of every clique in the down pass with ALL factors and messages of the clique (the
    outward solve takes a variable's proposals from the neighbours solved before it only); relIters the same inside the relative solve
    of the up pass (anchor fixed), before the samples of anchor^-1 * separator are taken.
    max_product ("relative" form): a Pose2 variable with more proposals than this takes its product in TWO stages -- partial products
    over chunks of at most max_product proposals (scratch blocks, all chunks of all variables of the step in one launch), then the
    product of the partial products (each enters through an identity row: a sampled-measurement Pose2Pose2 row whose samples are all
    zero).  The multiscale Gibbs product is ONE two-wave block per variable and its time grows with the square of the number of
    proposals (42 proposals: 14 ms with the rest of the chip idle; profiles/r05_tree_solve.txt); 0 = one product whatever the count."""

    def __init__(self, *a, **kw):
        from .graph import gc_paused
        with gc_paused():     # the level specs and plans are ~1e5 live objects: no cyclic garbage to find while they are built
            self._init(*a, **kw)

    def _init(self, fg, tree=None, order="mmd", last=(), messages="marginal", gibbsIters=3, downIters=1, rootIters=0, refineIters=0, relIters=0,
              max_product=8, backend=None, ctx=None, shard=None, message_tree="star"):
        """message_tree ("relative" form): the STRUCTURE of the message over a clique's separators -- a spanning tree T of relative
        messages, p(root) * prod_{(j,k) in T} p(s_k | s_j).  "star" (default): every separator tied to the ONE anchor.  "hop": Prim's
        tree from the anchor over shortest-path lengths in the clique-local graph (the clique's factors + the tree edges of its
        children's messages), host side; every internal node j of T is one more anchored outward solve of the clique (private copies,
        same launches; 1.7 per message on Manhattan-3500).  With EXACT pair marginals the tree halves the star's bias (linear-Gaussian
        surrogate, scripts/tree_linear_surrogate.py: 0.86 m against 2 - 3 m); with the one-shot outward solves of this class it does
        not (device, Manhattan-3500, after rigid alignment: 2.3 m against 1.3 m; profiles/r06_tree_forms.txt) -- what limits this
        solver is the clique solve and the belief-weighted down pass, not the message structure: elimination.py replaces both.
                shard: a factory `store -> distributed.FrontierShard` (the store exists only once the lifted universe is known): every level is
        then dealt to the ranks by clique -- share up-solve, ONE all-gather of the level's written blocks, one scatter; the block
        operations between levels run on every rank (each holds the whole store)."""
        from .graph import FactorGraph
        if messages not in ("relative", "marginal"):
            raise ValueError("messages must be 'relative' or 'marginal'")
        if message_tree not in ("hop", "star"):
            raise ValueError("message_tree must be 'hop' or 'star'")
        self.message_tree = message_tree
        self.fg, self.N, self.messages = fg, fg.N, messages
        self.tree = tree or BayesTree.build(list(fg.variables), [(fl, tuple(ls)) for fl, ls, _ in fg.factors], order=order, last=last)
        self.backend = backend or DeviceBackend(ctx)
        self.gibbsIters, self.downIters, self.rootIters, self.refineIters = int(gibbsIters), int(downIters), int(rootIters), int(refineIters)
        self.relIters, self.max_product = int(relIters), int(max_product or 0)
        if not 0 <= self.relIters <= 16:
            raise ValueError("relIters must be in 0..16")
        if not (1 <= self.gibbsIters <= 16 and 1 <= self.downIters <= 16 and 0 <= self.rootIters <= 16 and 0 <= self.refineIters <= 16):
            raise ValueError("gibbsIters / downIters must be in 1..16")       # Philox: run k draws from k << 36
        self.findex = {fl: (fl, ls, f) for fl, ls, f in fg.factors}
        U = FactorGraph(fg.N)
        for l, vt in fg.variables.items():
            U.addVariable(l, vt)
        self.universe = U
        if messages == "marginal":
            ups, downs = self._specs_marginal()
        else:
            ups, downs = self._specs_relative()
        self.up_specs, self.down_specs = ups, downs
        self.store = self.backend.Store(U)
        if ZERO in U.variables:
            self.store.put(ZERO, np.zeros((3, fg.N)))
        B = self.backend
        ops = lambda s: ([B.BlockOp(self.store, "copy", s.copies)] if s.copies else []) + ([B.BlockOp(self.store, "anchor", s.anchors)] if s.anchors else [])
        self.up_pre = [ops(s) for s in ups]
        self.shard = shard(self.store) if shard is not None else None
        if self.shard is not None:
            base = B
            B = _ShardedPlans(base, lambda s: self.shard.plan_level(s, base.Plan))
        self.up_plans = [B.Plan(self.store, s) if s.order else None for s in ups]
        self.up_post = [[B.BlockOp(self.store, "relative", s.relatives)] if s.relatives else [] for s in ups]
        self.down_plans = [B.Plan(self.store, s) if s.order else None for s in downs]
        self.root_plans = [B.Plan(self.store, s) if s is not None and s.order else None for s in getattr(self, "root_specs", [None] * len(ups))]
        self.refine_plans = [B.Plan(self.store, s) if s is not None and s.order else None for s in getattr(self, "refine_specs", [None] * len(ups))]
        self.rel_plans = [B.Plan(self.store, s) if s is not None and s.order else None for s in getattr(self, "rel_specs", [None] * len(ups))]
        self.runs = 0

    def _lift(self, L, fl, cid, tag, labels, factor):
        """the factor `fl` of clique `cid` over lifted labels, as a factor of the level graph L"""
        fg = self.fg
        nfl = "%s%s%d" % (fl, tag, cid)
        L.factors.append((nfl, labels, factor)); L._findex[nfl] = L.factors[-1]
        if fl in fg.multihypo:
            L.multihypo[nfl] = fg.multihypo[fl]
        if fl in getattr(fg, "nullhypo", {}):
            L.nullhypo[nfl] = fg.nullhypo[fl]
        return nfl

    def _split_products(self, L, cliques, pairs_of, smsgs):
        """staged products for Pose2 variables with more than max_product proposals (class docstring): a variable with K proposals is
        the root of a tree of partial products with fan-in <= max_product; a variable of update group g is solved in step
        g * (D + 1) + D, its partial products of depth d below it in step g * (D + 1) + D - d (D = the deepest tree of the level)"""
        from .factors import Pose2
        from .clique import SampledPose2Pose2
        G = self.max_product
        by_dest = {}
        for src, dst in smsgs:
            by_dest.setdefault(dst, []).append(src)
        count = lambda l: len(pairs_of.get(l, ())) + len(by_dest.get(l, ()))     # noqa: E731

        def depth(k):
            d = 0
            while k > G:
                k = -(-k // G); d += 1
            return d
        D = max((depth(count(l)) for upd, _ in cliques for l in upd if L.variables[l] is Pose2), default=0) if G > 1 else 0
        if D == 0:
            return cliques, pairs_of, smsgs
        out_cliques, out_smsgs = [], []
        for upd, grp in cliques:
            nu, ng = [], []

            def build(l, items, g, lvl):
                """make `l` the product of `items` ((kind, id): a factor row, a store message, or a partial-product label) in step
                g * (D + 1) + D - lvl"""
                if len(items) > G and L.variables[l] is Pose2:
                    nch = -(-len(items) // G)
                    parts = []
                    for k in range(nch):
                        pl = "%s^%d" % (l, k)
                        self._need(L, pl, Pose2)
                        build(pl, [(kind, x, l) for kind, x, _ in items[k::nch]], g, lvl + 1)
                        parts.append(("p", pl, l))
                    items = parts
                rows = []
                for kind, x, owner in items:
                    if kind == "f":          # a factor row of the ORIGINAL variable `owner`, retargeted to l
                        if owner == l:
                            rows.append(x)
                        else:
                            _, labels, f = L.getFactor(x)
                            nfl = "%s>%s" % (x, l)
                            L.factors.append((nfl, [l if o == owner else o for o in labels], f)); L._findex[nfl] = L.factors[-1]
                            if x in L.multihypo:
                                L.multihypo[nfl] = L.multihypo[x]
                            if x in L.nullhypo:
                                L.nullhypo[nfl] = L.nullhypo[x]
                            rows.append(nfl)
                    elif kind == "m":
                        out_smsgs.append((x, l))
                    else:                    # a partial product enters through an identity row (sampled row whose samples are zero)
                        idf = "=%s" % x
                        self._need(L, ZERO, Pose2)
                        L.factors.append((idf, [x, l], SampledPose2Pose2(ZERO))); L._findex[idf] = L.factors[-1]
                        rows.append(idf)
                pairs_of[l] = rows
                nu.append(l); ng.append(g * (D + 1) + D - lvl)
            for l, g in zip(upd, grp):
                build(l, [("f", fl, l) for fl in pairs_of.get(l, ())] + [("m", src, l) for src in by_dest.get(l, ())], g, 0)
            out_cliques.append((nu, ng))
        return out_cliques, pairs_of, out_smsgs

    @staticmethod
    def _message_tree(V, S, root, edges, can_anchor, star=False):
        """Spanning tree of relative messages over the separators S, rooted at `root`: Prim over shortest-path lengths in the
        clique-local graph (nodes V; edges [(u, v, length)]: the clique's own pairwise factors and the tree edges of its children's
        messages).  A node may only hang below one that can be an anchor (`can_anchor`: Pose2 -- the relative block operation takes
        its reference from a pose).  -> [(parent, child, length)] in attachment order; separators without a path are left out."""
        others = [s for s in S if s != root]
        if star:
            return [(root, s, 1.0) for s in others]
        adj = {v: [] for v in V}
        for a, b, w in edges:
            if a in adj and b in adj:
                adj[a].append((b, w)); adj[b].append((a, w))
        dist = {}
        for s0 in S:
            d = {s0: 0.0}; pq = [(0.0, 0, s0)]; n = 1
            while pq:
                d0, _, u = heapq.heappop(pq)
                if d0 > d[u]:
                    continue
                for v, w in adj[u]:
                    if d0 + w < d.get(v, np.inf):
                        d[v] = d0 + w; heapq.heappush(pq, (d0 + w, n, v)); n += 1
            dist[s0] = d
        intree, out = [root], []
        left = list(others)
        while left:
            best = None
            for j in intree:
                if not can_anchor(j):
                    continue
                for k in left:
                    w = dist[j].get(k)
                    if w is not None and (best is None or w < best[0]):
                        best = (w, j, k)
            if best is None:
                break
            out.append((best[1], best[2], best[0])); intree.append(best[2]); left.remove(best[2])
        return out

    def _need(self, L, label, vt):
        if label not in self.universe.variables:
            self.universe.addVariable(label, vt)
        if label not in L.variables:
            L.addVariable(label, vt)

    # ---------------------------------------------------------------- "marginal": IIF's per-variable separator beliefs
    def _specs_marginal(self):
        from .graph import FactorGraph
        fg, t = self.fg, self.tree
        ups, downs = [], []
        copy_label = lambda cid, s: "%s@%d" % (s, cid)
        for lvl in t.levels:
            for up in (True, False):
                L = FactorGraph(fg.N)
                cliques, pairs_of, smsgs, copies = [], {}, [], []
                for cid in lvl:
                    c = t.cliques[cid]
                    if not up and c.parent < 0:
                        cliques.append(([], [])); continue              # a root has no separators: its up-solve IS its posterior
                    lab = {v: v for v in c.frontals}
                    lab.update({s: (copy_label(cid, s) if up else s) for s in c.separators})
                    for v in c.frontals + c.separators:
                        self._need(L, lab[v], fg.variables[v])
                    upd = list(c.frontals) + (list(c.separators) if up else [])
                    nb = {v: set() for v in upd}
                    touching = {v: [] for v in upd}
                    for fl in c.factors:
                        _, ls, f = self.findex[fl]
                        nfl = self._lift(L, fl, cid, "@" if up else "!", [lab[v] for v in ls], f)
                        for v in ls:
                            if v in nb:
                                nb[v].update(o for o in ls if o != v and o in nb); touching[v].append(nfl)
                    col = _colour(upd, nb.__getitem__)                   # Gibbs order: colour classes of the clique's own factor graph
                    cliques.append(([lab[v] for v in upd], [col[v] for v in upd]))
                    for v in upd:
                        pairs_of[lab[v]] = touching[v]
                    if up:
                        copies += [(s, lab[s]) for s in c.separators]
                    for d in c.children:                                 # messages: the child's separator copies, written one level earlier
                        for s in t.cliques[d].separators:
                            if up or s in c.frontals:
                                self._need(L, copy_label(d, s), fg.variables[s])
                                smsgs.append((copy_label(d, s), lab[s]))
                (ups if up else downs).append(LevelSpec(L, cliques, pairs_of, smsgs, self.gibbsIters if up else self.downIters, copies=copies))
        return ups, downs

    # ---------------------------------------------------------------- "relative": anchor marginal + samples of anchor^-1 * separator
    def _specs_relative(self):
        from .graph import FactorGraph
        from .factors import Pose2, Point2
        from .clique import SampledPose2Pose2, SampledBearingRange
        fg, t = self.fg, self.tree
        ups, downs = [], []
        abs_msgs, rel_msgs = {}, {}          # clique -> [(source label, variable)] / [(anchor, separator, samples label)]
        msg_len = {}                         # clique -> [(anchor, separator, path length)]: the message tree's edges, for the parent's own tree
        self.anchor, self.unreached = {}, []
        down_parts, rel_parts = {}, {}
        for lvl in t.levels:
            L = FactorGraph(fg.N)
            cliques, pairs_of, smsgs, anchors, relatives = [], {}, [], [], []
            for cid in lvl:
                c = t.cliques[cid]
                F, S = c.frontals, c.separators
                vt = fg.variables
                pw = [(fl, self.findex[fl][1], self.findex[fl][2]) for fl in c.factors if len(self.findex[fl][1]) > 1]
                pri = [(fl, self.findex[fl][1][0], self.findex[fl][2]) for fl in c.factors if len(self.findex[fl][1]) == 1]
                for d in c.children:
                    for a_, s_, zl in rel_msgs[d]:
                        pw.append(("%s>%s|%d" % (a_, s_, d), [a_, s_], (SampledPose2Pose2 if vt[s_] is Pose2 else SampledBearingRange)(zl)))
                srcs = [(src, v) for d in c.children for src, v in abs_msgs[d]]
                anc = next((s for s in S if vt[s] is Pose2), None) if c.parent >= 0 else None
                self.anchor[cid] = anc
                upd, grp = [], []

                def add(rounds, lab, tag, with_unary):
                    for r, rnd in enumerate(rounds):
                        for v, fids, sids in rnd:
                            self._need(L, lab[v], vt[v])
                            rows = []
                            for fid, ls, f in pw:
                                if fid in fids:
                                    for o in ls:
                                        self._need(L, lab[o], vt[o])
                                    rows.append(self._lift(L, fid, cid, tag + lab[v] + ":", [lab[o] for o in ls], f))
                            if with_unary:
                                for fl, pv, f in pri:
                                    if pv == v:
                                        rows.append(self._lift(L, fl, cid, tag, [lab[v]], f))
                                for src in sids:
                                    if not str(src).startswith("prior:"):
                                        self._need(L, src, vt[v]); smsgs.append((src, lab[v]))
                            pairs_of[lab[v]] = rows
                            upd.append(lab[v]); grp.append(r)

                pwl = [(fid, ls) for fid, ls, _ in pw]
                unary = [("prior:" + fl, pv) for fl, pv, _ in pri] + srcs
                # ---- absolute solve: outward from the priors and the children's anchor marginals
                abs_msgs[cid], rel_msgs[cid] = [], []
                if unary:
                    lab = {v: v for v in F}
                    lab.update({s: "%s#%d" % (s, cid) for s in S})
                    rounds, left = _outward(list(F) + list(S), (), pwl, unary)
                    add(rounds, lab, "#", True)
                    reached = {v for rnd in rounds for v, _, _ in rnd}
                    if c.parent >= 0:
                        abs_msgs[cid] = [(lab[s], s) for s in ([anc] if anc is not None else S) if s in reached]
                    self.unreached += [(cid, v) for v in F if v not in reached]
                elif c.parent < 0:
                    raise ValueError("the root clique %r holds no prior and receives no absolute message: the graph has no gauge" % c)
                # ---- relative solves: outward from an anchor fixed at its current mean -- one per internal node of the message tree
                msg_len[cid] = []
                if anc is not None:
                    loc = [(ls[0], ls[1], 1.0) for fl in c.factors for ls in (self.findex[fl][1],) if len(ls) == 2]
                    loc += [e for d in c.children for e in msg_len[d]]
                    T = self._message_tree(list(F) + list(S), [s for s in S if vt[s] in (Pose2, Point2)], anc, loc,
                                           lambda v: vt[v] is Pose2, star=self.message_tree == "star")
                    kids = {}
                    for j, k, w in T:
                        kids.setdefault(j, []).append((k, w))
                    for j in [anc] + [j for j in kids if j != anc]:          # (the anchor's own solve exists even without children: relIters)
                        if j == anc:
                            lab = {v: "%s@%d" % (v, cid) for v in F + S}
                        else:
                            lab = {v: "%s@%d^%s" % (v, cid, j) for v in F + S}
                        self._need(L, lab[j], vt[j])
                        anchors.append((j, lab[j]))
                        rounds, left = _outward(list(F) + [s for s in S if s != j], (j,), pwl, ())
                        add(rounds, lab, "@" if j == anc else "@^%s" % j, False)
                        reached = {v for rnd in rounds for v, _, _ in rnd}
                        if j == anc:
                            rel_parts[cid] = (pw, reached, anc)
                        for k, w in kids.get(j, ()):
                            if k in reached:
                                zl = "%s~%d" % (k, cid)
                                if zl not in self.universe.variables:
                                    self.universe.addVariable(zl, vt[k])
                                relatives.append((lab[j], lab[k], zl)); rel_msgs[cid].append((j, k, zl)); msg_len[cid].append((j, k, w))
                cliques.append((upd, grp))
                down_parts[cid] = (pw, pri, srcs)
            cliques, pairs_of, smsgs = self._split_products(L, cliques, pairs_of, smsgs)
            ups.append(LevelSpec(L, cliques, pairs_of, smsgs, 1, anchors=anchors, relatives=relatives))
        # ---- down pass: the frontals outward from the separators (at their posteriors), the priors and the children's anchor marginals
        for lvl in t.levels:
            L = FactorGraph(fg.N)
            cliques, pairs_of, smsgs = [], {}, []
            for cid in lvl:
                c = t.cliques[cid]
                if c.parent < 0:
                    cliques.append(([], [])); continue
                pw, pri, srcs = down_parts[cid]
                vt = fg.variables
                F, S = c.frontals, c.separators
                rounds, left = _outward(list(F), S, [(fid, ls) for fid, ls, _ in pw], [("prior:" + fl, pv) for fl, pv, _ in pri] + srcs)
                upd, grp = [], []
                for r, rnd in enumerate(rounds):
                    for v, fids, sids in rnd:
                        self._need(L, v, vt[v])
                        rows = []
                        for fid, ls, f in pw:
                            if fid in fids:
                                for o in ls:
                                    self._need(L, o, vt[o])
                                rows.append(self._lift(L, fid, cid, "!" + v + ":", list(ls), f))
                        for fl, pv, f in pri:
                            if pv == v:
                                rows.append(self._lift(L, fl, cid, "!", [v], f))
                        for src in sids:
                            if not str(src).startswith("prior:"):
                                self._need(L, src, vt[v]); smsgs.append((src, v))
                        pairs_of[v] = rows
                        upd.append(v); grp.append(r)
                self.unreached += [(cid, v) for v in left]
                cliques.append((upd, grp))
            cliques, pairs_of, smsgs = self._split_products(L, cliques, pairs_of, smsgs)
            downs.append(LevelSpec(L, cliques, pairs_of, smsgs, 1))
        # ---- Gibbs sweeps over the frontals with every factor and message of the clique (roots after the up pass, the others after
        #      their outward down solve): colour classes of the clique's own graph
        def sweeps(lvl, roots, iters):
            if iters <= 0:
                return None
            L = FactorGraph(fg.N)
            cliques, pairs_of, smsgs = [], {}, []
            for cid in lvl:
                c = t.cliques[cid]
                if (c.parent < 0) != roots:
                    cliques.append(([], [])); continue
                pw, pri, srcs = down_parts[cid]
                F = list(c.frontals)
                nb = {v: set() for v in F}
                for _, ls, _ in pw:
                    for v in ls:
                        if v in nb:
                            nb[v].update(o for o in ls if o != v and o in nb)
                col = _colour(F, nb.__getitem__)
                for v in F:
                    self._need(L, v, fg.variables[v])
                    rows = []
                    for fid, ls, f in pw:
                        if v in ls:
                            for o in ls:
                                self._need(L, o, fg.variables[o])
                            rows.append(self._lift(L, fid, cid, "*" + v + ":", list(ls), f))
                    rows += [self._lift(L, fl, cid, "*", [v], f) for fl, pv, f in pri if pv == v]
                    for src, sv in srcs:
                        if sv == v:
                            self._need(L, src, fg.variables[v]); smsgs.append((src, v))
                    pairs_of[v] = rows
                cliques.append((F, [col[v] for v in F]))
            cliques, pairs_of, smsgs = self._split_products(L, cliques, pairs_of, smsgs)
            return LevelSpec(L, cliques, pairs_of, smsgs, iters)
        def rel_sweeps(lvl):
            if self.relIters <= 0:
                return None
            L = FactorGraph(fg.N)
            cliques, pairs_of = [], {}
            for cid in lvl:
                if cid not in rel_parts:
                    cliques.append(([], [])); continue
                pw, reached, anc = rel_parts[cid]
                lab = lambda v: "%s@%d" % (v, cid)
                known = reached | {anc}
                T = [v for v in t.cliques[cid].frontals + t.cliques[cid].separators if v in reached]
                nb = {v: set() for v in T}
                use = [(fid, ls, f) for fid, ls, f in pw if all(o in known for o in ls)]
                for _, ls, _ in use:
                    for v in ls:
                        if v in nb:
                            nb[v].update(o for o in ls if o != v and o in nb)
                col = _colour(T, nb.__getitem__)
                for v in T:
                    self._need(L, lab(v), fg.variables[v])
                    rows = []
                    for fid, ls, f in use:
                        if v in ls:
                            for o in ls:
                                self._need(L, lab(o), fg.variables[o])
                            rows.append(self._lift(L, fid, cid, "%" + v + ":", [lab(o) for o in ls], f))
                    pairs_of[lab(v)] = rows
                cliques.append(([lab(v) for v in T], [col[v] for v in T]))
            cliques, pairs_of, sm = self._split_products(L, cliques, pairs_of, [])
            return LevelSpec(L, cliques, pairs_of, sm, self.relIters)
        self.rel_specs = [rel_sweeps(lvl) for lvl in t.levels]
        self.root_specs = [sweeps(lvl, True, self.rootIters) for lvl in t.levels]
        self.refine_specs = [sweeps(lvl, False, self.refineIters) for lvl in t.levels]
        return ups, downs

    # ---------------------------------------------------------------- the solve
    def _run(self, plan, opts):
        o = type(opts).from_buffer_copy(opts)
        o.stream_offset = opts.stream_offset + (self.runs << 36)     # (it << 32) + family / product offsets stay below 2^36
        if self.shard is not None:
            self.shard.step(plan, o)
        else:
            plan.run(o)
        self.runs += 1

    def upload(self, fg=None):
        """current beliefs of every variable (IIF: initAll! has run) -> home blocks"""
        self.store.upload(fg or self.fg)

    def up(self, opts):
        if self.shard is not None:
            self.shard.bind_stream()      # before the first block operation: the pass is ONE in-order queue with the collectives
        for pre, pl, post, rp, rl in zip(self.up_pre, self.up_plans, self.up_post, self.root_plans, self.rel_plans):
            for op in pre:
                op.run()
            if pl is not None:
                self._run(pl, opts)
            if rl is not None:
                self._run(rl, opts)
            for op in post:
                op.run()
            if rp is not None:
                self._run(rp, opts)

    def down(self, opts):
        if self.shard is not None:
            self.shard.bind_stream()
        for pl, rf in zip(self.down_plans[::-1], self.refine_plans[::-1]):
            if pl is not None:
                self._run(pl, opts)
            if rf is not None:
                self._run(rf, opts)

    def solve(self, opts, passes=1):
        for _ in range(passes):
            self.up(opts)
            self.down(opts)

    def download(self, fg=None):
        """home blocks -> fg.vals (the posterior of every variable)"""
        self.store.download(fg or self.fg, labels=list(self.fg.variables))

    def stats(self):
        w = [len(l) for l in self.tree.levels]
        steps = lambda specs: sum(len(set(s.groups)) * s.gibbs_iters for s in specs if s.order)
        return dict(cliques=len(self.tree.cliques), levels=len(w), width_max=max(w), width_median=int(np.median(w)), messages=self.messages,
                    up_steps=steps(self.up_specs), down_steps=steps(self.down_specs),
                    up_rows=sum(len(s.pairs) for s in self.up_specs), down_rows=sum(len(s.pairs) for s in self.down_specs),
                    store_messages=sum(len(s.smsgs) for s in self.up_specs), relative_messages=sum(len(s.relatives) for s in self.up_specs),
                    blocks=len(self.universe.variables), unreached=len(getattr(self, "unreached", ())))


class _ShardedPlans:
    """backend view whose Plan() returns a FrontierShard level plan (share up-solve + exchange + scatter); block operations unchanged"""

    def __init__(self, backend, make):
        self.backend, self.make = backend, make

    def Plan(self, store, spec):
        return self.make(spec)

    def BlockOp(self, store, op, entries):
        return self.backend.BlockOp(store, op, entries)


class DeviceBackend:
    """TreeSolver on the device: `clique.DeviceStore` over the lifted universe, `TreeLevelPlan` per level, `rome_blockop_plan`s."""

    def __init__(self, ctx=None):
        self.ctx = ctx

    def Store(self, universe):
        from .clique import DeviceStore
        return DeviceStore(universe, ctx=self.ctx, upload=False)

    def Plan(self, store, spec, share=None, mirror=None):
        return TreeLevelPlan(store, spec, share=share, mirror=mirror)

    def BlockOp(self, store, op, entries):
        return BlockOpPlan(store, op, entries)


class BlockOpPlan:
    """block operations inside a DeviceStore (rome_blockop_plan), ONE launch per run:
    "copy" [(source, destination)], "anchor" [(belief, destination)], "relative" [(anchor block, separator block, destination)],
    "compose" [(a, b, destination, invert a, invert b)], "mix" [(pool, destination, p)]"""
    OPS = {"copy": 0, "anchor": 1, "relative": 2, "compose": 3, "mix": 4}

    def __init__(self, store, op, entries):
        import ctypes as C
        from . import _lib
        from .clique import DeviceStore
        self.store, self.ctx, self._lib = store, store.ctx, _lib.load()
        U = store.fg
        prm = None
        if op == "compose":      # entries (a, b, destination, invert a, invert b[, translation inflation, heading inflation])
            flags = [(0x100 if e[3] else 0) | (0x200 if e[4] else 0) for e in entries]
            if any(len(e) > 5 for e in entries):
                prm = np.ascontiguousarray([[e[5], e[6]] if len(e) > 5 else [1.0, 1.0] for e in entries], dtype=np.float64)
            entries = [e[:3] for e in entries]
        elif op == "mix":        # entries (pool, destination, p)
            flags = [int(e[2]) << 8 for e in entries]
            entries = [e[:2] for e in entries]
        else:
            flags = [0] * len(entries)
        ty = np.array([DeviceStore.TYPES.index(U.variables[e[-1]]) | f for e, f in zip(entries, flags)], dtype=np.int32)
        a = np.array([store.index[e[0]] for e in entries], dtype=np.int32)
        b = np.array([store.index[e[1]] for e in entries], dtype=np.int32) if op in ("relative", "compose") else None
        d = np.array([store.index[e[-1]] for e in entries], dtype=np.int32)
        self.dst_labels = [e[-1] for e in entries]
        PI = C.POINTER(C.c_int32)
        h = C.c_void_p()
        _lib.check(self._lib.rome_blockop_plan_create_ex(self.ctx.handle, store.handle, self.OPS[op], len(entries), ty.ctypes.data_as(PI), a.ctypes.data_as(PI),
                                                         b.ctypes.data_as(PI) if b is not None else None, d.ctypes.data_as(PI),
                                                         prm.ctypes.data_as(C.POINTER(C.c_double)) if prm is not None else None, C.byref(h)), self.ctx.handle)
        self.handle = h

    def run(self):
        from . import _lib
        _lib.check(self._lib.rome_blockop_plan_run(self.handle), self.ctx.handle)
        self.store.touched.update(self.dst_labels)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.rome_blockop_plan_destroy(self.handle); self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TreeLevelPlan:
    """A LevelSpec bound to a DeviceStore: `rome_upsolve_plan` with explicit (factor, destination) rows, groups and store-resident
    messages.  share: positions (within the level) of the cliques THIS plan updates -- Philox stream ids are positions in the WHOLE
    level's tables, so the shares of a level together draw what the unsharded plan draws."""

    def __init__(self, store, spec, share=None, mirror=None, outputs=False):
        import ctypes as C
        from . import _lib, api
        from .clique import CliqueBatch, CliqueUpsolveHost
        from .factors import Pose2, Point2, Pose3
        self.store, self.spec, self.ctx, self._lib = store, spec, store.ctx, _lib.load()
        L = spec.fg
        full = CliqueBatch(L, spec.pairs, var_index=store.index)
        sid = {pair: r for pair, (fam, r) in full.rows.items()}
        pos_t, cnt = {}, {Pose2: 0, Point2: 0, Pose3: 0}
        for l in spec.order:
            vt = L.variables[l]; pos_t[l] = cnt[vt]; cnt[vt] += 1
        mine = None if share is None else set(share)
        keep = [k for k in range(len(spec.order)) if mine is None or spec.owner[k] in mine]
        order = [spec.order[k] for k in keep]
        oset = set(order)
        self.order = order
        self.batch = full if mine is None else CliqueBatch(L, [p for p in spec.pairs if p[1] in oset], var_index=store.index, stream_ids=sid)
        for l in order:
            if l not in self.batch.vidx:
                self.batch.vidx[l] = store.index[l]
        u = CliqueUpsolveHost()
        kp = []
        self.res = self.batch._fill_upsolve(u, kp, order, spec.gibbs_iters, 1, "sequential", None, [spec.groups[k] for k in keep],
                                            up_stream=[pos_t[l] for l in order],
                                            up_mirror=None if mirror is None else [mirror.get(l, -1) for l in order], outputs=outputs)
        pos_of = {l: k for k, l in enumerate(order)}
        for ti, (vt, nm) in enumerate(((Pose2, "pose2"), (Point2, "point2"), (Pose3, "pose3"))):
            ms = [(store.index[s], pos_of[d]) for s, d in spec.smsgs if d in pos_of and L.variables[d] is vt]
            setattr(u, "n_smsg_" + nm, len(ms))
            if ms:
                a = np.array(ms, dtype=np.int32)
                src, up = np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1])
                kp += [src, up]
                setattr(u, "smsg_%s_src" % nm, src.ctypes.data_as(C.c_void_p)); setattr(u, "smsg_%s_up" % nm, up.ctypes.data_as(C.c_void_p))
        self.has_mirror = mirror is not None
        o = api.make_opts(N=L.N)
        o.layout = _lib.LAYOUT_SOA
        h = C.c_void_p()
        _lib.check(self._lib.rome_upsolve_plan_create(self.ctx.handle, store.handle, C.byref(o), C.byref(u), C.byref(h)), self.ctx.handle)
        self.handle = h

    def run(self, opts, mirror_out=None, mirror_stride=0):
        import ctypes as C
        from . import _lib
        o = _lib.Opts.from_buffer_copy(opts)
        o.layout = _lib.LAYOUT_SOA
        if hasattr(mirror_out, "data_ptr"):
            mirror_out = mirror_out.data_ptr()
        _lib.check(self._lib.rome_upsolve_plan_run(self.handle, C.byref(o), C.c_void_p(mirror_out or 0), int(mirror_stride)), self.ctx.handle)
        self.store.touched.update(self.order)
        if self.res:
            return {l: (new[k].copy(), bw[k].copy()) for vt, (ls, new, bw) in self.res.items() for k, l in enumerate(ls)}
        return None

    def close(self):
        if getattr(self, "handle", None):
            self._lib.rome_upsolve_plan_destroy(self.handle); self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
