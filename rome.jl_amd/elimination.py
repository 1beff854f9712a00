"""`solveTree!` for pose graphs as VARIABLE ELIMINATION IN RELATIVE-FACTOR ALGEBRA (`R.solveTree(fg, messages="elimination")`).

Why (round 6; docs/EXPERIMENTS.md, scripts/relative_elimination_surrogate.py, scripts/tree_gaussian_backend.py): the clique solves of
tree.TreeSolver are one-shot outward solves and their down pass multiplies proposals whose spread is the neighbours' ABSOLUTE spread --
the weights of a product then follow the (common, gauge-dominated) uncertainty of where the neighbours are instead of how tight each
factor is, and the solve stays at the quality of the init pass (measured on Manhattan-3500 against the converged MAP: init pass 0.8 /
tree solves 0.9 - 2.1 m after rigid alignment in the Gaussian restatement of the schedule; the exact elimination 0.18 m in ONE pass).
What an elimination needs is (a) the marginal over the neighbours of the eliminated variable and (b) the conditional of the variable
given its neighbours' VALUES.  For a pose graph both exist in sample form without any belief about where the poses are:

  edges     every factor between a and b is N samples of the relative pose a^-1 b (Pose2Pose2: mu + L xi; any SamplableBelief would do)
  merge     parallel edges between one pair multiply: the reference's `manifoldProduct` (manikde! bandwidths + multiscale Gibbs
            product) on the relative-pose samples
  eliminate v with neighbours u_1..u_m (edges z_k = v^-1 u_k): the pair marginal of (u_j, u_k) is EXACTLY the composition
            z_j^-1 (+) z_k, particle by particle (ROME_BLOCKOP_COMPOSE).  The dense marginal over the m neighbours is kept as a TREE
            (Chow-Liu node removal, as in pose-graph sparsification): the star about v's TIGHTEST neighbour c -- the minimum spanning
            tree when a pair costs spread(z_j) + spread(z_k).  m <= 2 is exact; the edge count never grows.
  order     rounds of independent (mutually non-adjacent) lowest-degree variables of the CURRENT graph, variables with priors last:
            every round is one product launch chain + one compose launch whatever the number of variables in it
  back      rounds in reverse: v = product over its neighbours of (mean of u_k) (+) z_k^-1 -- sampled-measurement convolutions from
            ANCHOR blocks (N copies of the neighbour's posterior mean: conditioning on the neighbours' values), multiplied by the
            reference's product.  A belief is therefore the conditional of a pose given its neighbours at their posterior means.

No init pass, no linearisation point, no iteration: a pass is a function of the seed.  `passes > 1` pool the particles of
independent passes (a belief is then a mixture over passes; its mean the running average).  Scope: Pose2 variables, Pose2Pose2 and
PriorPose2 factors (BASELINE configs[0-1]); anything else -> tree.TreeSolver.  Structural decisions (which neighbour is tightest, the
order) are taken on the host from first-order covariances of the edges ("shadow"), once per graph.

Reference: examples/ManhattanDatasetBatch.jl:43 (`tree = solveTree!(fg)`); the operations are the hot path's own
(sampled-measurement rows of `approxConvBelief`, `manikde!`, `manifoldProduct`)."""
import math

import numpy as np

from .tree import LevelSpec, ZERO, DeviceBackend, TreeSolver


def _inv(z):
    c, s = math.cos(z[2]), math.sin(z[2])
    return (-(c * z[0] + s * z[1]), -(-s * z[0] + c * z[1]), -z[2])


def _comp(a, b):
    c, s = math.cos(a[2]), math.sin(a[2])
    t = a[2] + b[2]
    return (a[0] + c * b[0] - s * b[1], a[1] + s * b[0] + c * b[1], math.atan2(math.sin(t), math.cos(t)))


class _Edge:
    """N samples of a^-1 b in store block `block`.  (m, vt, vth) = the SHADOW used for structural decisions only (which neighbour is
    tightest, what an elimination loses): mean relative pose, an isotropic translation variance and the heading variance, propagated to
    first order in plain float arithmetic (the lever arm |t|^2 * vth enters the translation variance of an inverse / a composition);
    s = (vt^2 vth)^(1/3), the geometric mean of the eigenvalues of diag(vt, vt, vth)."""
    __slots__ = ("a", "b", "block", "m", "vt", "vth", "s", "_rev")

    def __init__(self, a, b, block, m, vt, vth):
        self.a, self.b, self.block, self.m, self.vt, self.vth = a, b, block, m, vt, vth
        self.s = (vt * vt * vth) ** (1.0 / 3.0)
        self._rev = None

    def seen_from(self, v):
        """(mean, vt, vth) of v^-1 other"""
        if v == self.a:
            return self.m, self.vt, self.vth
        if self._rev is None:
            m = self.m
            self._rev = (_inv(m), self.vt + (m[0] * m[0] + m[1] * m[1]) * self.vth, self.vth)
        return self._rev


class RelativeEliminationSolver:
    """interface of tree.TreeSolver (upload / solve / download / stats / store); backend as there (device by default)"""

    def __init__(self, fg, backend=None, ctx=None, max_product=8, shard=None, loss_slack=1e9, loss_factor=1.0, priors_last=1, order_seed=0, structures=1, centre="tight", near_weight=0.0, mesh_max=0):
        from .factors import Pose2
        from .graph import FactorGraph
        why = self.covers(fg, why=True)
        if why is not True:
            raise TypeError("messages='elimination' covers Pose2 graphs of Pose2Pose2 / PriorPose2 factors without hypotheses (%s): use messages='relative' / 'marginal'" % why)
        self.fg, self.N, self.messages = fg, fg.N, "elimination"
        self.backend = backend or DeviceBackend(ctx)
        self.max_product = int(max_product or 0)
        self.loss_slack, self.loss_factor, self.priors_last = float(loss_slack), float(loss_factor), bool(priors_last)
        self.order_seed = int(order_seed)
        self.centre = centre
        self.near_weight = float(near_weight)
        self.mesh_max = int(mesh_max)
        self.findex = {fl: (fl, ls, f) for fl, ls, f in fg.factors}
        U = FactorGraph(fg.N)
        for l, vt in fg.variables.items():
            U.addVariable(l, vt)
        U.addVariable(ZERO, Pose2)
        self.universe = U
        self.Pose2 = Pose2
        for l in list(fg.variables):
            U.addVariable(l + "^", Pose2)          # anchor block: N copies of the posterior mean
            U.addVariable(l + "&", Pose2)          # pool block: the mixture over the passes so far (solve(passes > 1))
        from .graph import gc_paused
        self.structures = max(1, int(structures))
        with gc_paused():     # (~1e6 live small objects, nothing to free: the generation scans cost a third of the build)
            self._build(fg, shard)
        self._to_pool = self.backend.BlockOp(self.store, "copy", [(l, l + "&") for l in fg.variables])
        self._mix = {}
        self.runs = 0
        self.passes_pooled = 0

    def _build(self, fg, shard):
        import time
        U = self.universe
        t0 = time.perf_counter()
        schedules = [self._structure(k) for k in range(self.structures)]
        self.schedules = schedules
        t1 = time.perf_counter()
        self.store = self.backend.Store(U)
        self.store.put(ZERO, np.zeros((3, fg.N)))
        B = self.backend
        self.shard = shard(self.store) if shard is not None else None
        if self.shard is not None:
            from .tree import _ShardedPlans
            base = B
            B = _ShardedPlans(base, lambda s: self.shard.plan_level(s, base.Plan))
        self._B = B
        self.steps = []
        for sched in schedules:
            st = []
            for kind, x in sched:
                st.append(("plan", B.Plan(self.store, x)) if kind == "plan" else ("op", B.BlockOp(self.store, kind, x)))
            self.steps.append(st)
        self.build_s = dict(structure=t1 - t0, store_and_plans=time.perf_counter() - t1)

    @staticmethod
    def covers(fg, why=False):
        """True when the graph is in this solver's scope (else False, or the reason with why=True)"""
        from .factors import Pose2, Pose2Pose2, PriorPose2
        bad = None
        for l, vt in fg.variables.items():
            if vt is not Pose2:
                bad = "variable %s is %s" % (l, getattr(vt, "__name__", type(vt).__name__.lstrip("_"))); break
        if bad is None:
            for fl, ls, f in fg.factors:
                if not isinstance(f, (Pose2Pose2, PriorPose2)) or fl in fg.multihypo or fl in getattr(fg, "nullhypo", {}):
                    bad = "factor %s" % fl; break
        if bad is None and not any(isinstance(f, PriorPose2) for _, _, f in fg.factors):
            bad = "no prior"
        return (True if bad is None else bad) if why else bad is None

    # borrowed helpers (they use self.universe / self.max_product / self.fg only)
    _need = TreeSolver._need
    _lift = TreeSolver._lift
    _split_products = TreeSolver._split_products

    # ------------------------------------------------------------------------------------------------ structure (host, once per graph)
    def _spec(self, L, entries, smsgs=()):
        """entries: [(destination label, [factor labels])] -> a one-group LevelSpec (every destination its own 'clique': a level is dealt
        to the ranks by variable), two-stage products where a destination has more than max_product proposals"""
        cliques = [([l], [0]) for l, _ in entries]
        pairs_of = {l: list(rows) for l, rows in entries}
        cliques, pairs_of, sm = self._split_products(L, cliques, pairs_of, list(smsgs))
        return LevelSpec(L, cliques, pairs_of, sm, 1)

    def _loss(self, v, adj):
        """what eliminating v NOW costs, and the centre of its star: the star about neighbour c replaces the pair (j, k) of spread
        s_j + s_k by the path j - c - k of spread s_j + s_k + 2 s_c; summed relative loss of information over the dropped pairs,
        discounted where the pair already has a direct edge (scalar spreads of the shadow covariances).  c = the neighbour with the
        smallest loss (centre="loss") or the tightest one (centre="tight").  Degree <= 2: exact, loss 0.  -> (loss, c)"""
        nb = adj[v]
        s = {u: (es[0].s if len(es) == 1 else 1.0 / sum(1.0 / e.s for e in es)) for u, es in nb.items()}
        if not s:
            return 0.0, None
        tight = min(s, key=lambda u: (s[u], self._pos[u]))
        if len(nb) <= max(2, self.mesh_max):
            return 1e-9 * (len(nb) > 2), tight
        us = list(s)
        ex = {}
        for a in range(len(us)):
            for b in range(a + 1, len(us)):
                es = adj[us[a]].get(us[b])
                ex[(a, b)] = sum(1.0 / e.s for e in es) if es else 0.0
        best = None
        for ci in (range(len(us)) if self.centre == "loss" else [us.index(tight)]):
            sc, tot = s[us[ci]], 0.0
            for a in range(len(us)):
                if a == ci:
                    continue
                for b in range(a + 1, len(us)):
                    if b == ci:
                        continue
                    sj = s[us[a]] + s[us[b]]
                    it = 1.0 / sj
                    tot += (it - 1.0 / (sj + 2 * sc)) / (it + ex[(a, b)])
            if best is None or (tot, self._pos[us[ci]]) < (best[0], self._pos[best[1]]):
                best = (tot, us[ci])
        return best

    def _structure(self, k_struct=0):
        from .clique import SampledPose2Pose2
        from .factors import Pose2Pose2, PriorPose2
        from .graph import FactorGraph
        fg, N, Pose2 = self.fg, self.N, self.Pose2
        U = self.universe
        n_edge = [0]

        def new_block(prefix):
            l = "%s%d.%d~" % (prefix, k_struct, n_edge[0]); n_edge[0] += 1
            U.addVariable(l, Pose2)
            return l

        adj = {v: {} for v in fg.variables}            # v -> {u: [edges]}
        unary = {v: [] for v in fg.variables}          # v -> [blocks of ABSOLUTE samples of v]: priors, and priors transported along edges
        sched = []
        # ---- step 0: the samples of every factor's measurement (a row from the ZERO block: 0 (+) z = z; a prior: its own row), ONE launch chain
        L = FactorGraph(N); L.addVariable(ZERO, Pose2)
        ent = []
        for fl, ls, f in fg.factors:
            if isinstance(f, PriorPose2):
                blk = new_block("p")
                L.addVariable(blk, Pose2)
                ent.append((blk, [self._lift(L, fl, 0, "p", [blk], f)]))
                unary[ls[0]].append(blk)
                continue
            a, b = ls
            blk = new_block("e")
            L.addVariable(blk, Pose2)
            nfl = "s:" + fl
            L.factors.append((nfl, [ZERO, blk], f)); L._findex[nfl] = L.factors[-1]
            ent.append((blk, [nfl]))
            C = f.Z.cov
            e = _Edge(a, b, blk, (float(f.Z.mu[0]), float(f.Z.mu[1]), float(f.Z.mu[2])), math.sqrt(max(float(C[0, 0] * C[1, 1] - C[0, 1] * C[1, 0]), 1e-300)), float(C[2, 2]))
            adj[a].setdefault(b, []).append(e); adj[b].setdefault(a, []).append(e)
        pri = {v for v in fg.variables if unary[v]}
        if not pri:
            raise ValueError("the graph holds no prior: it has no gauge")
        sched.append(("plan", self._spec(L, ent)))
        self.n_factor_edges = len(ent) - sum(len(u) for u in unary.values())
        # ---- rounds
        alive = set(fg.variables)
        hold = set(pri) if self.priors_last else set()
        pos = {v: k for k, v in enumerate(fg.variables)}
        if self.order_seed or k_struct:   # another tie-break among equal losses / degrees: another structure, another set of approximations
            perm = np.random.default_rng(self.order_seed + 7919 * k_struct).permutation(len(pos))
            pos = {v: int(perm[k]) for v, k in pos.items()}
        self._pos = pos
        dist = {v: 0.0 for v in fg.variables}
        if self.near_weight:      # hops to the nearest prior variable, scaled to [0, 1]
            from collections import deque
            dq = deque(pri); seen = {v: 0 for v in pri}
            while dq:
                v = dq.popleft()
                for u in adj[v]:
                    if u not in seen:
                        seen[u] = seen[v] + 1; dq.append(u)
            mx = max(seen.values()) or 1
            dist = {v: seen.get(v, mx) / mx for v in fg.variables}
        rounds, down = [], []
        n_merge = n_comp = n_approx = n_transport = 0

        def identity_rows(Lx, dst, blocks, flip=()):
            """rows that re-emit the samples of `blocks` as proposals of `dst` (0 (+) z = z; flipped: 0 (-) z = z^-1)"""
            rows = []
            for k, blk in enumerate(blocks):
                if blk not in Lx.variables:
                    Lx.addVariable(blk, Pose2)
                nfl = "g:%s:%d" % (dst, k)
                labels = [dst, ZERO] if k in flip else [ZERO, dst]
                Lx.factors.append((nfl, labels, SampledPose2Pose2(blk))); Lx._findex[nfl] = Lx.factors[-1]
                rows.append(nfl)
            return rows

        while alive:
            pool = (alive - hold) or alive
            lc = {v: self._loss(v, adj) for v in pool}
            loss = {v: x[0] for v, x in lc.items()}
            cand = sorted(pool, key=lambda v: (loss[v] + self.near_weight * dist[v], len(adj[v]), pos[v]))
            sel, blocked = [], set()
            cap = loss[cand[0]] * self.loss_factor + self.loss_slack
            for v in cand:
                if loss[v] > cap:
                    break
                if v in blocked:
                    continue
                if not adj[v] and not unary[v]:
                    raise ValueError("variable %s is not connected to a prior" % v)
                sel.append(v); blocked.add(v); blocked.update(adj[v])
            Lm = FactorGraph(N); Lm.addVariable(ZERO, Pose2)
            Lt = FactorGraph(N)
            merges, transports, comps, dn = [], [], [], []
            for v in sel:
                nb = {}
                for u, es in adj[v].items():
                    if len(es) > 1:     # parallel edges -> one: product of their samples, all seen from v
                        blk = new_block("m")
                        Lm.addVariable(blk, Pose2)
                        m0 = es[0].seen_from(v)[0]
                        it = ith = hx = hy = hth = 0.0
                        for e in es:
                            m_, vt_, vth_ = e.seen_from(v)
                            dth = m_[2] - m0[2]
                            dth = math.atan2(math.sin(dth), math.cos(dth))
                            it += 1.0 / vt_; ith += 1.0 / vth_
                            hx += (m_[0] - m0[0]) / vt_; hy += (m_[1] - m0[1]) / vt_; hth += dth / vth_
                        mm = (m0[0] + hx / it, m0[1] + hy / it, m0[2] + hth / ith)
                        merges.append((blk, identity_rows(Lm, blk, [e.block for e in es], flip={k for k, e in enumerate(es) if e.a != v})))
                        n_merge += 1
                        nb[u] = _Edge(v, u, blk, mm, 1.0 / it, 1.0 / ith)
                    else:
                        nb[u] = es[0]
                if len(unary[v]) > 1:   # several absolute beliefs of v -> one
                    blk = new_block("q")
                    Lm.addVariable(blk, Pose2)
                    merges.append((blk, identity_rows(Lm, blk, unary[v]))); n_merge += 1
                    unary[v] = [blk]
                for u in nb:
                    del adj[u][v]
                if nb and 2 < len(nb) <= self.mesh_max:
                    # STAR-MESH transform (Kron reduction): the marginal of a star with leg variances v_k is EXACTLY (commuting case) the
                    # full mesh whose edge (j, k) has the composed mean and the variance v_j + v_k + v_j v_k sum_{i != j,k} 1 / v_i --
                    # the composition's own spread, inflated by what the pair shares with the other legs
                    us = list(nb)
                    seen = {u: nb[u].seen_from(v) for u in us}
                    it_all = sum(1.0 / seen[u][1] for u in us); ith_all = sum(1.0 / seen[u][2] for u in us)
                    for ai in range(len(us)):
                        for bi in range(ai + 1, len(us)):
                            j, k = us[ai], us[bi]
                            zj, vtj, vthj = seen[j]; zk, vtk, vthk = seen[k]
                            zi = _inv(zj); vti = vtj + (zj[0] * zj[0] + zj[1] * zj[1]) * vthj
                            vt0 = vti + vtk + (zk[0] * zk[0] + zk[1] * zk[1]) * vthj
                            vth0 = vthj + vthk
                            gt = 1.0 + vtj * vtk * (it_all - 1.0 / vtj - 1.0 / vtk) / (vtj + vtk)
                            gth = 1.0 + vthj * vthk * (ith_all - 1.0 / vthj - 1.0 / vthk) / (vthj + vthk)
                            blk = new_block("c")
                            e = _Edge(j, k, blk, _comp(zi, zk), vt0 * gt, vth0 * gth)
                            comps.append((nb[j].block, nb[k].block, blk, nb[j].a == v, nb[k].a != v, math.sqrt(gt), math.sqrt(gth)))
                            adj[j].setdefault(k, []).append(e); adj[k].setdefault(j, []).append(e)
                            n_comp += 1
                elif nb:
                    # tightest neighbour: smallest log det of the covariance of v^-1 u
                    c = lc[v][1]
                    zi, vti, vthi = nb[c].seen_from(c)             # c^-1 v
                    for k, ek in nb.items():
                        if k == c:
                            continue
                        zk, vtk, vthk = ek.seen_from(v)
                        blk = new_block("c")
                        e = _Edge(c, k, blk, _comp(zi, zk), vti + vtk + (zk[0] * zk[0] + zk[1] * zk[1]) * vthi, vthi + vthk)
                        # c^-1 k = (v^-1 c)^-1 (+) (v^-1 k): invert the block of c when it holds v^-1 c, the block of k when it holds k^-1 v
                        comps.append((nb[c].block, ek.block, blk, nb[c].a == v, ek.a != v))
                        adj[c].setdefault(k, []).append(e); adj[k].setdefault(c, []).append(e)
                        n_comp += 1
                    if unary[v]:        # the absolute belief of v travels to c: p(c) = p(v) (+) z_c -- a convolution of BELIEF samples
                        blk, pv, ec = new_block("a"), unary[v][0], nb[c]
                        for l in (blk, pv, ec.block):
                            if l not in Lt.variables:
                                Lt.addVariable(l, Pose2)
                        nfl = "t:%s" % blk
                        Lt.factors.append((nfl, [pv, blk] if ec.a == v else [blk, pv], SampledPose2Pose2(ec.block))); Lt._findex[nfl] = Lt.factors[-1]
                        transports.append((blk, [nfl])); unary[c].append(blk); n_transport += 1
                    n_approx += len(nb) > 2
                dn.append((v, list(nb.items()), list(unary[v])))
                del adj[v]
            if merges:
                sched.append(("plan", self._spec(Lm, merges)))
            if transports:
                sched.append(("plan", self._spec(Lt, transports)))
            if comps:
                sched.append(("compose", comps))
            rounds.append(len(sel)); down.append(dn)
            alive -= set(sel)
        # ---- back substitution: rounds in reverse; a variable = product of its own absolute beliefs (store-resident messages) and the
        #      sampled-measurement convolutions from its neighbours' ANCHOR blocks
        anchor = lambda v: v + "^"     # noqa: E731
        anchored = set()
        for dn in reversed(down):
            Ld = FactorGraph(N)
            need, ent, sm = [], [], []
            for v, nbs, un in dn:
                Ld.addVariable(v, Pose2)
                rows = []
                for u, e in nbs:
                    if u not in anchored:
                        anchored.add(u); need.append((u, anchor(u)))
                    for l in (anchor(u), e.block):
                        if l not in Ld.variables:
                            Ld.addVariable(l, Pose2)
                    nfl = "d:%s:%s:%s" % (v, u, e.block)
                    labels = [v, anchor(u)] if e.a == v else [anchor(u), v]
                    Ld.factors.append((nfl, labels, SampledPose2Pose2(e.block))); Ld._findex[nfl] = Ld.factors[-1]
                    rows.append(nfl)
                for blk in un:
                    if blk not in Ld.variables:
                        Ld.addVariable(blk, Pose2)
                    sm.append((blk, v))
                ent.append((v, rows))
            if need:
                sched.append(("anchor", need))
            sched.append(("plan", self._spec(Ld, ent, sm)))
        self.rounds = rounds
        self._stats = dict(rounds=len(rounds), round_sizes=rounds[:16], merges=n_merge, compositions=n_comp, transports=n_transport,
                           approximated_eliminations=n_approx, factor_edges=self.n_factor_edges, prior_variables=len(pri), blocks=len(U.variables),
                           launch_steps=len(sched), plan_steps=sum(1 for k, _ in sched if k == "plan"), structures=self.structures)
        return sched

    # ------------------------------------------------------------------------------------------------ the solve
    def _run(self, plan, opts):
        o = type(opts).from_buffer_copy(opts)
        o.stream_offset = opts.stream_offset + (self.runs << 36)
        if self.shard is not None:
            self.shard.step(plan, o)
        else:
            plan.run(o)
        self.runs += 1

    def upload(self, fg=None):
        """(nothing to upload: the solve starts from the factors alone; present so that solveTree treats every solver alike)"""

    def solve(self, opts, passes=1):
        """passes > 1 (and repeated calls): independent passes -- other Philox streams, the next of the `structures` -- POOLED: after pass p
        a belief holds ~N / p particles of every pass so far (ROME_BLOCKOP_MIX), its mean is the running average.  reset() starts over."""
        for _ in range(passes):
            if self.shard is not None:
                self.shard.bind_stream()
            p = self.passes_pooled + 1
            if p > 1:
                self._to_pool.run()
            for kind, x in self.steps[(p - 1) % self.structures]:
                if kind == "plan":
                    self._run(x, opts)
                else:
                    x.run()
            if p > 1:
                if p not in self._mix:
                    self._mix[p] = self.backend.BlockOp(self.store, "mix", [(l + "&", l, p) for l in self.fg.variables])
                self._mix[p].run()
            self.passes_pooled = p

    def reset(self):
        self.passes_pooled = 0

    def download(self, fg=None):
        self.store.download(fg or self.fg, labels=list(self.fg.variables))

    def stats(self):
        return dict(self._stats, messages=self.messages, build_s=self.build_s)
