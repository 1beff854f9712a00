"""CPU stand-in for DeviceGraph (TEST INFRASTRUCTURE): just enough of its interface for the multi-GPU drivers of
rome_jl_amd.distributed -- families(), family_table(), bel, _plan() -- on torch CPU tensors, with the ORACLE doing the compute
of a launch, so that the drivers' buffering / ghost addressing / collectives run for real over gloo."""
import numpy as np
import torch


class OracleDG:
    def __init__(self, R, fg, seed=5):
        import oracle as ro
        self.torch, self.N, self.R, self.ro, self.seed = torch, fg.N, R, ro, seed
        pk = R.PackedGraph(fg)
        self.packed = pk
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.int32)))
        self.bel = {vt: torch.as_tensor(pk.beliefs(fg, vt)) if len(pk.labels[vt]) else torch.zeros((0, vt.dim, fg.N), dtype=torch.float64)
                    for vt in (R.Pose2, R.Point2, R.Pose3)}
        self.tabs = {}
        if pk.p2p2["F"] or pk.prior2["F"]:
            factor, dr, fixed, target = R.PackedGraph.conv_table(pk.p2p2)
            F, P = pk.p2p2["F"], pk.prior2["F"]
            mu = np.concatenate([pk.p2p2["mu"].reshape(F, 3), pk.prior2["mu"].reshape(P, 3)])
            cov = np.concatenate([pk.p2p2["cov"].reshape(F, 3, 3), pk.prior2["cov"].reshape(P, 3, 3)])
            rows = np.stack([np.concatenate([factor, F + np.arange(P)]), np.concatenate([dr, np.full(P, 2)]),
                             np.concatenate([fixed, pk.prior2["var"]]), np.concatenate([target, pk.prior2["var"]])], axis=1)
            self.tabs["p2p2"] = dict(n=2 * F + P, fn="p2p2", vt_fixed=R.Pose2, vt_target=R.Pose2, dir_all=0, rows4=i32(rows),
                                     mu=mu, L=np.array([ro.cholesky_lower(c) for c in cov]), alt=None, w=None)
        if pk.br["F"]:
            b = pk.br; r0 = b["rows0"]; Fb = b["F"]
            mh = bool((b["alt"] >= 0).any())   # multihypo: alternative landmark per row (-1: none), P(primary) per row
            self.tabs["br1"] = dict(n=Fb, fn="br1", vt_fixed=R.Point2, vt_target=R.Pose2, dir_all=1, mu=b["mu"], L=b["sigma"],
                                    alt=i32(b["alt"]) if mh else None, w=np.asarray(b["w"], dtype=np.float64) if mh else None,
                                    rows4=i32(np.stack([np.arange(Fb), np.ones(Fb), b["point"], b["pose"]], axis=1)))
            self.tabs["br0"] = dict(n=len(r0["factor"]), fn="br0", vt_fixed=R.Pose2, vt_target=R.Point2, dir_all=0, mu=b["mu"], L=b["sigma"],
                                    alt=i32(r0["alt"]) if mh else None, w=np.asarray(r0["w"], dtype=np.float64) if mh else None,
                                    rows4=i32(np.stack([r0["factor"], np.zeros(len(r0["factor"])), r0["pose"], r0["point"]], axis=1)))

    def families(self):
        return [f for f in ("p2p2", "br1", "br0") if f in self.tabs]

    def family_table(self, fam):
        return self.tabs[fam]

    def _plan(self, fn, opts, **kw):
        ro, N = self.ro, self.N
        rows = kw["rows4"].numpy()
        out, mu, L = kw["out"], kw["mu"], kw["L"]
        bf, bt = kw["bel_fixed"], kw["bel_target"]
        mirror_rows, mirror_out = kw.get("mirror_row", ()), kw.get("mirror_out")
        if kw.get("mirror_map") is not None:   # row -> slot map: the rows in slot order
            mm = kw["mirror_map"].numpy()
            mirror_rows = [int(np.nonzero(mm == m)[0][0]) for m in range(int(mm.max()) + 1)] if (mm >= 0).any() else ()
        if kw.get("nullhypo") is not None:
            raise NotImplementedError("the oracle stand-in does not model per-row nullhypo columns")
        alt = kw["alt_var"].numpy() if kw.get("alt_var") is not None else None
        hw = np.asarray(kw["hypo_w"], dtype=np.float64) if kw.get("hypo_w") is not None else None
        o_keep = type(opts).from_buffer_copy(opts) if opts is not None else None   # (a cached plan's stream offset may be updated in place)
        seed = self.seed

        def launch():
            base = int(o_keep.stream_offset) if o_keep is not None else 0
            n = len(rows)
            res = np.zeros(tuple(out.shape))
            for k in range(n):   # one oracle call per row: Philox stream = base + row position, like the kernel
                o = ro.make_opts(N=N, solver=ro.SOLVER_NEWTON, seed=seed, stream_offset=base + k)
                f, d, fv, tv = (int(x) for x in rows[k])
                if fn == "p2p2":
                    if d == 2:
                        res[k] = ro.sample_priorpose2(o, mu[f], L[f])[0]
                    else:
                        res[k] = ro.conv_pose2pose2(o, mu, L, bf.numpy(), [fv], [tv], [d], factor=[f])[0]
                else:
                    mhkw = {} if alt is None else dict(alt_var=[int(alt[k])], hypo_w=[float(hw[k])])
                    res[k] = ro.conv_pose2point2br(o, 1 if fn == "br1" else 0, mu, L, bf.numpy(), bt.numpy(), [fv], [tv], factor=[f], **mhkw)[0]
            out.copy_(torch.as_tensor(res))
            for m, r in enumerate(mirror_rows):
                blk = out[r].reshape(-1)
                mirror_out[m * blk.numel():(m + 1) * blk.numel()].copy_(blk)
        launch._keep = (None, o_keep)
        return launch

    # ---- the row-range stages of TargetShardedSweep.solve_step, computed by the oracle ----
    STREAM_PROD2 = 3 << 28

    def kde_bandwidth_rows(self, dim, n_rows, prop, circ, bw_out):
        bw_out.copy_(torch.as_tensor(self.ro.kde_bandwidths(prop[:n_rows].numpy(), circ)))

    def product_gibbs_rows(self, opts, dim, V, ptr, rows, prop, bw, n_rows, bel_in, bel_out, circ, iters, max_k):
        o = self.ro.make_opts(N=self.N, seed=self.seed, stream_offset=int(opts.stream_offset))
        out = self.ro.product_msgibbs(o, dim, ptr.numpy(), rows.numpy(), prop[:max(n_rows, 1)].numpy(), bw[:max(n_rows, 1)].numpy(),
                                      bel_in.numpy(), circ, iters)
        bel_out.copy_(torch.as_tensor(out))


# ---------------------------------------------------------------------------------------------------------------------------------
# stand-ins for rome_jl_amd.clique.DeviceStore / UpsolvePlan / ScatterPlan (the device-resident frontier path of FrontierShard): the
# "device" is a private copy of the graph's beliefs, the up-solve is the oracle's restatement of rome_clique_upsolve with the stream ids
# of plan_frontier, the mirror / scatter are torch copies -- so that FrontierShard's dealing, block layout and collective run for real.
class OracleStore:
    def __init__(self, R, fg):
        import copy
        self.R, self.fg, self.N = R, fg, fg.N
        self.dev = copy.copy(fg)                 # same variables / factors, private belief dict = the "HBM" copy
        self.dev.vals = {l: v.copy() for l, v in fg.vals.items()}
        self.index = {}
        cnt = {}
        for l, t in fg.variables.items():
            self.index[l] = cnt.get(t, 0); cnt[t] = cnt.get(t, 0) + 1

    def get(self, label):
        return self.dev.vals[label]


class OraclePlan:
    def __init__(self, store, cliques, share=None, gibbsIters=3, Niter=1, mirror=None, usable=None, outputs=False, seed_solver=1):
        from rome_jl_amd.clique import plan_frontier
        self.store, self.gi, self.pi, self.mirror = store, gibbsIters, Niter, mirror
        self.fp = plan_frontier(store.fg, cliques, share, usable)

    def run(self, opts, mirror_out=None, mirror_stride=0):
        from solve_ref import upsolve_ref
        st, fp = self.store, self.fp
        ref = upsolve_ref(st.R, st.dev, fp["order"], st.N, seed=int(opts.seed), gibbs_iters=self.gi, product_iters=self.pi, groups=fp["groups"],
                          stream_offset=int(opts.stream_offset), stream_ids=fp["stream_ids"], up_stream=fp["up_stream"], solver=int(opts.solver))
        for l in fp["order"]:
            st.dev.vals[l] = ref[l]
            if self.mirror is not None and self.mirror.get(l, -1) >= 0:
                blk = torch.as_tensor(ref[l].reshape(-1))
                o = self.mirror[l] * int(mirror_stride or 6 * st.N)
                mirror_out[o:o + blk.numel()].copy_(blk)


class OracleScatter:
    def __init__(self, store, labels, src_blocks, stride=0):
        self.store, self.labels, self.blocks, self.stride = store, list(labels), list(src_blocks), int(stride or 6 * store.N)

    def run(self, src):
        st = self.store
        for l, b in zip(self.labels, self.blocks):
            d = st.fg.variables[l].dim
            st.dev.vals[l] = src[b * self.stride:b * self.stride + d * st.N].numpy().reshape(d, st.N).copy()


# ---------------------------------------------------------------------------------------------------------------------------------
# stand-in backend for rome_jl_amd.tree.TreeSolver: the "HBM" is a dict over the lifted universe (home labels + separator copies), a
# level plan is the oracle's restatement of rome_clique_upsolve over the level's own factor graph (same rows, groups, stream ids and
# store-resident messages as tree.TreeLevelPlan hands the library).
class OracleTreeStore:
    def __init__(self, R, universe):
        self.R, self.fg, self.N = R, universe, universe.N
        self.vals = {}
        self.index = {}
        cnt = {}
        for l, t in universe.variables.items():
            self.index[l] = cnt.get(t, 0); cnt[t] = cnt.get(t, 0) + 1

    def upload(self, fg, labels=None):
        for l in self.fg.variables:
            if fg.isInitialized(l) and (labels is None or l in labels):
                self.vals[l] = np.array(fg.getVal(l), dtype=np.float64)

    def download(self, fg, labels=None):
        for l, v in self.vals.items():
            if labels is None or l in labels:
                fg.vals[l] = v.copy()

    def get(self, label):
        return self.vals[label]

    def put(self, label, pts):
        self.vals[label] = np.array(pts, dtype=np.float64)


class OracleTreePlan:
    def __init__(self, store, spec, share=None, mirror=None):
        self.store, self.spec, self.share, self.mirror = store, spec, share, mirror
        L = spec.fg
        from rome_jl_amd.clique import CliqueBatch
        full = CliqueBatch(L, spec.pairs)
        self.sid = {pair: r for pair, (fam, r) in full.rows.items()}
        types = (store.R.Pose2, store.R.Point2, store.R.Pose3)
        self.pos_t, cnt = {}, {t: 0 for t in types}
        for l in spec.order:
            vt = L.variables[l]; self.pos_t[l] = cnt[vt]; cnt[vt] += 1
        mine = None if share is None else set(share)
        self.keep = [k for k in range(len(spec.order)) if mine is None or spec.owner[k] in mine]

    def run(self, opts, mirror_out=None, mirror_stride=0):
        from solve_ref import upsolve_ref
        st, sp = self.store, self.spec
        L = sp.fg
        L.vals = {l: st.vals[l] for l in L.variables if l in st.vals}
        order = [sp.order[k] for k in self.keep]
        if not order:
            return
        oset = set(order)
        msgs = {}
        for src, dst in sp.smsgs:
            if dst in oset:
                msgs.setdefault(dst, []).append(st.vals[src])
        # the share's own factor graph view: upsolve_ref enumerates the factors of its destinations itself
        ref = upsolve_ref(st.R, L, order, st.N, seed=int(opts.seed), gibbs_iters=sp.gibbs_iters, product_iters=1,
                          groups=[sp.groups[k] for k in self.keep], stream_offset=int(opts.stream_offset), stream_ids=self.sid,
                          up_stream=self.pos_t, solver=int(opts.solver), messages=msgs, usable=lambda l: True, meas_vals=st.vals,
                          pairs=[p for p in sp.pairs if p[1] in oset])
        for l in order:
            st.vals[l] = ref[l]
            if self.mirror is not None and self.mirror.get(l, -1) >= 0:
                blk = torch.as_tensor(ref[l].reshape(-1))
                o = self.mirror[l] * int(mirror_stride or 6 * st.N)
                mirror_out[o:o + blk.numel()].copy_(blk)


class OracleTreeBlockOp:
    """numpy restatement of rome_blockop_plan: copy / anchor (N copies of the mean; Pose2: circular mean heading; Pose3: rotation of
    particle 0) / relative (tangent coordinates of ref^-1 * s_i, or (bearing, range) of a landmark seen from ref)"""

    def __init__(self, store, op, entries):
        self.store, self.op, self.entries = store, op, list(entries)

    def run(self):
        v, N = self.store.vals, self.store.N
        for e in self.entries:
            if self.op == "copy":
                v[e[1]] = v[e[0]].copy()
            elif self.op == "anchor":
                b = v[e[0]]
                m = b.mean(axis=1)
                if b.shape[0] == 3:
                    m[2] = np.arctan2(np.sin(b[2]).sum(), np.cos(b[2]).sum())
                elif b.shape[0] == 6:
                    m[3:] = b[3:, 0]
                v[e[1]] = np.repeat(m[:, None], N, axis=1)
            elif self.op == "mix":
                keep = (np.arange(N) % int(e[2])) == int(e[2]) - 1
                v[e[1]] = np.where(keep[None, :], v[e[1]], v[e[0]])
            elif self.op == "compose":
                def inv(z):
                    c, s = np.cos(z[2]), np.sin(z[2])
                    return np.stack([-(c * z[0] + s * z[1]), -(-s * z[0] + c * z[1]), -z[2]])
                A, B = (inv(v[e[0]]) if e[3] else v[e[0]]), (inv(v[e[1]]) if e[4] else v[e[1]])
                c, s = np.cos(A[2]), np.sin(A[2])
                t = A[2] + B[2]
                D = np.stack([A[0] + c * B[0] - s * B[1], A[1] + s * B[0] + c * B[1], np.arctan2(np.sin(t), np.cos(t))])
                if len(e) > 5 and (e[5] != 1.0 or e[6] != 1.0):     # star-mesh inflation of the deviations about the mean
                    mx, my, mt = D[0].mean(), D[1].mean(), np.arctan2(np.sin(D[2]).sum(), np.cos(D[2]).sum())
                    dt = np.arctan2(np.sin(D[2] - mt), np.cos(D[2] - mt))
                    th = mt + e[6] * dt
                    D = np.stack([mx + e[5] * (D[0] - mx), my + e[5] * (D[1] - my), np.arctan2(np.sin(th), np.cos(th))])
                v[e[2]] = D
            else:
                ref, s = v[e[0]][:, 0], v[e[1]]
                c, sn = np.cos(ref[2]), np.sin(ref[2])
                dx, dy = s[0] - ref[0], s[1] - ref[1]
                lx, ly = c * dx + sn * dy, -sn * dx + c * dy
                if s.shape[0] == 3:
                    dt = s[2] - ref[2]
                    v[e[2]] = np.stack([lx, ly, np.arctan2(np.sin(dt), np.cos(dt))])
                else:
                    v[e[2]] = np.stack([np.arctan2(ly, lx), np.sqrt(lx * lx + ly * ly)])


class OracleTreeBackend:
    def __init__(self, R):
        self.R = R

    def Store(self, universe):
        return OracleTreeStore(self.R, universe)

    def Plan(self, store, spec, share=None, mirror=None):
        return OracleTreePlan(store, spec, share=share, mirror=mirror)

    def BlockOp(self, store, op, entries):
        return OracleTreeBlockOp(store, op, entries)


class OracleTreeScatter:
    """receive side of a level exchange over an OracleTreeStore: slot b (units of `stride` doubles) of the receive buffer -> the block of a
    lifted label"""

    def __init__(self, store, labels, src_blocks, stride=0):
        self.store, self.labels, self.blocks, self.stride = store, list(labels), list(src_blocks), int(stride or 6 * store.N)

    def run(self, src):
        st = self.store
        for l, b in zip(self.labels, self.blocks):
            d = st.fg.variables[l].dim
            st.vals[l] = src[b * self.stride:b * self.stride + d * st.N].numpy().reshape(d, st.N).copy()
