"""Oracle-side restatement of DeviceGraph.solve (test infrastructure): same schedule, same Philox
streams, all compute through oracle/ (CPU)."""
import numpy as np

import oracle as ro


def solve_ref(R, fg, n_sweeps, N, seed=0x524F4D45, solver=1, bandwidth="silverman", product="importance"):
    pk = R.PackedGraph(fg)
    bel2 = pk.beliefs(fg, R.Pose2)
    bell = pk.beliefs(fg, R.Point2) if len(pk.labels[R.Point2]) else np.zeros((0, 2, N))
    factor, dr, fixed, target = R.PackedGraph.conv_table(pk.p2p2)
    F, P = pk.p2p2["F"], pk.prior2["F"]
    mu = np.concatenate([pk.p2p2["mu"].reshape(F, 3), pk.prior2["mu"].reshape(P, 3)])
    cov = np.concatenate([pk.p2p2["cov"].reshape(F, 3, 3), pk.prior2["cov"].reshape(P, 3, 3)])
    L = np.array([ro.cholesky_lower(c) for c in cov])
    factor = np.concatenate([factor, F + np.arange(P)]); dr = np.concatenate([dr, np.full(P, 2)])
    fixed = np.concatenate([fixed, pk.prior2["var"]]); target = np.concatenate([target, pk.prior2["var"]])
    hyp = R.PackedGraph.conv_hypotheses(pk.p2p2)   # multihypo Pose2Pose2 factors: alternative / probability per row + extra rows
    E = 0
    if hyp is not None:
        alt2, w2, ex = hyp
        E = len(ex["factor"])
        target = np.concatenate([target, ex["target"]])
    C2 = 2 * F + P + E
    Fb = pk.br["F"]
    tg2 = np.concatenate([target, pk.br["pose"]]) if Fb else target
    r0 = pk.br["rows0"]; Fb0 = len(r0["factor"])
    tgl = r0["point"] if Fb else np.zeros(0, np.int32)
    Ppt = pk.priorpt2["F"]                                   # landmark priors (PriorPoint2): proposal rows behind the sightings
    if Ppt:
        tgl = np.concatenate([tgl, pk.priorpt2["var"]])
        Lpt = np.array([ro.cholesky_lower(c) for c in pk.priorpt2["cov"].reshape(Ppt, 2, 2)])
    mh = bool((pk.br["alt"] >= 0).any()) if Fb else False

    def csr(tg, nv):
        order = np.argsort(tg, kind="stable").astype(np.int32)
        ptr = np.zeros(nv + 1, dtype=np.int64); np.add.at(ptr, np.asarray(tg, dtype=np.int64) + 1, 1)
        return np.cumsum(ptr).astype(np.int32), order
    ptr2, rows2 = csr(tg2, bel2.shape[0]); ptrl, rowsl = csr(tgl, bell.shape[0])
    S = dict(P2P2=0, BR1=1 << 28, BR0=2 << 28, PROD2=3 << 28, PRODL=4 << 28)
    nh2 = pk.p2p2.get("nh"); nhb = pk.br.get("nh") if Fb else None
    nh2 = nh2 if (nh2 is not None and np.any(nh2 > 0)) else None
    nhb = nhb if (nhb is not None and np.any(nhb > 0)) else None
    for s in range(n_sweeps):
        base = s << 32
        mk = lambda off, nh=0.0: ro.make_opts(N=N, solver=solver, seed=seed, stream_offset=base + off, nullhypo=float(nh))
        prop2 = np.zeros((C2 + Fb, 3, N))
        mhkw = {} if hyp is None else dict(alt_var=alt2, hypo_w=w2)
        if nh2 is None:
            prop2[:2 * F] = ro.conv_pose2pose2(mk(S["P2P2"]), mu, L, bel2, fixed[:2 * F], target[:2 * F], dr[:2 * F], factor=factor[:2 * F], **mhkw)
        else:   # nullhypo= factors: the oracle takes one probability per call -> row by row (Philox stream = row index either way)
            for r in range(2 * F):
                kw = {} if hyp is None else dict(alt_var=alt2[r:r + 1], hypo_w=w2[r:r + 1])
                prop2[r] = ro.conv_pose2pose2(mk(S["P2P2"] + r, nh2[factor[r]]), mu, L, bel2, fixed[r:r + 1], target[r:r + 1], dr[r:r + 1],
                                              factor=factor[r:r + 1], **kw)[0]
        if E:   # the proposals of the second candidates: rows behind the priors, Philox stream = row index
            for e in range(E):
                r = 2 * F + P + e
                prop2[r] = ro.conv_pose2pose2(mk(S["P2P2"] + r, 0.0 if nh2 is None else nh2[ex["factor"][e]]), mu, L, bel2, ex["fixed"][e:e + 1],
                                              ex["target"][e:e + 1], ex["dir"][e:e + 1], factor=ex["factor"][e:e + 1], alt_var=ex["alt"][e:e + 1],
                                              hypo_w=ex["w"][e:e + 1])[0]
        # prior rows: stream id = row index
        for k in range(P):
            o = ro.make_opts(N=N, seed=seed, stream_offset=base + S["P2P2"] + 2 * F + k)
            prop2[2 * F + k] = ro.sample_priorpose2(o, mu[F + k], L[F + k])[0]
        propl = np.zeros((Fb0 + Ppt, 2, N))
        if Ppt:
            propl[Fb0:] = ro.sample_priorpoint2(mk(7 << 28), pk.priorpt2["mu"].reshape(Ppt, 2), Lpt)
        if Fb and nhb is None:
            prop2[C2:] = ro.conv_pose2point2br(mk(S["BR1"]), 1, pk.br["mu"], pk.br["sigma"], bell, bel2, pk.br["point"], pk.br["pose"],
                                               alt_var=pk.br["alt"] if mh else None, hypo_w=pk.br["w"] if mh else None)
            propl[:Fb0] = ro.conv_pose2point2br(mk(S["BR0"]), 0, pk.br["mu"], pk.br["sigma"], bel2, bell, r0["pose"], r0["point"], factor=r0["factor"],
                                             alt_var=r0["alt"] if mh else None, hypo_w=r0["w"] if mh else None)
        elif Fb:
            for r in range(Fb):
                kw = dict(alt_var=pk.br["alt"][r:r + 1], hypo_w=pk.br["w"][r:r + 1]) if mh else {}
                prop2[C2 + r] = ro.conv_pose2point2br(mk(S["BR1"] + r, nhb[r]), 1, pk.br["mu"], pk.br["sigma"], bell, bel2, pk.br["point"][r:r + 1],
                                                      pk.br["pose"][r:r + 1], factor=[r], **kw)[0]
            for r in range(Fb0):
                kw = dict(alt_var=r0["alt"][r:r + 1], hypo_w=r0["w"][r:r + 1]) if mh else {}
                propl[r] = ro.conv_pose2point2br(mk(S["BR0"] + r, nhb[r0["factor"][r]]), 0, pk.br["mu"], pk.br["sigma"], bel2, bell, r0["pose"][r:r + 1],
                                                 r0["point"][r:r + 1], factor=r0["factor"][r:r + 1], **kw)[0]
        lcv = bandwidth == "lcv"
        if product == "gibbs":   # the reference's product: multiscale Gibbs sampling on the manikde! bandwidths of the proposals
            bel2 = ro.product_msgibbs(mk(S["PROD2"]), 3, ptr2, rows2, prop2, ro.kde_bandwidths(prop2, 0b100), bel2, 0b100, 1)
            if Fb or Ppt:
                bell = ro.product_msgibbs(mk(S["PRODL"]), 2, ptrl, rowsl, propl, ro.kde_bandwidths(propl, 0), bell, 0, 1)
            continue
        bel2 = ro.product(mk(S["PROD2"]), 3, ptr2, rows2, prop2, bel2, ro.kde_bandwidths(prop2, 0b100) if lcv else None)
        if Fb or Ppt:
            bell = ro.product(mk(S["PRODL"]), 2, ptrl, rowsl, propl, bell, ro.kde_bandwidths(propl, 0) if lcv else None)
    return bel2, bell


def upsolve_ref(R, fg, frontals, N, seed=0x524F4D45, gibbs_iters=3, product_iters=1, schedule="sequential", solver=1, messages=None,
                groups=None, stream_offset=0, stream_ids=None, up_stream=None, usable=None, meas_vals=None, pairs=None):
    """Oracle-side restatement of rome_clique_upsolve / R.upGibbsCliqueDensity (IIF upGibbsCliqueDensity): same pairs, same row
    tables (multihypo / nullhypo columns included), same Philox streams; every convolution, bandwidth and product through oracle/
    (CPU).  stream_ids: {(factor, target): id within the family} / up_stream: {label: product stream id} (default: row index /
    position within the type, as the library).  -> {label: points (dim, N)}"""
    from rome_jl_amd.clique import CliqueBatch
    usable = usable or fg.isInitialized
    if pairs is None:     # (explicit pairs: a tree level hands over exactly the rows of its plan)
        pairs = []
        for dest in frontals:
            for flabel, labels, _ in fg.factors:
                if dest in labels and all(usable(l) or l in frontals for l in labels if l != dest):
                    pairs.append((flabel, dest))
    batch = CliqueBatch(fg, pairs)
    for l in frontals:
        if l not in batch.vidx:
            t = fg.variables[l]
            batch.vidx[l] = len(batch.vars[t]); batch.vars[t].append(l)
    types = (R.Pose2, R.Point2, R.Pose3)
    bel = {vt: batch.beliefs(vt) for vt in types}
    T = batch.tabs
    rows = {f: np.array(batch.fam_rows[f], dtype=np.int64).reshape(-1, 4) for f in ("p2p2", "br1", "br0", "p3p3", "prpt2")}
    sid = {f: list(range(len(rows[f]))) for f in rows}
    if stream_ids is not None:
        for pair, (fam, r) in batch.rows.items():
            sid[fam][r] = stream_ids[pair]
    mupt = np.array(T["prpt2"]["mu"]).reshape(-1, 2); Lpt = np.array([ro.cholesky_lower(np.asarray(c).reshape(2, 2)) for c in T["prpt2"]["spread"]]).reshape(-1, 3)
    mu2 = np.array(T["p2p2"]["mu"]).reshape(-1, 3); L2 = np.array([ro.cholesky_lower(np.asarray(c).reshape(3, 3)) for c in T["p2p2"]["spread"]]).reshape(-1, 6)
    mub = np.array(T["br"]["mu"]).reshape(-1, 2); sgb = np.array(T["br"]["spread"]).reshape(-1, 2)
    mu3 = np.array(T["p3p3"]["mu"]).reshape(-1, 6); L3 = np.array([ro.cholesky_lower(np.asarray(c).reshape(6, 6)) for c in T["p3p3"]["spread"]]).reshape(-1, 21)
    S = dict(p2p2=0, br1=1 << 28, br0=2 << 28, p3p3=5 << 28, prpt2=7 << 28)
    PROD = {R.Pose2: 3 << 28, R.Point2: 4 << 28, R.Pose3: 6 << 28}
    circ = {R.Pose2: 0b100, R.Point2: 0, R.Pose3: 0}            # product: SE(3) rotations are handled in the chart of each proposal
    circ_bw = {R.Pose2: 0b100, R.Point2: 0, R.Pose3: 0b111000}  # manikde! bandwidth rule: which coordinates are angles
    pos_in_type = {}
    for l in frontals:
        vt = fg.variables[l]
        pos_in_type[l] = sum(1 for m in frontals[:frontals.index(l)] if fg.variables[m] is vt)
    if up_stream is not None:
        pos_in_type = dict(up_stream)

    def proposals_for(targets, base):
        """{label: list of (dim, N) proposals in the device's CSR order (p2p2 rows, br1 rows | br0 rows, messages)}"""
        out = {l: [] for l in targets}
        for fam in ("p2p2", "br1", "br0", "prpt2", "p3p3"):
            rw = rows[fam]
            vt = R.Point2 if fam in ("br0", "prpt2") else (R.Pose3 if fam == "p3p3" else R.Pose2)
            tv = {batch.vidx[l]: l for l in targets if fg.variables[l] is vt}
            sel = [r for r in range(len(rw)) if rw[r, 3] in tv]
            for r in sel:
                alt, hw, nh = batch.fam_hyp[fam][r]
                nz = None
                if batch.fam_meas[fam][r] != -1:   # sampled-measurement factor: z = 0 + I xi with xi = the samples themselves
                    flab = next(p_[0] for p_, (fm, rr) in batch.rows.items() if fm == fam and rr == r)
                    nz = np.asarray(meas_vals[fg.getFactor(flab)[2].meas], dtype=float)[None]
                so = base + S[fam] + sid[fam][r]
                o = ro.make_opts(N=N, solver=solver, seed=seed, stream_offset=so, nullhypo=nh)
                mh = {} if alt < 0 else dict(alt_var=[alt], hypo_w=[hw])
                f, d, fx, tg = rw[r]
                if fam == "p2p2" and d == 2:
                    p = ro.sample_priorpose2(ro.make_opts(N=N, seed=seed, stream_offset=so), mu2[f], L2[f])[0]
                elif fam == "p2p2":
                    p = ro.conv_pose2pose2(o, mu2, L2, bel[R.Pose2], [fx], [tg], [d], factor=[f], noise=nz, **mh)[0]
                elif fam == "prpt2":
                    p = ro.sample_priorpoint2(ro.make_opts(N=N, seed=seed, stream_offset=so), mupt[f], Lpt[f])[0]
                elif fam == "p3p3" and d == 2:
                    p = ro.sample_priorpose3(ro.make_opts(N=N, seed=seed, stream_offset=so), mu3[f], L3[f])[0]
                elif fam == "p3p3":
                    p = ro.conv_pose3pose3(o, mu3, L3, bel[R.Pose3], [fx], [tg], [d], factor=[f])[0]
                elif fam == "br1":
                    p = ro.conv_pose2point2br(o, 1, mub, sgb, bel[R.Point2], bel[R.Pose2], [fx], [tg], factor=[f], noise=nz, **mh)[0]
                else:
                    p = ro.conv_pose2point2br(o, 0, mub, sgb, bel[R.Pose2], bel[R.Point2], [fx], [tg], factor=[f], noise=nz, **mh)[0]
                out[tv[tg]].append(p)
        for l in targets:
            for pts in (messages or {}).get(l, []):
                out[l].append(np.asarray(pts, dtype=float))
        return out

    for it in range(gibbs_iters):
        base = stream_offset + (it << 32)
        if groups is not None:   # update groups: variables of one group together, groups in order
            steps = [[l for l, g in zip(frontals, groups) if g == gg] for gg in sorted(set(groups))]
        else:
            steps = [[l] for l in frontals] if schedule == "sequential" else [list(frontals)]
        for group in steps:
            props = proposals_for(group, base)
            new = {}
            for vt in types:   # one bandwidth call and (when the product streams are consecutive) one product call per variable type: the
                ls = [l for l in group if fg.variables[l] is vt and props[l]]      # oracle's OpenMP loops run over the whole group
                if not ls:
                    continue
                P = np.concatenate([np.stack(props[l]) for l in ls])
                ptr = np.concatenate([[0], np.cumsum([len(props[l]) for l in ls])]).astype(np.int32)
                bw = ro.kde_bandwidths(P, circ_bw[vt])
                pos = [pos_in_type[l] for l in ls]
                B = np.stack([bel[vt][batch.vidx[l]] for l in ls])
                if all(b - a == 1 for a, b in zip(pos, pos[1:])):
                    o = ro.make_opts(N=N, seed=seed, stream_offset=base + PROD[vt] + pos[0])
                    out = ro.product_msgibbs(o, vt.dim, ptr, np.arange(len(P), dtype=np.int32), P, bw, B, circ[vt], product_iters)
                    for k, l in enumerate(ls):
                        new[l] = out[k]
                else:
                    for k, l in enumerate(ls):
                        o = ro.make_opts(N=N, seed=seed, stream_offset=base + PROD[vt] + pos[k])
                        Pk, bk = P[ptr[k]:ptr[k + 1]], bw[ptr[k]:ptr[k + 1]]
                        new[l] = ro.product_msgibbs(o, vt.dim, np.array([0, len(Pk)], dtype=np.int32), np.arange(len(Pk), dtype=np.int32), Pk, bk,
                                                    B[k:k + 1], circ[vt], product_iters)[0]
            for l, b in new.items():
                bel[fg.variables[l]][batch.vidx[l]] = b
    return {l: bel[fg.variables[l]][batch.vidx[l]].copy() for l in frontals}
