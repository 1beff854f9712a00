"""Clique-level batching behind the factor-plugin surface (IIF `proposalbeliefs!` / `predictbelief`).

What `solveTree!` (examples/ManhattanDatasetBatch.jl:43) asks of the convolution path is never ONE convolution: a Gibbs step on a
clique variable calls `proposalbeliefs!(dfg, destlbl, factors, ...)`, i.e. `approxConvBelief` for EVERY factor of the variable, and
a clique sweep does that for every frontal variable.  `proposalbeliefs` below hands the whole batch to the library in one call
(`rome_clique_proposals`): every belief crosses PCIe once, one kernel launch per factor family, one synchronisation -- instead of
three belief blocks and ~45 µs of latency per convolution.  The per-factor path (`approxConv`) and this one draw the same Philox
streams when asked to (stream = family offset + row), so they agree bit for bit (tests/test_gpu_clique.py).
"""
import ctypes as C

import numpy as np

from . import _lib, api
from .factors import (Pose2, Point2, Pose3, Pose2Pose2, PriorPose2, Pose2Point2BearingRange, Pose3Pose3, PriorPose3, PriorPoint2)

FAMILY_STREAM = {"p2p2": 0, "br1": 1 << 28, "br0": 2 << 28, "p3p3": 5 << 28, "prpt2": 7 << 28}   # = DeviceGraph.STREAM_* / rome_clique_proposals


class CliqueHost(C.Structure):
    _fields_ = [("n_pose2", C.c_int32), ("n_point2", C.c_int32), ("n_pose3", C.c_int32), ("reserved0", C.c_int32),
                ("bel_pose2", C.c_void_p), ("bel_point2", C.c_void_p), ("bel_pose3", C.c_void_p),
                ("n_p2p2", C.c_int32), ("f_p2p2", C.c_int32), ("p2p2_rows4", C.c_void_p), ("p2p2_mu", C.c_void_p), ("p2p2_cov", C.c_void_p),
                ("out_p2p2", C.c_void_p),
                ("n_br1", C.c_int32), ("n_br0", C.c_int32), ("f_br", C.c_int32), ("reserved1", C.c_int32),
                ("br1_rows4", C.c_void_p), ("br0_rows4", C.c_void_p), ("br_mu", C.c_void_p), ("br_sigma", C.c_void_p),
                ("out_br1", C.c_void_p), ("out_br0", C.c_void_p),
                ("n_p3p3", C.c_int32), ("f_p3p3", C.c_int32), ("p3p3_rows4", C.c_void_p), ("p3p3_mu", C.c_void_p), ("p3p3_cov", C.c_void_p),
                ("out_p3p3", C.c_void_p),
                ("n_prpt2", C.c_int32), ("f_prpt2", C.c_int32), ("prpt2_rows4", C.c_void_p), ("prpt2_mu", C.c_void_p), ("prpt2_cov", C.c_void_p),
                ("out_prpt2", C.c_void_p),
                ("p2p2_alt", C.c_void_p), ("p2p2_hypo_w", C.c_void_p), ("p2p2_nullhypo", C.c_void_p),
                ("br1_alt", C.c_void_p), ("br1_hypo_w", C.c_void_p), ("br1_nullhypo", C.c_void_p),
                ("br0_alt", C.c_void_p), ("br0_hypo_w", C.c_void_p), ("br0_nullhypo", C.c_void_p),
                ("p3p3_nullhypo", C.c_void_p),
                ("p2p2_stream", C.c_void_p), ("br1_stream", C.c_void_p), ("br0_stream", C.c_void_p), ("p3p3_stream", C.c_void_p),
                ("prpt2_stream", C.c_void_p), ("p2p2_meas", C.c_void_p), ("br1_meas", C.c_void_p), ("br0_meas", C.c_void_p)]


class SampledPose2Pose2:
    """A Pose2Pose2 factor whose measurement distribution is a set of N SAMPLES held in a belief block of a device-resident store (IIF
    accepts any SamplableBelief as `Z`): `meas` = label of the Pose2 block with the tangent coordinates (x, y, theta).  This is what the
    relative up-message of a child clique is to its parent (tree.TreeSolver, messages="relative"); only plans over a store take it."""
    variable_types = (Pose2, Pose2)

    def __init__(self, meas):
        self.meas = meas


class SampledBearingRange:
    """The same for a pose -> landmark relation: `meas` = label of a POINT2 block whose N (x, y) entries are (bearing, range) samples."""
    variable_types = (Pose2, Point2)

    def __init__(self, meas):
        self.meas = meas


class CliqueBatch:
    """The (factor, target) pairs of a clique as row tables over the clique's own variables.

    pairs: iterable of (factor_label, target_label).  Row r of a family is the r-th pair of that family in the order given;
    `rows[(flabel, target)] = (family, r)`.
    `multihypo=[1, w, 1-w]` factors (test/testMultimodalRangeBearing.jl:53) and `nullhypo=p` factors (test/testPose3Pose3NH.jl:118)
    travel as per-row columns (`<fam>_alt / _hypo_w / _nullhypo` of rome_clique_host) -- IIF resolves them inside the same
    proposalbeliefs! / upGibbsCliqueDensity loop.
    var_index: {label: index within its type} of a device-resident store (DeviceStore.index): the row tables then address the store
               instead of clique-local arrays (UpsolvePlan).
    stream_ids: {(factor_label, target_label): Philox stream id of the row within its family}: a clique that is a subset of a larger
               table draws what the whole table draws (partition-independent results); default: the row's index."""

    def __init__(self, fg, pairs, var_index=None, stream_ids=None):
        self.fg, self.N = fg, fg.N
        self.vars = {Pose2: [], Point2: [], Pose3: []}
        self.vidx = {}
        self.rows = {}
        self.var_index = var_index
        tabs = {k: dict(rows4=[], fidx={}, mu=[], spread=[]) for k in ("p2p2", "br", "p3p3", "prpt2")}
        self.fam_rows = {"p2p2": [], "br1": [], "br0": [], "p3p3": [], "prpt2": []}
        self.fam_hyp = {k: [] for k in self.fam_rows}      # per row: (alt, hypo_w, nullhypo)
        self.fam_sid = {k: [] for k in self.fam_rows}      # per row: Philox stream id (or None)
        self.fam_meas = {k: [] for k in self.fam_rows}     # per row: store block of its measurement samples, -1 = ordinary
        nullh = getattr(fg, "nullhypo", {})

        def var(l):
            if var_index is not None:
                if l not in self.vidx:
                    self.vidx[l] = var_index[l]; self.vars[fg.variables[l]].append(l)
                return self.vidx[l]
            if l not in self.vidx:
                t = fg.variables[l]
                self.vidx[l] = len(self.vars[t]); self.vars[t].append(l)
            return self.vidx[l]

        def fac(kind, flabel, mu, spread):
            tb = tabs[kind]
            if flabel not in tb["fidx"]:
                tb["fidx"][flabel] = len(tb["mu"]); tb["mu"].append(np.asarray(mu, float)); tb["spread"].append(np.asarray(spread, float))
            return tb["fidx"][flabel]

        for flabel, target in pairs:
            _, labels, f = fg.getFactor(flabel)
            if target not in labels:
                raise KeyError("%s is not connected to factor %s" % (target, flabel))
            mh = fg.multihypo.get(flabel)
            alt, hw = -1, 1.0
            if mh is not None:   # [first, cand1, cand2] with P(cand1) = mh[0], P(cand2) = mh[1]
                first, c1, c2 = labels
                if target == first:
                    d, other, alt, hw = 1, c1, var(c2), mh[0]
                else:
                    own, oth, hw = (c1, c2, mh[0]) if target == c1 else (c2, c1, mh[1])
                    d, other, alt = 0, first, var(oth)
            meas = -1
            if isinstance(f, SampledPose2Pose2):
                if mh is not None:
                    raise TypeError("a sampled-measurement factor carries no multihypo")
                d = 0 if labels[1] == target else 1
                other = labels[0] if d == 0 else labels[1]
                fam, meas = "p2p2", (var_index[f.meas] if var_index is not None else -2)   # (-2: no store -- only the oracle-side restatement)
                row = (fac(fam, "<sampled>", np.zeros(3), np.eye(3)), d, var(other), var(target))
            elif isinstance(f, SampledBearingRange):
                if mh is not None:
                    raise TypeError("a sampled-measurement factor carries no multihypo")
                d = 0 if labels[1] == target else 1
                other = labels[0] if d == 0 else labels[1]
                fam, meas = ("br0" if d == 0 else "br1"), (var_index[f.meas] if var_index is not None else -2)
                row = (fac("br", "<sampled>", [0.0, 0.0], [1.0, 1.0]), d, var(other), var(target))
            elif isinstance(f, (Pose2Pose2, Pose3Pose3)):
                fam = "p2p2" if isinstance(f, Pose2Pose2) else "p3p3"
                if mh is None:
                    d = 0 if labels[1] == target else 1
                    other = labels[0] if d == 0 else labels[1]
                row = (fac(fam, flabel, f.Z.mu, f.Z.cov), d, var(other), var(target))
            elif isinstance(f, (PriorPose2, PriorPose3)):
                fam = "p2p2" if isinstance(f, PriorPose2) else "p3p3"
                row = (fac(fam, flabel, f.Z.mu, f.Z.cov), 2, var(target), var(target))
            elif isinstance(f, PriorPoint2):   # landmark prior: its samples are the proposal
                fam = "prpt2"
                row = (fac(fam, flabel, f.Z.mu, f.Z.cov), 2, var(target), var(target))
            elif isinstance(f, Pose2Point2BearingRange):
                if mh is None:
                    d = 0 if labels[1] == target else 1
                    other = labels[0] if d == 0 else labels[1]
                fam = "br0" if d == 0 else "br1"
                row = (fac("br", flabel, [f.bearing.mu, f.range.mu], [f.bearing.sigma, f.range.sigma]), d, var(other), var(target))
            else:
                raise TypeError("factor type %s is outside the hot path" % type(f).__name__)
            self.rows[(flabel, target)] = (fam, len(self.fam_rows[fam]))
            self.fam_rows[fam].append(row)
            self.fam_hyp[fam].append((alt, hw, float(nullh.get(flabel, 0.0)) if row[1] != 2 else 0.0))
            self.fam_sid[fam].append(None if stream_ids is None else stream_ids[(flabel, target)])
            self.fam_meas[fam].append(meas)
        self.tabs = tabs

    def beliefs(self, vt):
        """(n, dim, N) belief blocks of the clique's variables of one type; uninitialised variables start at zero
        (approxConv's convention: the start point of a root-find with a unique root does not matter)."""
        ls = self.vars[vt]
        out = np.zeros((len(ls), vt.dim, self.N))
        for k, l in enumerate(ls):
            if self.fg.isInitialized(l):
                out[k] = self.fg.getVal(l)
        return out

    def _fill(self, q, keep, with_out=True):
        """row tables, factor tables and beliefs of the clique -> the fields of a rome_clique_host"""
        def ptr(a, dt=np.float64):
            a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
            return a.ctypes.data_as(C.c_void_p) if a.size else None

        if self.var_index is None:   # clique-local arrays: the beliefs travel with the call
            bel = {vt: self.beliefs(vt) for vt in (Pose2, Point2, Pose3)}
            q.n_pose2, q.n_point2, q.n_pose3 = (len(self.vars[t]) for t in (Pose2, Point2, Pose3))
            q.bel_pose2, q.bel_point2, q.bel_pose3 = ptr(bel[Pose2]), ptr(bel[Point2]), ptr(bel[Pose3])
        else:                        # the tables address a device-resident store (UpsolvePlan)
            q.n_pose2 = q.n_point2 = q.n_pose3 = 0
        out = {}
        for fam, dt in (("p2p2", 3), ("br1", 3), ("br0", 2), ("p3p3", 6), ("prpt2", 2)):
            out[fam] = np.zeros((len(self.fam_rows[fam]) if with_out else 0, dt, self.N))
        t = self.tabs
        q.n_p2p2, q.f_p2p2 = len(self.fam_rows["p2p2"]), len(t["p2p2"]["mu"])
        q.p2p2_rows4 = ptr(np.array(self.fam_rows["p2p2"], dtype=np.int32).reshape(-1, 4), np.int32)
        q.p2p2_mu, q.p2p2_cov = ptr(np.array(t["p2p2"]["mu"]).reshape(-1, 3)), ptr(np.array(t["p2p2"]["spread"]).reshape(-1, 9))
        q.n_br1, q.n_br0, q.f_br = len(self.fam_rows["br1"]), len(self.fam_rows["br0"]), len(t["br"]["mu"])
        q.br1_rows4 = ptr(np.array(self.fam_rows["br1"], dtype=np.int32).reshape(-1, 4), np.int32)
        q.br0_rows4 = ptr(np.array(self.fam_rows["br0"], dtype=np.int32).reshape(-1, 4), np.int32)
        q.br_mu, q.br_sigma = ptr(np.array(t["br"]["mu"]).reshape(-1, 2)), ptr(np.array(t["br"]["spread"]).reshape(-1, 2))
        q.n_p3p3, q.f_p3p3 = len(self.fam_rows["p3p3"]), len(t["p3p3"]["mu"])
        q.p3p3_rows4 = ptr(np.array(self.fam_rows["p3p3"], dtype=np.int32).reshape(-1, 4), np.int32)
        q.p3p3_mu, q.p3p3_cov = ptr(np.array(t["p3p3"]["mu"]).reshape(-1, 6)), ptr(np.array(t["p3p3"]["spread"]).reshape(-1, 36))
        q.n_prpt2, q.f_prpt2 = len(self.fam_rows["prpt2"]), len(t["prpt2"]["mu"])
        q.prpt2_rows4 = ptr(np.array(self.fam_rows["prpt2"], dtype=np.int32).reshape(-1, 4), np.int32)
        q.prpt2_mu, q.prpt2_cov = ptr(np.array(t["prpt2"]["mu"]).reshape(-1, 2)), ptr(np.array(t["prpt2"]["spread"]).reshape(-1, 4))
        q.out_p2p2, q.out_br1, q.out_br0, q.out_p3p3, q.out_prpt2 = (out[f].ctypes.data_as(C.c_void_p) if out[f].size else None
                                                                    for f in ("p2p2", "br1", "br0", "p3p3", "prpt2"))
        for fam in ("p2p2", "br1", "br0"):
            if any(m == -2 for m in self.fam_meas[fam]):
                raise TypeError("a sampled-measurement factor needs a device-resident store (UpsolvePlan / tree.TreeLevelPlan)")
            if any(m >= 0 for m in self.fam_meas[fam]):
                setattr(q, fam + "_meas", ptr(np.array(self.fam_meas[fam], dtype=np.int32), np.int32))
        # hypothesis / stream-id columns, only where a row of the family carries one
        for fam in ("p2p2", "br1", "br0", "p3p3", "prpt2"):
            hyp, sid = self.fam_hyp[fam], self.fam_sid[fam]
            if fam in ("p2p2", "br1", "br0") and any(h[0] >= 0 for h in hyp):
                setattr(q, fam + "_alt", ptr(np.array([h[0] for h in hyp], dtype=np.int32), np.int32))
                setattr(q, fam + "_hypo_w", ptr(np.array([h[1] for h in hyp], dtype=np.float64)))
            if fam != "prpt2" and any(h[2] > 0 for h in hyp):
                setattr(q, fam + "_nullhypo", ptr(np.array([h[2] for h in hyp], dtype=np.float64)))
            if sid and any(x is not None for x in sid):
                if any(x is None for x in sid):
                    raise ValueError("stream ids must be given for every row of a family or for none")
                setattr(q, fam + "_stream", ptr(np.array(sid, dtype=np.int32), np.int32))
        return out

    def run(self, opts, ctx=None):
        """-> {family: (rows, dt, N) proposals} through rome_clique_proposals (ONE call)."""
        ctx = ctx or api.default_context()
        q = CliqueHost()
        keep = []
        out = self._fill(q, keep)
        o = _lib.Opts.from_buffer_copy(opts)
        o.layout = _lib.LAYOUT_SOA
        _lib.check(_lib.load().rome_clique_proposals(ctx.handle, C.byref(o), C.byref(q)), ctx.handle)
        return out

    def _fill_upsolve(self, u, keep, up_labels, gibbs_iters, product_iters, schedule, messages, groups, up_stream=None, up_mirror=None,
                      outputs=True):
        """the fields of a rome_clique_upsolve_host -> {vartype: (labels, new, bw)} result arrays (None without `outputs`)"""
        self._fill(u.clique, keep, with_out=False)
        types = (Pose2, Point2, Pose3)
        u.gibbs_iters, u.product_iters = int(gibbs_iters), int(product_iters)
        u.schedule = {"sequential": 0, "jacobi": 1}[schedule]
        u.n_up = len(up_labels)
        upt = np.array([types.index(self.fg.variables[l]) for l in up_labels], dtype=np.int32)
        upv = np.array([self.vidx[l] for l in up_labels], dtype=np.int32)
        keep += [upt, upv]
        u.up_type, u.up_var = upt.ctypes.data_as(C.c_void_p), upv.ctypes.data_as(C.c_void_p)
        for name, arr in (("up_group", groups), ("up_stream", up_stream), ("up_mirror", up_mirror)):
            if arr is not None:   # groups: variables of one group are updated together, groups in order (a frontier of independent cliques)
                a = np.ascontiguousarray(arr, dtype=np.int32); keep.append(a)
                if len(a) != len(up_labels):
                    raise ValueError("%s needs one entry per updated variable" % name)
                setattr(u, name, a.ctypes.data_as(C.c_void_p))
        names = ("pose2", "point2", "pose3")
        res = {}
        pos_of = {l: k for k, l in enumerate(up_labels)}
        for ti, vt in enumerate(types):
            ls = [l for l in up_labels if self.fg.variables[l] is vt]
            if outputs:
                new = np.zeros((len(ls), vt.dim, self.N)); bw = np.zeros((len(ls), vt.dim))
                keep += [new, bw]
                res[vt] = (ls, new, bw)
                setattr(u, "new_" + names[ti], new.ctypes.data_as(C.c_void_p) if new.size else None)
                setattr(u, "bw_" + names[ti], bw.ctypes.data_as(C.c_void_p) if bw.size else None)
            msgs, pos = [], []
            for l, plist in (messages or {}).items():
                if self.fg.variables[l] is vt:
                    for pts in plist:
                        msgs.append(np.asarray(pts, dtype=float)); pos.append(pos_of[l])
            setattr(u, "n_msg_" + names[ti], len(msgs))
            if msgs:
                m = np.ascontiguousarray(np.stack(msgs)); pp = np.array(pos, dtype=np.int32)
                keep += [m, pp]
                setattr(u, "msg_" + names[ti], m.ctypes.data_as(C.c_void_p)); setattr(u, "msg_" + names[ti] + "_up", pp.ctypes.data_as(C.c_void_p))
        return res

    def upsolve(self, opts, up_labels, gibbs_iters=3, product_iters=1, schedule="sequential", messages=None, ctx=None, groups=None,
                up_stream=None):
        """IIF `upGibbsCliqueDensity` on the device in ONE call (rome_clique_upsolve): gibbs_iters x {proposals of every
        (factor, target) pair, manikde! bandwidths, multiscale Gibbs product, write-back} for the variables `up_labels` (the
        clique's frontals, in Gibbs order; the pairs of this batch must be grouped by target in that order).
        messages: {label: [(dim, N) points, ...]} upward messages of child cliques on updated variables.
        -> {label: (points (dim, N), bandwidths (dim,))}"""
        ctx = ctx or api.default_context()
        u = CliqueUpsolveHost()
        keep = []
        res = self._fill_upsolve(u, keep, list(up_labels), gibbs_iters, product_iters, schedule, messages, groups, up_stream)
        o = _lib.Opts.from_buffer_copy(opts)
        o.layout = _lib.LAYOUT_SOA
        _lib.check(_lib.load().rome_clique_upsolve(ctx.handle, C.byref(o), C.byref(u)), ctx.handle)
        return {l: (new[k].copy(), bw[k].copy()) for vt, (ls, new, bw) in res.items() for k, l in enumerate(ls)}


class CliqueUpsolveHost(C.Structure):
    _fields_ = [("clique", CliqueHost), ("gibbs_iters", C.c_int32), ("product_iters", C.c_int32), ("schedule", C.c_int32), ("n_up", C.c_int32),
                ("up_type", C.c_void_p), ("up_var", C.c_void_p),
                ("n_msg_pose2", C.c_int32), ("n_msg_point2", C.c_int32), ("n_msg_pose3", C.c_int32), ("reserved0", C.c_int32),
                ("msg_pose2", C.c_void_p), ("msg_pose2_up", C.c_void_p), ("msg_point2", C.c_void_p), ("msg_point2_up", C.c_void_p),
                ("msg_pose3", C.c_void_p), ("msg_pose3_up", C.c_void_p),
                ("new_pose2", C.c_void_p), ("bw_pose2", C.c_void_p), ("new_point2", C.c_void_p), ("bw_point2", C.c_void_p),
                ("new_pose3", C.c_void_p), ("bw_pose3", C.c_void_p), ("up_group", C.c_void_p), ("up_stream", C.c_void_p),
                ("up_mirror", C.c_void_p),
                ("n_smsg_pose2", C.c_int32), ("n_smsg_point2", C.c_int32), ("n_smsg_pose3", C.c_int32), ("reserved1", C.c_int32),
                ("smsg_pose2_src", C.c_void_p), ("smsg_pose2_up", C.c_void_p), ("smsg_point2_src", C.c_void_p), ("smsg_point2_up", C.c_void_p),
                ("smsg_pose3_src", C.c_void_p), ("smsg_pose3_up", C.c_void_p)]


# ------------------------------------------------------------------------------------------ device-resident store + plans
class DeviceStore:
    """The beliefs of a whole graph resident in HBM across clique up-solves (rome_store): [n][dim][N] SoA blocks per variable type,
    variables numbered per type in the graph's insertion order (`index[label]`).  `wrap=`: {vartype: torch tensor} of caller-owned
    device memory (e.g. DeviceGraph.bel) instead of an allocation of the library's own."""
    TYPES = (Pose2, Point2, Pose3)

    def __init__(self, fg, ctx=None, wrap=None, upload=True):
        self.ctx = ctx or api.default_context()
        self._lib = _lib.load()
        self.fg, self.N = fg, fg.N
        self.labels = {vt: [] for vt in self.TYPES}
        self.index = {}
        self.touched = set()                 # variables whose block holds a belief (uploaded, or written by a plan): what download() copies back
        for l, t in fg.variables.items():
            self.index[l] = len(self.labels[t]); self.labels[t].append(l)
        h = C.c_void_p()
        n = [len(self.labels[vt]) for vt in self.TYPES]
        if wrap is None:
            _lib.check(self._lib.rome_store_create(self.ctx.handle, self.N, n[0], n[1], n[2], C.byref(h)), self.ctx.handle)
        else:
            self._keep = wrap
            self.touched.update(fg.variables)        # caller-owned tensors already hold beliefs: download() returns all of them
            ptrs = [C.c_void_p(wrap[vt].data_ptr()) if n[k] else None for k, vt in enumerate(self.TYPES)]
            _lib.check(self._lib.rome_store_wrap(self.ctx.handle, self.N, n[0], ptrs[0], n[1], ptrs[1], n[2], ptrs[2], C.byref(h)), self.ctx.handle)
        self.handle = h
        if upload and wrap is None:
            self.upload(fg)

    def upload(self, fg, labels=None):
        """beliefs of `labels` (default: every initialised variable) host -> store"""
        for ti, vt in enumerate(self.TYPES):
            ls = self.labels[vt]
            want = [k for k, l in enumerate(ls) if fg.isInitialized(l) and (labels is None or l in labels)]
            k = 0
            while k < len(want):   # runs of consecutive indices: one copy each
                j = k
                while j + 1 < len(want) and want[j + 1] == want[j] + 1:
                    j += 1
                self.touched.update(ls[i] for i in want[k:j + 1])
                blk = np.ascontiguousarray(np.stack([fg.getVal(ls[i]) for i in want[k:j + 1]]), dtype=np.float64)
                _lib.check(self._lib.rome_store_upload(self.handle, _lib.LAYOUT_SOA, ti, want[k], j + 1 - k,
                                                       blk.ctypes.data_as(C.POINTER(C.c_double))), self.ctx.handle)
                k = j + 1

    def put(self, label, pts):
        """(dim, N) belief of one variable, host -> store"""
        vt = self.fg.variables[label]
        self.touched.add(label)
        blk = np.ascontiguousarray(np.asarray(pts, dtype=np.float64).reshape(1, vt.dim, self.N))
        _lib.check(self._lib.rome_store_upload(self.handle, _lib.LAYOUT_SOA, self.TYPES.index(vt), self.index[label], 1,
                                               blk.ctypes.data_as(C.POINTER(C.c_double))), self.ctx.handle)

    def get(self, label):
        """(dim, N) belief of one variable, store -> host"""
        vt = self.fg.variables[label]
        out = np.zeros((1, vt.dim, self.N))
        _lib.check(self._lib.rome_store_download(self.handle, _lib.LAYOUT_SOA, self.TYPES.index(vt), self.index[label], 1,
                                                 out.ctypes.data_as(C.POINTER(C.c_double))), self.ctx.handle)
        return out[0]

    def download(self, fg=None, labels=None):
        """store -> fg.vals: `labels`, or (default) every variable whose block HOLDS a belief -- uploaded, or written by a plan over this
        store.  A block that was never written is zero-initialised memory, not a belief: copying it back would make the variable count as
        initialised (FactorGraph.isInitialized) with all-zero particles."""
        fg = fg or self.fg
        labels = self.touched if labels is None else set(labels)
        for ti, vt in enumerate(self.TYPES):
            ls = self.labels[vt]
            want = [k for k, l in enumerate(ls) if l in labels and l in fg.variables]
            if not want:
                continue
            lo, hi = want[0], want[-1] + 1          # only the index range that holds requested blocks crosses PCIe
            out = np.zeros((hi - lo, vt.dim, self.N))
            _lib.check(self._lib.rome_store_download(self.handle, _lib.LAYOUT_SOA, ti, lo, hi - lo, out.ctypes.data_as(C.POINTER(C.c_double))),
                       self.ctx.handle)
            for k in want:
                fg.vals[ls[k]] = out[k - lo].copy()

    def device_ptr(self, vt):
        p, n = C.c_void_p(), C.c_int32()
        _lib.check(self._lib.rome_store_ptr(self.handle, self.TYPES.index(vt), C.byref(p), C.byref(n)), self.ctx.handle)
        return p.value, n.value

    def close(self):
        if getattr(self, "handle", None):
            self._lib.rome_store_destroy(self.handle); self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frontier_order(cliques):
    """update order and groups of a frontier of independent cliques: group g = the g-th frontal of every clique"""
    order, groups = [], []
    for g in range(max((len(c) for c in cliques), default=0)):
        for c in cliques:
            if len(c) > g:
                order.append(c[g]); groups.append(g)
    return order, groups


def frontier_pairs(fg, cliques, order, usable=None):
    """(factor, destination) pairs of a frontier in update order; raises when the cliques are not independent (a factor links a
    frontal of one clique to a frontal of another).  usable(label): does the variable have a belief to convolve from (default:
    fg.isInitialized)"""
    allf = [l for c in cliques for l in c]
    if len(set(allf)) != len(allf):
        raise ValueError("a variable is frontal in two cliques of the frontier")
    fset = set(allf)
    owner = {l: k for k, c in enumerate(cliques) for l in c}
    usable = usable or fg.isInitialized
    by_var = {}
    for flabel, labels, _ in fg.factors:            # one pass over the factors (a frontier has thousands of destinations)
        for l in labels:
            if l in fset:
                by_var.setdefault(l, []).append((flabel, labels))
    pairs = []
    for dest in order:
        for flabel, labels in by_var.get(dest, ()):
            others = [l for l in labels if l != dest]
            if any(l in fset and owner[l] != owner[dest] for l in others):
                raise ValueError("cliques of the frontier are not independent: %s links %s and a frontal of another clique" % (flabel, dest))
            if flabel in fg.multihypo and dest != labels[0]:
                others = [labels[0]]       # a candidate of a multihypo factor needs the certain variable only (schedule.init_rounds)
            if all(usable(l) or l in fset for l in others):
                pairs.append((flabel, dest))
    return pairs


def plan_frontier(fg, cliques, share=None, usable=None, var_index=None):
    """The host-side plan of a frontier up-solve, for the whole frontier or one rank's share of it:
      order / groups   this share's update list (group g = the g-th frontal of every clique) and `order_all` of the whole frontier
      pairs            this share's (factor, destination) pairs in update order
      stream_ids       {(factor, destination): row index within its family in the WHOLE frontier's tables}
      up_stream        {label: position among the updated variables of its type in the WHOLE frontier}
    Philox streams are positions in the whole frontier, so the shares of a frontier together draw exactly what the unsharded call
    draws (partition-independent results).  Independence is checked over the whole frontier on every rank."""
    cliques = [list(c) for c in cliques]
    order_all, groups_all = frontier_order(cliques)
    pairs_all = frontier_pairs(fg, cliques, order_all, usable)
    full = CliqueBatch(fg, pairs_all, var_index=var_index)
    sid = {pair: r for pair, (fam, r) in full.rows.items()}
    pos_t, cnt = {}, {Pose2: 0, Point2: 0, Pose3: 0}
    for l in order_all:
        vt = fg.variables[l]; pos_t[l] = cnt[vt]; cnt[vt] += 1
    mine = set(order_all) if share is None else {l for k in share for l in cliques[k]}
    return dict(order=[l for l in order_all if l in mine], groups=[g for l, g in zip(order_all, groups_all) if l in mine],
                pairs=[p for p in pairs_all if p[1] in mine], stream_ids=sid, up_stream=pos_t, order_all=order_all)


class UpsolvePlan:
    """One clique / frontier up-solve bound to a DeviceStore (rome_upsolve_plan): the validated row tables live on the device, `run`
    issues only kernel launches on the context's stream -- beliefs never leave HBM.  `cliques`: list of frontal lists (one clique, or
    a frontier of independent cliques: group g = the g-th frontals).
    share: indices of the cliques THIS plan updates (a rank's share of the frontier; default all): see plan_frontier.
    mirror: {label: block} of a device send buffer that the new belief of `label` is also written to by the product kernel."""

    def __init__(self, store, cliques, share=None, gibbsIters=3, Niter=1, mirror=None, usable=None, outputs=False):
        self.store, fg = store, store.fg
        self.ctx, self._lib = store.ctx, _lib.load()
        fp = plan_frontier(fg, cliques, share, usable, var_index=store.index)
        order = fp["order"]
        self.batch = CliqueBatch(fg, fp["pairs"], var_index=store.index, stream_ids=fp["stream_ids"])
        for l in order:
            if l not in self.batch.vidx:
                self.batch.vidx[l] = store.index[l]
        self.order, self.order_all = order, fp["order_all"]
        u = CliqueUpsolveHost()
        keep = []
        self.res = self.batch._fill_upsolve(u, keep, order, gibbsIters, Niter, "sequential", None, fp["groups"],
                                            up_stream=[fp["up_stream"][l] for l in order],
                                            up_mirror=None if mirror is None else [mirror.get(l, -1) for l in order], outputs=outputs)
        self._keep = keep
        self.has_mirror = mirror is not None
        o = api.make_opts(N=fg.N)
        o.layout = _lib.LAYOUT_SOA
        h = C.c_void_p()
        _lib.check(self._lib.rome_upsolve_plan_create(self.ctx.handle, store.handle, C.byref(o), C.byref(u), C.byref(h)), self.ctx.handle)
        self.handle = h

    def run(self, opts, mirror_out=None, mirror_stride=0):
        """gibbsIters x {proposals -> manikde! -> product -> in-place write}; asynchronous unless the plan was created with outputs.
        mirror_out: the send buffer -- a device pointer (int) or a tensor (its data_ptr())."""
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


class ScatterPlan:
    """The receive side of a frontier exchange (rome_scatter_plan): block src_block[k] of a device buffer becomes the belief of
    labels[k] in the store -- one launch, index lists uploaded once."""

    def __init__(self, store, labels, src_blocks, stride=0):
        self.store, self.ctx, self._lib = store, store.ctx, _lib.load()
        self.labels = list(labels)
        fg = store.fg
        ty = np.array([DeviceStore.TYPES.index(fg.variables[l]) for l in labels], dtype=np.int32)
        va = np.array([store.index[l] for l in labels], dtype=np.int32)
        sb = np.ascontiguousarray(src_blocks, dtype=np.int32)
        h = C.c_void_p()
        PI = C.POINTER(C.c_int32)
        _lib.check(self._lib.rome_scatter_plan_create(self.ctx.handle, store.handle, len(labels), ty.ctypes.data_as(PI), va.ctypes.data_as(PI),
                                                      sb.ctypes.data_as(PI), int(stride), C.byref(h)), self.ctx.handle)
        self.handle = h

    def run(self, src_dev):
        """src_dev: device pointer (int) or tensor of the receive buffer"""
        if hasattr(src_dev, "data_ptr"):
            src_dev = src_dev.data_ptr()
        _lib.check(self._lib.rome_scatter_plan_run(self.handle, C.c_void_p(src_dev)), self.ctx.handle)
        self.store.touched.update(self.labels)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.rome_scatter_plan_destroy(self.handle); self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def upGibbsCliqueFrontier(fg, cliques, gibbsIters=3, Niter=1, solver=_lib.SOLVER_NEWTON, seed=None, ctx=None, setvals=True, **optkw):
    """A FRONTIER of independent cliques through ONE `rome_clique_upsolve` call: `cliques` = list of frontal lists (Gibbs
    order inside each); the g-th frontals of all cliques form update group g, so every launch covers the whole frontier.  The cliques
    must be independent: a frontal of one may appear in another's factors only as a fixed (separator) variable that is not itself a
    frontal of the frontier.  Factors with `multihypo` / `nullhypo` are part of the batch.  -> {frontal: (points, bandwidths)}"""
    order, groups = frontier_order(cliques)
    pairs = frontier_pairs(fg, cliques, order)
    batch = CliqueBatch(fg, pairs)
    for l in order:
        if l not in batch.vidx:
            t = fg.variables[l]
            batch.vidx[l] = len(batch.vars[t]); batch.vars[t].append(l)
    opts = api.make_opts(N=fg.N, solver=solver, seed=seed, **optkw)
    res = batch.upsolve(opts, order, gibbs_iters=gibbsIters, product_iters=Niter, groups=groups, ctx=ctx)
    if setvals:
        for l, (pts, _) in res.items():
            fg.initVariable(l, pts)
    return res


def upGibbsCliqueDensity(fg, frontals, factor_labels=None, gibbsIters=3, Niter=1, schedule="sequential", messages=None,
                         solver=_lib.SOLVER_NEWTON, seed=None, ctx=None, setvals=True, **optkw):
    """IIF `upGibbsCliqueDensity` for one clique: Gibbs over the `frontals` (in order) with the factors `factor_labels` (default:
    every factor of a frontal whose other variable has a belief), all on the device in one library call
    (`rome_clique_upsolve`).  With `setvals` the new beliefs are stored in `fg` (IIF setValKDE!).
    -> {frontal: (points (dim, N), manikde! bandwidths (dim,))}"""
    pairs = []
    for dest in frontals:
        for flabel, labels, _ in fg.factors:
            if dest not in labels or (factor_labels is not None and flabel not in factor_labels):
                continue
            if not all(fg.isInitialized(l) or l in frontals for l in labels if l != dest):
                continue
            pairs.append((flabel, dest))
    batch = CliqueBatch(fg, pairs)
    for l in frontals:   # a frontal without any usable factor still needs its slot in the clique's arrays
        if l not in batch.vidx:
            t = fg.variables[l]
            batch.vidx[l] = len(batch.vars[t]); batch.vars[t].append(l)
    opts = api.make_opts(N=fg.N, solver=solver, seed=seed, **optkw)
    res = batch.upsolve(opts, list(frontals), gibbs_iters=gibbsIters, product_iters=Niter, schedule=schedule, messages=messages, ctx=ctx)
    if setvals:
        for l, (pts, _) in res.items():
            fg.initVariable(l, pts)
    return res


def proposalbeliefs(fg, destlabels, factor_labels=None, solver=_lib.SOLVER_NEWTON, seed=None, ctx=None, **optkw):
    """IIF `proposalbeliefs!` for one or several destination variables in ONE library call: the proposal of every factor of
    every destination (or of `factor_labels` only) -> {(factor_label, dest_label): (dim, N) proposal points}.
    Row r of a factor family draws Philox stream `stream_offset + FAMILY_STREAM[family] + r` (rows in destination, then factor
    order), which is what `approxConv(fg, flabel, dest, stream_offset=...)` draws when given that stream."""
    if isinstance(destlabels, str):
        destlabels = [destlabels]
    pairs = []
    for dest in destlabels:
        for flabel, labels, _ in fg.factors:
            if dest in labels and (factor_labels is None or flabel in factor_labels):
                pairs.append((flabel, dest))
    batch = CliqueBatch(fg, pairs)
    opts = api.make_opts(N=fg.N, solver=solver, seed=seed, **optkw)
    out = batch.run(opts, ctx)
    return {pair: out[fam][r] for pair, (fam, r) in batch.rows.items()}, batch


def predictbelief(fg, destlabel, factor_labels=None, solver=_lib.SOLVER_NEWTON, seed=None, ctx=None, Niter=1, **optkw):
    """IIF `predictbelief(dfg, destlbl, factors; N)`: the proposals of every usable factor of `destlabel` (one library call,
    `proposalbeliefs`) multiplied by `manifoldProduct` (multiscale Gibbs product on the `manikde!` bandwidths, one more call) ->
    (dim, N) points of the predicted belief.  A factor is usable when its other variable has a belief (priors always are).
    Multihypo factors take the per-factor path (`approxConv`, which routes them to the `*_mh` entry points); their proposals are
    stacked with the others before the product."""
    from .convolution import approxConv
    usable, mh = [], []
    for flabel, labels, f in fg.factors:
        if destlabel not in labels or (factor_labels is not None and flabel not in factor_labels):
            continue
        if not all(fg.isInitialized(l) for l in labels if l != destlabel):
            continue
        (mh if fg.multihypo.get(flabel) is not None else usable).append(flabel)
    if not usable and not mh:
        raise ValueError("predictbelief: no factor of %s has initialised neighbours" % destlabel)
    plist = []
    if usable:
        props, _ = proposalbeliefs(fg, destlabel, factor_labels=usable, solver=solver, seed=seed, ctx=ctx, **optkw)
        plist += [props[(fl, destlabel)] for fl in usable]
    for fl in mh:
        plist.append(np.asarray(approxConv(fg, fl, destlabel, solver=solver, seed=seed, ctx=ctx, **optkw)))
    P = np.stack(plist)
    usable = usable + mh
    if len(usable) == 1:
        return P[0]
    return api.manifoldProduct(P, Niter=Niter, opts=api.make_opts(N=fg.N, seed=seed, **{k: v for k, v in optkw.items() if k == "stream_offset"}), ctx=ctx)
