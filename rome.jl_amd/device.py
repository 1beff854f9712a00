"""Device-resident belief store + graph-indexed convolution sweeps (torch = device memory/streams only).

One `sweep_*` call = one kernel launch over a whole table of (factor, direction) convolutions with
the beliefs staying in HBM -- what a clique/whole-graph pass of `solveTree!`
(examples/ManhattanDatasetBatch.jl:43; IIF upGibbsCliqueDensity -> approxConvBelief) issues.
"""
import ctypes as C

import numpy as np

from . import _lib
from .factors import Pose2, Point2, Pose3
from .graph import PackedGraph
from .api import cholesky_lower


def _require_torch_cuda():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("rome_jl_amd.device needs a HIP device (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch


class _PlanStub:
    """what DeviceGraph._plan returns on a plan-only graph: the launch descriptor's arguments, not a launch"""

    def __init__(self, fn, opts, kw):
        self.fn, self.kw = fn, kw
        self.opts = _lib.Opts.from_buffer_copy(opts)
        self._keep = (None, self.opts)

    def __call__(self):
        raise RuntimeError("plan-only DeviceGraph (bench.py --dry-run): there is no device to launch on")


class DeviceGraph:
    def __init__(self, fg_or_packed, device="cuda:0", ctx=None, plan_only=False):
        """plan_only: build the same tables on CPU tensors WITHOUT a device or a context -- the multi-GPU drivers can then lay out
        every rank's arena / exchange plan on a machine without GPUs (`bench.py --gpus 8 --dry-run`); nothing can be launched."""
        self.plan_only = bool(plan_only)
        if plan_only:
            import torch
            device = "cpu"
        else:
            torch = _require_torch_cuda()
        self.torch = torch
        self.device = torch.device(device)
        pk = fg_or_packed if isinstance(fg_or_packed, PackedGraph) else PackedGraph(fg_or_packed)
        self.packed = pk
        self.N = pk.N
        self.ctx = None if plan_only else (ctx or _lib.Context(self.device.index or 0))
        self._lib = None if plan_only else _lib.load()
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=self.device)
        f64, i32 = torch.float64, torch.int32
        self.bel = {Pose2: torch.zeros((len(pk.labels[Pose2]), 3, self.N), dtype=f64, device=self.device),
                    Point2: torch.zeros((len(pk.labels[Point2]), 2, self.N), dtype=f64, device=self.device),
                    Pose3: torch.zeros((len(pk.labels[Pose3]), 6, self.N), dtype=f64, device=self.device)}
        self.tab = {}
        # relative factors: both directions interleaved, then one ROME_DIR_PRIOR row per prior factor, so a
        # whole-graph sweep of one variable family is a single launch
        for name, tab, ptab, d in (("p2p2", pk.p2p2, pk.prior2, 3), ("p3p3", pk.p3p3, pk.prior3, 6)):
            if tab["F"] == 0 and ptab["F"] == 0:
                continue
            factor, dr, fixed, target = PackedGraph.conv_table(tab)
            F, P = tab["F"], ptab["F"]
            mu = np.concatenate([tab["mu"].reshape(F, d), ptab["mu"].reshape(P, d)])
            cov = np.concatenate([tab["cov"].reshape(F, d, d), ptab["cov"].reshape(P, d, d)])
            factor = np.concatenate([factor, F + np.arange(P, dtype=np.int32)])
            dr = np.concatenate([dr, np.full(P, 2, dtype=np.int32)])
            fixed = np.concatenate([fixed, ptab["var"]]); target = np.concatenate([target, ptab["var"]])
            # multihypo factors (Pose2Pose2 over two candidates of the second pose): alternative / probability per row, and one
            # more row per such factor -- the proposal of the second candidate -- BEHIND the prior rows (rows 2f+dir and 2F+p keep
            # their meaning).  A table with hypotheses runs on the general kernel, one without on the lean one.
            hyp = PackedGraph.conv_hypotheses(tab) if "alt" in tab else None
            E = 0
            alt = w = None
            if hyp is not None:
                alt2, w2, ex = hyp
                E = len(ex["factor"])
                alt = np.concatenate([alt2, np.full(P, -1, np.int32), ex["alt"]]); w = np.concatenate([w2, np.ones(P), ex["w"]])
                factor = np.concatenate([factor, ex["factor"]]); dr = np.concatenate([dr, ex["dir"]])
                fixed = np.concatenate([fixed, ex["fixed"]]); target = np.concatenate([target, ex["target"]])
            # nullhypo=p factors: one probability per row (both directions and the extra row of the factor; prior rows 0)
            nhf = tab.get("nh")
            nh = None
            if nhf is not None and np.any(nhf > 0):
                nh = np.concatenate([np.repeat(nhf, 2), np.zeros(P)] + ([nhf[ex["factor"]]] if E else []))
            # rows4: the four table columns interleaved (one 16-byte scalar load per convolution; selects the lean kernel)
            self.tab[name] = dict(F=F, P=P, E=E, C_rel=2 * F, C=2 * F + P + E, mu=t(mu, f64), L=t(cholesky_lower(cov), f64),
                                  nh=t(nh, f64) if nh is not None else None,
                                  factor=t(factor, i32), dir=t(dr, i32), fixed=t(fixed, i32), target=t(target, i32),
                                  rows4=t(np.stack([factor, dr, fixed, target], axis=1), i32), mh=hyp is not None,
                                  alt=t(alt, i32) if alt is not None else None, w=t(w, f64) if w is not None else None)
        if pk.br["F"]:
            b = pk.br
            r0 = b["rows0"]
            mh = bool((b["alt"] >= 0).any())
            nhb = b.get("nh")
            has_nh = nhb is not None and bool(np.any(nhb > 0))
            self.tab["br"] = dict(F=b["F"], F0=len(r0["factor"]), mh=mh, mu=t(b["mu"], f64), sigma=t(b["sigma"], f64),
                                  nh=t(nhb, f64) if has_nh else None, nh0=t(nhb[r0["factor"]], f64) if has_nh else None,
                                  pose=t(b["pose"], i32), point=t(b["point"], i32), alt=t(b["alt"], i32), w=t(b["w"], f64),
                                  factor0=t(r0["factor"], i32), pose0=t(r0["pose"], i32), point0=t(r0["point"], i32),
                                  alt0=t(r0["alt"], i32), w0=t(r0["w"], f64),
                                  rows4_0=t(np.stack([r0["factor"], np.zeros(len(r0["factor"]), np.int32), r0["pose"], r0["point"]], axis=1), i32),
                                  rows4_1=t(np.stack([np.arange(b["F"], dtype=np.int32), np.ones(b["F"], np.int32), b["point"], b["pose"]], axis=1), i32))
        for name, tab, d in (("prior2", pk.prior2, 3), ("prior3", pk.prior3, 6), ("priorpt2", pk.priorpt2, 2)):
            if tab["F"]:
                self.tab[name] = dict(F=tab["F"], mu=t(tab["mu"], f64), L=t(cholesky_lower(tab["cov"]), f64), var=t(tab["var"], i32))

        self._build_solve_tables(t, i32)

    # ---- proposal buffers + CSR (variable -> proposal rows) for the product / solve loop ----
    def _build_solve_tables(self, t, i32):
        torch, pk = self.torch, self.packed
        f64 = torch.float64
        C2 = self.tab["p2p2"]["C"] if "p2p2" in self.tab else 0
        Fb = self.tab["br"]["F"] if "br" in self.tab else 0
        Fb0 = self.tab["br"]["F0"] if "br" in self.tab else 0
        C3 = self.tab["p3p3"]["C"] if "p3p3" in self.tab else 0
        Ppt = self.tab["priorpt2"]["F"] if "priorpt2" in self.tab else 0   # landmark priors: one proposal row each, behind the sightings
        self.n_prop = {Pose2: C2 + Fb, Point2: Fb0 + Ppt, Pose3: C3}
        self.prop_bw = {}
        self.prop = {Pose2: torch.zeros((max(C2 + Fb, 1), 3, self.N), dtype=f64, device=self.device),
                     Point2: torch.zeros((max(Fb0 + Ppt, 1), 2, self.N), dtype=f64, device=self.device),
                     Pose3: torch.zeros((max(C3, 1), 6, self.N), dtype=f64, device=self.device)}
        self.bel_next = {vt: torch.zeros_like(self.bel[vt]) for vt in (Pose2, Point2, Pose3)}
        tgt2 = [self.tab["p2p2"]["target"].cpu().numpy()] if C2 else []
        if Fb:
            tgt2.append(pk.br["pose"])
        self._prop_targets = {Pose2: np.concatenate(tgt2) if tgt2 else np.zeros(0, np.int32),
                              Point2: np.concatenate([pk.br["rows0"]["point"] if Fb else np.zeros(0, np.int32),
                                                      pk.priorpt2["var"] if Ppt else np.zeros(0, np.int32)]),
                              Pose3: self.tab["p3p3"]["target"].cpu().numpy() if C3 else np.zeros(0, np.int32)}
        self.frozen = set()
        self._build_csr()

    def _build_csr(self):
        """variable -> proposal rows; frozen (marginalized) variables get no rows, so the product keeps their belief."""
        pk = self.packed
        t = lambda a, dt: self.torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=self.device)
        i32 = self.torch.int32
        self.csr = {}
        for vt in (Pose2, Point2, Pose3):
            tg = np.asarray(self._prop_targets[vt], dtype=np.int64)
            nv = len(pk.labels[vt])
            live = np.ones(nv + 1, dtype=bool)
            if self.frozen:
                for i, l in enumerate(pk.labels[vt]):
                    if l in self.frozen:
                        live[i] = False
            rows = np.nonzero(live[tg])[0] if len(tg) else np.zeros(0, np.int64)
            order = rows[np.argsort(tg[rows], kind="stable")].astype(np.int32)
            ptr = np.zeros(nv + 1, dtype=np.int32)
            np.add.at(ptr, tg[rows] + 1, 1)
            ptr = np.cumsum(ptr).astype(np.int32)
            self.csr[vt] = dict(ptr=t(ptr, i32), rows=t(order if len(order) else np.zeros(1, np.int32), i32),
                                ptr_h=ptr, rows_h=order)

    def set_frozen(self, labels):
        """Fixed-lag operation (IIF `fifoFreeze!` / isMarginalized): the beliefs of `labels` are no longer updated by product_step /
        solve; they still serve as the fixed side of every convolution they take part in (test/testFixedLagFG.jl:86-121)."""
        labels = set(labels)
        known = set(self.packed.labels[Pose2]) | set(self.packed.labels[Point2]) | set(self.packed.labels[Pose3])
        if not labels <= known:
            raise KeyError("set_frozen: unknown variables %s" % sorted(labels - known))
        self.frozen = labels
        self._build_csr()

    # ---- one uniform view of the convolution tables (what the multi-GPU drivers plan launches from) ----
    FAMILIES = {"p2p2": ("rome_conv_pose2pose2_dev", Pose2, Pose2, 0), "p3p3": ("rome_conv_pose3pose3_dev", Pose3, Pose3, 0),
                "br1": ("rome_conv_pose2point2br_dev", Point2, Pose2, 1), "br0": ("rome_conv_pose2point2br_dev", Pose2, Point2, 0)}

    def families(self):
        """Convolution families present in this graph, in launch order."""
        out = [f for f in ("p2p2", "p3p3") if f in self.tab and self.tab[f]["C"]]
        if "br" in self.tab:
            out += ["br1", "br0"]
        return out

    def family_table(self, fam):
        """-> dict(n, fn, vt_fixed, vt_target, dir_all, rows4 [n,4] int32 (factor, dir, fixed, target), mu, L, alt, w)."""
        name, vf, vt, d = self.FAMILIES[fam]
        fn = name if self.plan_only else getattr(self._lib, name)
        if fam in ("p2p2", "p3p3"):
            tb = self.tab[fam]
            return dict(n=tb["C"], fn=fn, vt_fixed=vf, vt_target=vt, dir_all=0, rows4=tb["rows4"], mu=tb["mu"], L=tb["L"],
                        alt=tb["alt"] if tb["mh"] else None, w=tb["w"] if tb["mh"] else None, nh=tb["nh"])
        tb = self.tab["br"]
        if fam == "br1":
            return dict(n=tb["F"], fn=fn, vt_fixed=vf, vt_target=vt, dir_all=1, rows4=tb["rows4_1"], mu=tb["mu"], L=tb["sigma"],
                        alt=tb["alt"] if tb["mh"] else None, w=tb["w"] if tb["mh"] else None, nh=tb["nh"])
        return dict(n=tb["F0"], fn=fn, vt_fixed=vf, vt_target=vt, dir_all=0, rows4=tb["rows4_0"], mu=tb["mu"], L=tb["sigma"],
                    alt=tb["alt0"] if tb["mh"] else None, w=tb["w0"] if tb["mh"] else None, nh=tb["nh0"])

    # ---- belief store ----
    def upload_beliefs(self, fg):
        for vt in (Pose2, Point2, Pose3):
            if len(self.packed.labels[vt]):
                self.bel[vt].copy_(self.torch.as_tensor(self.packed.beliefs(fg, vt)))

    def _bind_stream(self):
        self.ctx.set_stream(self.torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(x):
        return None if x is None else C.c_void_p(x.data_ptr())

    def _launch(self, fn, opts, **kw):
        self._plan(fn, opts, **kw)()

    def _plan(self, fn, opts, _ctx=None, **kw):
        """Pre-builds the rome_conv_dev descriptor once; the returned callable only binds the current
        torch stream and issues the launch (what a captured / replayed step calls).  `_ctx`: a Context whose stream the
        caller has fixed (one per pipeline slot) -- the launch then is a single C call with no stream lookup."""
        if self.plan_only:
            return _PlanStub(fn, opts, kw)
        cd = _lib.ConvDev()
        keep = []
        for k, v in kw.items():
            if k == "mirror_row":
                for m, r in enumerate(v):
                    cd.mirror_row[m] = int(r)
            elif isinstance(v, int):
                setattr(cd, k, v)
            elif v is not None:
                keep.append(v)
                setattr(cd, k, v.data_ptr())
        o = _lib.Opts.from_buffer_copy(opts)
        ctx, check, cur = self.ctx, _lib.check, self.torch.cuda.current_stream
        h = ctx.handle
        po, pc = C.byref(o), C.byref(cd)

        if _ctx is not None:
            hf = _ctx.handle

            def launch_fixed():
                rc = fn(hf, po, pc)
                if rc:
                    check(rc, hf)
            launch_fixed._keep = (keep, o, cd, _ctx)
            return launch_fixed

        def launch():
            ctx.set_stream(cur(self.device).cuda_stream)
            rc = fn(h, po, pc)
            if rc:
                check(rc, h)
        launch._keep = (keep, o, cd)
        return launch

    def plan_sweep_pose2pose2(self, opts, out, noise=None, status=None, fixed_ctx=None):
        """fixed_ctx: a Context whose stream the caller has set once (Context.set_stream): the launch is then ONE C call, without the
        per-launch lookup of torch's current stream (5.4 -> ~2 us of host time per launch)"""
        tb = self.tab["p2p2"]
        mh = dict(alt_var=tb["alt"], hypo_w=tb["w"]) if tb["mh"] else {}
        if tb["nh"] is not None:
            mh["nullhypo"] = tb["nh"]
        if fixed_ctx is not None:
            mh["_ctx"] = fixed_ctx
        return self._plan(self._lib.rome_conv_pose2pose2_dev, opts, n_conv=tb["C"], dir_all=0,
                          factor=tb["factor"], dir=tb["dir"], fixed_var=tb["fixed"], target_var=tb["target"], rows4=tb["rows4"],
                          mu=tb["mu"], L=tb["L"], bel_fixed=self.bel[Pose2], bel_target=self.bel[Pose2],
                          noise=noise, out=out, status=status, **mh)

    def plan_sample_priors(self, opts, out, kind="prior2", noise=None):
        tb = self.tab[kind]
        fn = self._lib.rome_sample_priorpose2_dev if kind == "prior2" else self._lib.rome_sample_priorpose3_dev
        return self._plan(fn, opts, n_conv=tb["F"], dir_all=0, mu=tb["mu"], L=tb["L"], noise=noise, out=out)

    # ---- solve loop pieces ----
    STREAM_P2P2, STREAM_BR1, STREAM_BR0, STREAM_PROD2, STREAM_PRODL, STREAM_P3P3, STREAM_PROD3, STREAM_PRIORPT2 = \
        0, 1 << 28, 2 << 28, 3 << 28, 4 << 28, 5 << 28, 6 << 28, 7 << 28

    def _opts_at(self, opts, offset):
        o = _lib.Opts.from_buffer_copy(opts)
        o.stream_offset = opts.stream_offset + offset
        return o

    def conv_step(self, opts, sweep=0):
        """All factor convolutions of the graph with the current beliefs -> self.prop (one launch per factor
        family/direction).  Philox streams: base + sweep·2³² + family offset + row."""
        base = sweep << 32
        C2 = self.tab["p2p2"]["C"] if "p2p2" in self.tab else 0
        if C2 and "br" in self.tab and self.tab["br"]["F"] and self.tab["br"]["F0"]:
            # a Pose2 / Point2 graph: ONE library call for the three families (one fused launch when the tables are plain)
            Fb, Fb0 = self.tab["br"]["F"], self.tab["br"]["F0"]
            self.sweep_graph_pose2(self._opts_at(opts, base), self.prop[Pose2][:C2], self.prop[Pose2][C2:C2 + Fb], self.prop[Point2][:Fb0])
        else:
            if C2:
                self.sweep_pose2pose2(self._opts_at(opts, base + self.STREAM_P2P2), out=self.prop[Pose2][:C2])
            if "br" in self.tab:
                Fb, Fb0 = self.tab["br"]["F"], self.tab["br"]["F0"]
                self.sweep_bearingrange(self._opts_at(opts, base + self.STREAM_BR1), 1, out=self.prop[Pose2][C2:C2 + Fb])
                self.sweep_bearingrange(self._opts_at(opts, base + self.STREAM_BR0), 0, out=self.prop[Point2][:Fb0])
        if "priorpt2" in self.tab:   # PriorPoint2 (src/factors/Point2D.jl:8-18): the landmark priors' samples, rows behind the sightings
            tp = self.tab["priorpt2"]
            Fb0 = self.tab["br"]["F0"] if "br" in self.tab else 0
            self._launch(self._lib.rome_sample_priorpoint2_dev, self._opts_at(opts, base + self.STREAM_PRIORPT2), n_conv=tp["F"], dir_all=0,
                         mu=tp["mu"], L=tp["L"], out=self.prop[Point2][Fb0:Fb0 + tp["F"]])
        if "p3p3" in self.tab and self.tab["p3p3"]["C"]:
            self.sweep_pose3pose3(self._opts_at(opts, base + self.STREAM_P3P3), out=self.prop[Pose3][:self.tab["p3p3"]["C"]])

    def product_step(self, opts, sweep=0, bandwidth="silverman", product="importance", gibbs_iters=1):
        """bel <- product of the proposals targeting each variable (Jacobi update: computed into bel_next, copied back in place
        so that launch plans holding the belief pointers stay valid).
        product:   "gibbs" = the reference's algorithm, ⚠AMP manifoldProduct / KDE.jl multiscale Gibbs sampling
                   (rome_product_gibbs_dev; Point2 / Pose2 / Pose3, N <= 256; always on the `manikde!` bandwidths of the proposals);
                   "importance" = the round-1 importance-sampling stand-in (rome_product_bw_dev).
        bandwidth: "silverman" (in-kernel rule on the proposal spread; importance product only) or "lcv" (leave-one-out
                   likelihood bandwidths of every proposal by rome_kde_bandwidth_dev first -- what the reference's `manikde!`
                   attaches to each convolution result)."""
        if bandwidth not in ("silverman", "lcv"):
            raise ValueError("bandwidth must be 'silverman' or 'lcv'")
        if product not in ("importance", "gibbs"):
            raise ValueError("product must be 'importance' or 'gibbs'")
        self._bind_stream()
        base = sweep << 32
        for vt, dim, off in ((Pose2, 3, self.STREAM_PROD2), (Point2, 2, self.STREAM_PRODL), (Pose3, 6, self.STREAM_PROD3)):
            V = self.bel[vt].shape[0]
            if V == 0:
                continue
            o = self._opts_at(opts, base + off)
            c = self.csr[vt]
            bw_ptr = None
            rows = self.n_prop[vt]
            gibbs = product == "gibbs"
            circ = 0b100 if vt is Pose2 else (0b111000 if vt is Pose3 else 0)   # bandwidth rule: which coordinates are angles
            if (bandwidth == "lcv" or gibbs) and rows:
                if vt not in self.prop_bw:
                    self.prop_bw[vt] = self.torch.empty((self.prop[vt].shape[0], dim), dtype=self.torch.float64, device=self.device)
                _lib.check(self._lib.rome_kde_bandwidth_dev(self.ctx.handle, dim, rows, self.N, self.prop[vt].data_ptr(), circ, 0.0, 0.0,
                                                            self.prop_bw[vt].data_ptr()),
                           self.ctx.handle)
                bw_ptr = self.prop_bw[vt].data_ptr()
            if gibbs and rows:
                max_k = max(1, int(np.diff(c["ptr_h"]).max())) if len(c["ptr_h"]) > 1 else 1
                _lib.check(self._lib.rome_product_gibbs_dev(self.ctx.handle, C.byref(o), dim, V, c["ptr"].data_ptr(), c["rows"].data_ptr(),
                                                            self.prop[vt].data_ptr(), bw_ptr, rows, self.bel[vt].data_ptr(),
                                                            self.bel_next[vt].data_ptr(), 0b100 if vt is Pose2 else 0, int(gibbs_iters), max_k),
                           self.ctx.handle)   # (Pose3: rotations are handled in the chart of each proposal, no wrapped coordinate)
            else:
                _lib.check(self._lib.rome_product_bw_dev(self.ctx.handle, C.byref(o), dim, V, c["ptr"].data_ptr(), c["rows"].data_ptr(),
                                                         self.prop[vt].data_ptr(), bw_ptr, self.bel[vt].data_ptr(),
                                                         self.bel_next[vt].data_ptr()), self.ctx.handle)
            self.bel[vt].copy_(self.bel_next[vt])

    def solve(self, opts, n_sweeps=10, bandwidth="silverman", product="importance", gibbs_iters=1):
        """n_sweeps x (convolution sweep, product): whole-graph nonparametric inference, a Jacobi schedule in place of the
        clique-by-clique Gibbs of `solveTree!` (no Bayes tree)."""
        self.check_particle_limits(bandwidth, product)
        for s in range(n_sweeps):
            self.conv_step(opts, s)
            self.product_step(opts, s, bandwidth, product, gibbs_iters)

    # ---- row-range forms of the two `next` stages, as the sharded drivers call them (rome_jl_amd.distributed) ----
    def kde_bandwidth_rows(self, dim, n_rows, prop, circ, bw_out):
        """manikde! bandwidths of `n_rows` proposal blocks (a contiguous device slice) -> bw_out [n_rows][dim]"""
        self._bind_stream()
        _lib.check(self._lib.rome_kde_bandwidth_dev(self.ctx.handle, dim, n_rows, self.N, prop.data_ptr(), circ, 0.0, 0.0, bw_out.data_ptr()),
                   self.ctx.handle)

    def product_gibbs_rows(self, opts, dim, V, ptr, rows, prop, bw, n_rows, bel_in, bel_out, circ, iters, max_k):
        """multiscale Gibbs product of V variables whose proposals are `rows` (CSR `ptr`) of the slice `prop` / `bw`"""
        self._bind_stream()
        _lib.check(self._lib.rome_product_gibbs_dev(self.ctx.handle, C.byref(opts), dim, V, ptr.data_ptr(), rows.data_ptr(), prop.data_ptr(),
                                                    bw.data_ptr(), n_rows, bel_in.data_ptr(), bel_out.data_ptr(), circ, iters, max_k),
                   self.ctx.handle)

    def check_particle_limits(self, bandwidth="silverman", product="importance"):
        """The stages of one solve iteration have different particle limits (include/rome_mi355.h): fail BEFORE the first launch,
        naming the stage, instead of part-way through an iteration."""
        N = self.N
        if product == "gibbs" and N > _lib.MAX_PARTICLES_GIBBS:
            raise ValueError("product='gibbs' (multiscale Gibbs product, the reference's manifoldProduct) takes N <= %d particles; this graph has "
                             "N = %d.  Use product='importance' (N <= %d) or fewer particles." % (_lib.MAX_PARTICLES_GIBBS, N, _lib.MAX_PARTICLES_PRODUCT))
        if (bandwidth == "lcv" or product == "gibbs") and N > _lib.MAX_PARTICLES_KDE:
            raise ValueError("manikde! bandwidths (bandwidth='lcv') take N <= %d particles; N = %d" % (_lib.MAX_PARTICLES_KDE, N))
        lim = _lib.MAX_PARTICLES_PRODUCT_POSE3 if self.bel[Pose3].shape[0] else _lib.MAX_PARTICLES_PRODUCT
        if product == "importance" and N > lim:
            raise ValueError("the importance product takes N <= %d particles here; N = %d (convolution sweeps alone go to %d)" % (lim, N, _lib.MAX_PARTICLES))

    def init_from_means(self, means, sigma=None, seed=3):
        """Beliefs = per-variable mean ⊕ N(0, diag σ²) jitter: e.g. means from solveGraphParametric (IIF can
        initialise the nonparametric solve from the parametric one: initParametricFrom!/autoinit)."""
        rng = np.random.default_rng(seed)
        for vt in (Pose2, Point2):
            ls = self.packed.labels[vt]
            if not ls:
                continue
            sg = np.asarray(sigma[vt] if sigma is not None else ([0.05, 0.05, 0.01] if vt is Pose2 else [0.1, 0.1]))
            m = np.stack([np.asarray(means[l], dtype=float) for l in ls])
            b = m[:, :, None] + sg[None, :, None] * rng.standard_normal((len(ls), vt.dim, self.N))
            self.bel[vt][:len(ls)].copy_(self.torch.as_tensor(b))
        ls = self.packed.labels[Pose3]
        if ls:   # Pose3: translation jitter added, rotation jitter composed on the right (R ← R Exp(e))
            from scipy.spatial.transform import Rotation as Rot
            sg = np.asarray(sigma[Pose3] if sigma is not None and Pose3 in sigma else [0.05, 0.05, 0.05, 0.01, 0.01, 0.01])
            m = np.stack([np.asarray(means[l], dtype=float) for l in ls])
            e = sg[None, :, None] * rng.standard_normal((len(ls), 6, self.N))
            b = np.empty_like(e)
            b[:, :3] = m[:, :3, None] + e[:, :3]
            for k in range(len(ls)):
                b[k, 3:] = (Rot.from_rotvec(m[k, 3:]) * Rot.from_rotvec(e[k, 3:].T)).as_rotvec().T
            self.bel[Pose3][:len(ls)].copy_(self.torch.as_tensor(b))

    def belief_stats(self, vartype):
        """(mean [V,dim], std [V,dim]) of every belief of one variable type, on device."""
        self._bind_stream()
        b = self.bel[vartype]
        V, d, N = b.shape
        mean = self.torch.empty((V, d), dtype=self.torch.float64, device=self.device)
        sd = self.torch.empty((V, d), dtype=self.torch.float64, device=self.device)
        _lib.check(self._lib.rome_belief_stats_dev(self.ctx.handle, d, V, N, b.data_ptr(), mean.data_ptr(), sd.data_ptr()), self.ctx.handle)
        return mean, sd

    def kde_bandwidths(self, vartype, tol_euclid=0.0, tol_circular=0.0):
        """[V, dim] leave-one-out likelihood bandwidths of every belief of one variable type (`manikde!` rule), on device."""
        self._bind_stream()
        b = self.bel[vartype]
        V, d, N = b.shape
        bw = self.torch.empty((V, d), dtype=self.torch.float64, device=self.device)
        _lib.check(self._lib.rome_kde_bandwidth_dev(self.ctx.handle, d, V, N, b.data_ptr(), 0b100 if vartype is Pose2 else 0,
                                                    float(tol_euclid), float(tol_circular), bw.data_ptr()), self.ctx.handle)
        return bw

    def download_beliefs(self, fg):
        for vt in (Pose2, Point2, Pose3):
            h = self.bel[vt].cpu().numpy()
            for k, l in enumerate(self.packed.labels[vt]):
                fg.vals[l] = h[k].copy()

    def capture(self, fn, warmup=2):
        """Capture `fn` (a sequence of launches on the current stream) into a hipGraph; returns the
        torch.cuda.CUDAGraph (call .replay()).  Launch-bound inner loops (one sweep is ~10-20 µs of GPU
        time) are replayed without per-launch host cost."""
        torch = self.torch
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        return g

    # ---- sweeps ----
    def sweep_pose2pose2(self, opts, out=None, noise=None, status=None, conv_slice=None):
        """All (factor, direction) Pose2Pose2 convolutions (row 2f+dir) followed by the PriorPose2 rows (and one row per multihypo
        factor for its second candidate) -> proposals [2F+P+E, 3, N], one launch."""
        tb = self.tab["p2p2"]
        lo, hi = (0, tb["C"]) if conv_slice is None else conv_slice
        n = hi - lo
        if out is None:
            out = self.torch.empty((n, 3, self.N), dtype=self.torch.float64, device=self.device)
        o = _lib.Opts.from_buffer_copy(opts); o.stream_offset = opts.stream_offset + lo
        mh = dict(alt_var=tb["alt"][lo:hi], hypo_w=tb["w"][lo:hi]) if tb["mh"] else {}
        if tb["nh"] is not None:
            mh["nullhypo"] = tb["nh"][lo:hi]
        self._launch(self._lib.rome_conv_pose2pose2_dev, o, n_conv=n, dir_all=0,
                     factor=tb["factor"][lo:hi], dir=tb["dir"][lo:hi], fixed_var=tb["fixed"][lo:hi], target_var=tb["target"][lo:hi],
                     rows4=tb["rows4"][lo:hi], mu=tb["mu"], L=tb["L"], bel_fixed=self.bel[Pose2], bel_target=self.bel[Pose2],
                     noise=noise, out=out, status=status, **mh)
        return out

    def sweep_pose3pose3(self, opts, out=None, noise=None, status=None):
        tb = self.tab["p3p3"]
        if out is None:
            out = self.torch.empty((tb["C"], 6, self.N), dtype=self.torch.float64, device=self.device)
        self._launch(self._lib.rome_conv_pose3pose3_dev, opts, n_conv=tb["C"], dir_all=0,
                     factor=tb["factor"], dir=tb["dir"], fixed_var=tb["fixed"], target_var=tb["target"], rows4=tb["rows4"],
                     mu=tb["mu"], L=tb["L"], bel_fixed=self.bel[Pose3], bel_target=self.bel[Pose3],
                     noise=noise, out=out, status=status, **({"nullhypo": tb["nh"]} if tb["nh"] is not None else {}))
        return out

    def sweep_bearingrange(self, opts, direction, out=None, noise=None, status=None):
        """direction 0: poses -> landmark proposals [F0,2,N] (one row per (factor, candidate landmark));
        1: landmarks -> pose proposals [F,3,N].  Multihypo factors draw the landmark per particle."""
        tb = self.tab["br"]
        dt = 2 if direction == 0 else 3
        nrow = tb["F0"] if direction == 0 else tb["F"]
        if out is None:
            out = self.torch.empty((nrow, dt, self.N), dtype=self.torch.float64, device=self.device)
        if direction == 0:
            kw = dict(factor=tb["factor0"], fixed_var=tb["pose0"], target_var=tb["point0"], rows4=tb["rows4_0"],
                      bel_fixed=self.bel[Pose2], bel_target=self.bel[Point2])
            if tb["mh"]:
                kw.update(alt_var=tb["alt0"], hypo_w=tb["w0"])
            if tb["nh0"] is not None:
                kw.update(nullhypo=tb["nh0"])
        else:
            kw = dict(factor=None, fixed_var=tb["point"], target_var=tb["pose"], rows4=tb["rows4_1"],
                      bel_fixed=self.bel[Point2], bel_target=self.bel[Pose2])
            if tb["mh"]:
                kw.update(alt_var=tb["alt"], hypo_w=tb["w"])
            if tb["nh"] is not None:
                kw.update(nullhypo=tb["nh"])
        self._launch(self._lib.rome_conv_pose2point2br_dev, opts, n_conv=nrow, dir_all=int(direction), dir=None,
                     mu=tb["mu"], L=tb["sigma"], noise=noise, out=out, status=status, **kw)
        return out

    def _conv_dev(self, keep, **kw):
        cd = _lib.ConvDev()
        for k, v in kw.items():
            if isinstance(v, int):
                setattr(cd, k, v)
            elif v is not None:
                keep.append(v)
                setattr(cd, k, v.data_ptr())
        return cd

    def sweep_graph_pose2(self, opts, out_p2p2, out_br1, out_br0, family_offsets=None):
        """The whole convolution sweep of a Pose2 / Point2 graph in ONE library call (rome_sweep_pose2_dev): Pose2Pose2 + PriorPose2
        rows, bearing-range -> pose rows, bearing-range -> landmark rows.  Plain tables run as one fused launch; tables with multihypo
        columns take the per-family launches inside the library -- the proposals are the same bit for bit either way.
        family_offsets: Philox stream offsets of the three families (default: STREAM_P2P2 / STREAM_BR1 / STREAM_BR0)."""
        keep = []
        t2, tb = self.tab["p2p2"], self.tab["br"]
        mh2 = dict(alt_var=t2["alt"], hypo_w=t2["w"]) if t2["mh"] else {}
        if t2["nh"] is not None:
            mh2["nullhypo"] = t2["nh"]
        c2 = self._conv_dev(keep, n_conv=t2["C"], dir_all=0, rows4=t2["rows4"], factor=t2["factor"], dir=t2["dir"], fixed_var=t2["fixed"],
                            target_var=t2["target"], mu=t2["mu"], L=t2["L"], bel_fixed=self.bel[Pose2], bel_target=self.bel[Pose2], out=out_p2p2, **mh2)
        mh1 = dict(alt_var=tb["alt"], hypo_w=tb["w"]) if tb["mh"] else {}
        if tb["nh"] is not None:
            mh1["nullhypo"] = tb["nh"]
        c1 = self._conv_dev(keep, n_conv=tb["F"], dir_all=1, rows4=tb["rows4_1"], fixed_var=tb["point"], target_var=tb["pose"], mu=tb["mu"],
                            L=tb["sigma"], bel_fixed=self.bel[Point2], bel_target=self.bel[Pose2], out=out_br1, **mh1)
        mh0 = dict(alt_var=tb["alt0"], hypo_w=tb["w0"]) if tb["mh"] else {}
        if tb["nh0"] is not None:
            mh0["nullhypo"] = tb["nh0"]
        c0 = self._conv_dev(keep, n_conv=tb["F0"], dir_all=0, rows4=tb["rows4_0"], factor=tb["factor0"], fixed_var=tb["pose0"],
                            target_var=tb["point0"], mu=tb["mu"], L=tb["sigma"], bel_fixed=self.bel[Pose2], bel_target=self.bel[Point2],
                            out=out_br0, **mh0)
        offs = (C.c_uint64 * 3)(*(family_offsets if family_offsets is not None else (self.STREAM_P2P2, self.STREAM_BR1, self.STREAM_BR0)))
        self._bind_stream()
        _lib.check(self._lib.rome_sweep_pose2_dev(self.ctx.handle, C.byref(opts), C.byref(c2), C.byref(c1), C.byref(c0), offs), self.ctx.handle)

    def sample_priors(self, opts, kind="prior2", out=None, noise=None):
        tb = self.tab[kind]
        d = 3 if kind == "prior2" else 6
        if out is None:
            out = self.torch.empty((tb["F"], d, self.N), dtype=self.torch.float64, device=self.device)
        fn = self._lib.rome_sample_priorpose2_dev if kind == "prior2" else self._lib.rome_sample_priorpose3_dev
        self._launch(fn, opts, n_conv=tb["F"], dir_all=0, factor=None, dir=None, fixed_var=None, target_var=None,
                     mu=tb["mu"], L=tb["L"], bel_fixed=None, bel_target=None, noise=noise, out=out, status=None)
        return out
