"""Multi-GPU layer: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

The hot path shards by graph segment / clique sub-tree: every rank owns a contiguous range of the
variables and runs the convolutions that TARGET its variables -- no collective on the data path of a
sweep.  The one real exchange step is the separator message of the Bayes tree (IIF `LikelihoodMessage`
holding a `TreeBelief` of N points per separator variable; SURVEY.md §5 "distributed communication
backend"): after a sweep every rank publishes the beliefs of its boundary variables and receives the
ones its cut factors read (`SeparatorExchange`).  Messages are N x dim doubles (2.4 kB per Pose2), i.e.
latency-bound: the collective is posted asynchronously right after a sweep and completed just before the
next one, so it overlaps with the next sweep's launch instead of serialising with it (the cut factors then
read separator beliefs that are one sweep old -- the same asynchrony IIF's clique tasks have).
"""
import numpy as np


def shard_range(n_items, world, rank):
    """Contiguous balanced partition of range(n_items): -> (lo, hi) of `rank`."""
    q, r = divmod(int(n_items), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def owner_of(index, n_items, world):
    """Inverse of shard_range: rank owning item `index`."""
    q, r = divmod(int(n_items), int(world))
    cut = r * (q + 1)
    return index // (q + 1) if index < cut else r + (index - cut) // max(q, 1)


def shard_convolutions_by_target(target_var, n_vars, world, rank):
    """Rows of a convolution table whose TARGET variable is owned by `rank` (variables are owned in
    contiguous ranges).  Returns the sorted row indices (int64)."""
    lo, hi = shard_range(n_vars, world, rank)
    t = np.asarray(target_var)
    return np.nonzero((t >= lo) & (t < hi))[0]


class SeparatorExchange:
    """All-gather of separator beliefs between graph segments.

    Every rank publishes `n_sep` belief blocks (rows `publish_rows` of a source tensor [rows, dim, N]);
    `ghosts` lists, for each ghost block this rank keeps, which (source rank, source slot) feeds it.
    One fixed-size `all_gather_into_tensor` per exchange (the payload is tiny, so one collective beats
    per-edge send/recv).  `post()` launches it asynchronously, `complete()` waits and scatters.
    """

    def __init__(self, torch, dist, world, rank, publish_rows, ghosts, dim, N, device, dtype=None, always_collective=False):
        self.torch, self.dist = torch, dist
        self.world, self.rank = world, rank
        self.always_collective = always_collective  # issue the collective even for world == 1 (smoke tests)
        dtype = dtype or torch.float64
        n_sep = len(publish_rows)
        self.publish_rows = torch.as_tensor(np.asarray(publish_rows, dtype=np.int64), device=device)
        self.send = torch.zeros((n_sep, dim, N), dtype=dtype, device=device)
        self.recv = torch.zeros((world * n_sep, dim, N), dtype=dtype, device=device)
        self.ghost_blocks = torch.as_tensor(np.asarray([g[0] for g in ghosts], dtype=np.int64), device=device)
        self.ghost_src = torch.as_tensor(np.asarray([(g[1] % world) * n_sep + g[2] for g in ghosts], dtype=np.int64), device=device)
        self.pending = None
        self.have_data = False

    def post(self, source):
        """Gather this rank's separator rows from `source` and launch the collective (asynchronously)."""
        self.torch.index_select(source, 0, self.publish_rows, out=self.send)
        if self.world > 1 or self.always_collective:
            self.pending = self.dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1), async_op=True)
        else:
            self.recv.copy_(self.send)
        self.have_data = True

    def complete(self, beliefs):
        """Wait for the posted collective (if any) and overwrite the ghost blocks of `beliefs` [V, dim, N]."""
        if not self.have_data:
            return
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
        beliefs.index_copy_(0, self.ghost_blocks, self.recv.index_select(0, self.ghost_src))
        self.have_data = False

    def exchange(self, source, beliefs):
        """Synchronous form: post + complete."""
        self.post(source)
        self.complete(beliefs)


def chain_segment_exchange(torch, dist, world, rank, N, device, publish_rows, ghost_prev, ghost_next, always_collective=False):
    """The exchange used by bench.py / the weak-scaling layout: segments in a ring; every rank publishes two
    rows of its proposal table (slot 0: its first pose, slot 1: its last pose);
    ghost_prev <- previous rank's slot 1, ghost_next <- next rank's slot 0."""
    return SeparatorExchange(torch, dist, world, rank, publish_rows,
                             [(ghost_prev, rank - 1, 1), (ghost_next, rank + 1, 0)], 3, N, device,
                             always_collective=always_collective)


class SeparatorPipeline:
    """Weak-scaling sweep driver for a rank-local graph segment of ANY mix of factor families (Pose2Pose2, Pose3Pose3,
    bearing-range in both directions), with the separator exchange fully off the critical path.

    Every rank holds its own segment as a DeviceGraph in which the remote separator variables appear as *ghost* variables.
    `publish`: [(family, conv_row)] -- proposal rows of this rank whose block is the message for the neighbours (the sweep
    kernel itself mirrors them into the RCCL send buffer through `rome_conv_dev.mirror_map`: no gather kernel, any number of rows);
    the k-th published block of a variable type is that type's *slot* k.  Every rank must publish the same number of blocks
    per variable type (one fixed-size all-gather per step carries all types).
    `ghosts`: [(vartype, local_ghost_index, source_rank, source_slot)].

    Memory: one arena of doubles owned by the pipeline (dg.bel is left untouched): the belief stores of the variable types,
    then `depth` receive buffers.  All region offsets are multiples of 6N doubles (= lcm of the block sizes 2N, 3N, 6N), so a
    block of the receive buffer IS block number g of the type's store for an integer g: the all-gather lands straight in ghost
    blocks (no scatter kernel) and the cut factors' table entries simply point there.
    Send and receive buffers are `depth`-fold buffered (default 2): sweep k reads the separators gathered after sweep k-depth
    while the collectives k-depth+1 .. k-1 are still in flight, so nothing races and the result is deterministic for a fixed
    schedule.  Host work per step: one launch per family + one async collective.
    """

    def __init__(self, dg, opts, dist, world, rank, publish, ghosts, always_collective=False, depth=2, rccl_comms=None):
        torch = dg.torch
        self.dg, self.dist, self.world, self.rank = dg, dist, world, rank
        self.collective = world > 1 or always_collective
        if depth < 2:
            raise ValueError("SeparatorPipeline needs depth >= 2")
        self.depth = D = int(depth)
        N = dg.N
        fams = dg.families()
        ftab = {f: dg.family_table(f) for f in fams}
        vts = []
        for f in fams:
            for vt in (ftab[f]["vt_fixed"], ftab[f]["vt_target"]):
                if vt not in vts:
                    vts.append(vt)
        dim = {vt: int(vt.dim) for vt in vts}
        U = 6 * N
        pad = lambda n: -(-n // U) * U
        any_bel = dg.bel[vts[0]]
        dev, dt = any_bel.device, any_bel.dtype
        # published slots per variable type; per family a contiguous slot range (one mirror_out pointer per launch)
        pub_by_f = {f: [int(r) for ff, r in publish if ff == f] for f in fams}
        for ff, _ in publish:
            if ff not in fams:
                raise ValueError("publish: family %r is not in this graph" % (ff,))
        for f, v in pub_by_f.items():
            if len(set(v)) != len(v):
                raise ValueError("family %s publishes a row twice" % f)
        slot0, npub = {}, {vt: 0 for vt in vts}
        for f in fams:
            vt = ftab[f]["vt_target"]
            slot0[f] = npub[vt]
            npub[vt] += len(pub_by_f[f])
        sec_off, off = {}, 0
        for vt in vts:
            sec_off[vt] = off
            off += pad(npub[vt] * dim[vt] * N)
        self.payload = payload = max(off, U)
        store_off, off = {}, 0
        for vt in vts:
            store_off[vt] = off
            off += pad(dg.bel[vt].shape[0] * dim[vt] * N)
        recv_off = [off + b * world * payload for b in range(D)]
        self.arena = torch.zeros(off + D * world * payload, dtype=dt, device=dev)
        self.V = {vt: dg.bel[vt].shape[0] for vt in vts}
        self.store = {}
        for vt in vts:
            nblk = (self.arena.numel() - store_off[vt]) // (dim[vt] * N)
            self.store[vt] = self.arena[store_off[vt]: store_off[vt] + nblk * dim[vt] * N].view(nblk, dim[vt], N)
            self.store[vt][:self.V[vt]].copy_(dg.bel[vt])
        self.recv = [self.arena[recv_off[b]: recv_off[b] + world * payload] for b in range(D)]
        self.send = [torch.zeros(payload, dtype=dt, device=dev) for _ in range(D)]
        self.out = [{f: torch.empty((ftab[f]["n"], dim[ftab[f]["vt_target"]], N), dtype=dt, device=dev) for f in fams} for _ in range(D)]
        self.families = fams
        self.props = [o.get("p2p2") for o in self.out]   # (the Pose2Pose2 tables, what the chain-of-segments bench reads)
        self.prop = self.props[0]
        self.works = [None] * D
        # direct RCCL form (rccl.create_comms, one communicator per slot): slot b owns stream b and a rome_ctx bound to it; sweeps
        # and ncclAllGather are enqueued back to back on that stream, so the collective needs no event join and the host work per
        # step is a few C calls.  Otherwise torch.distributed's collective (≈ 30 µs of host time per call) with stream-level joins.
        self.comms = rccl_comms if (rccl_comms and self.collective and any_bel.is_cuda) else None
        self.streams = None
        slot_ctx = [None] * D
        if self.collective and hasattr(torch, "cuda") and any_bel.is_cuda:
            cur = torch.cuda.current_stream(dev)
            self.streams = [torch.cuda.Stream(dev) for _ in range(D)]
            for st in self.streams:
                st.wait_stream(cur)
            if self.comms is not None:
                if len(self.comms) != D:
                    raise ValueError("need one RCCL communicator per pipeline slot")
                from . import _lib as _l
                for b in range(D):
                    slot_ctx[b] = _l.Context(dev.index or 0)
                    slot_ctx[b].set_stream(self.streams[b].cuda_stream)

        def ghost_block(vt, b, src_rank, src_slot):
            o = recv_off[b] + (src_rank % world) * payload + sec_off[vt] + src_slot * dim[vt] * N - store_off[vt]
            assert o % (dim[vt] * N) == 0
            return o // (dim[vt] * N)

        self.plans = []
        for b in range(D):
            remap = {vt: {} for vt in vts}
            for vt, gi, src, slot in ghosts:
                if slot >= npub[vt]:
                    raise ValueError("ghost refers to slot %d of %s but only %d are published" % (slot, vt.__name__, npub[vt]))
                g = ghost_block(vt, b, src, slot)
                remap[vt][int(gi)] = g
                self.store[vt][g].copy_(dg.bel[vt][int(gi)])   # until the first message arrives: the local initial belief
            plans_b = []
            for f in fams:
                tb = ftab[f]
                rows = tb["rows4"].clone()
                for col, vt in ((2, tb["vt_fixed"]), (3, tb["vt_target"])):
                    for gi, g in remap[vt].items():
                        rows[:, col][tb["rows4"][:, col] == gi] = g
                alt = tb["alt"]
                if alt is not None:
                    vl = tb["vt_fixed"] if f == "br1" else tb["vt_target"]   # the landmark slot
                    alt = alt.clone()
                    for gi, g in remap[vl].items():
                        alt[tb["alt"] == gi] = g
                vt_t = tb["vt_target"]
                kw = dict(n_conv=tb["n"], dir_all=tb["dir_all"], rows4=rows, mu=tb["mu"], L=tb["L"],
                          bel_fixed=self.store[tb["vt_fixed"]], bel_target=self.store[vt_t], out=self.out[b][f])
                if alt is not None:
                    kw.update(alt_var=alt, hypo_w=tb["w"])
                if tb.get("nh") is not None:     # nullhypo= factors: one probability per row
                    kw.update(nullhypo=tb["nh"])
                if pub_by_f[f]:
                    lo = sec_off[vt_t] + slot0[f] * dim[vt_t] * N
                    # any number of separator rows per family: row -> slot map (rome_conv_dev.mirror_map), -1 = not published
                    mm = torch.full((tb["n"],), -1, dtype=torch.int32)
                    mm[torch.as_tensor(pub_by_f[f], dtype=torch.long)] = torch.arange(len(pub_by_f[f]), dtype=torch.int32)
                    kw.update(mirror_map=mm.to(dev), mirror_out=self.send[b][lo: lo + len(pub_by_f[f]) * dim[vt_t] * N])
                if slot_ctx[b] is not None:
                    kw["_ctx"] = slot_ctx[b]
                plans_b.append(dg._plan(tb["fn"], opts, **kw))
            self.plans.append(plans_b)
        self._ag_args = [(self.send[b].data_ptr(), self.recv[b].data_ptr(), self.send[b].numel(),
                          self.streams[b].cuda_stream if self.streams is not None else 0) for b in range(D)]
        self.k = 0
        # Steps k, k+1, .. run round-robin on `depth` streams.  In the torch.distributed form the join "sweep k+depth waits for
        # collective k" is a wait-for-event packet at the head of stream k % depth (≈ 8 µs of queue latency on this stack, measured);
        # with one stream it would sit between two sweeps, here it elapses while the other streams' sweeps keep the GPU busy.

    def _step_on_current_stream(self, b):
        w = self.works[b]
        if w is not None:
            w.wait()   # stream-level join (the host runs several steps ahead of the GPU: a host-side query cannot replace it)
        for pl in self.plans[b]:
            pl()
        if self.collective:
            self.works[b] = self.dist.all_gather_into_tensor(self.recv[b], self.send[b], async_op=True)
        else:
            self.recv[b][:self.payload].copy_(self.send[b])

    def step(self):
        b = self.k % self.depth
        if self.comms is not None:
            for pl in self.plans[b]:
                pl()                                         # sweeps of step k on stream b (their rome_ctx is bound to it)
            self.comms[b].all_gather_f64(*self._ag_args[b])  # ... followed on the same stream by the separator exchange
        elif self.streams is not None:
            with self.dg.torch.cuda.stream(self.streams[b]):
                self._step_on_current_stream(b)
        else:
            self._step_on_current_stream(b)
        self.prop = self.props[b]   # the table the latest step writes
        self.k += 1

    def drain(self):
        """Wait for the collectives in flight and re-join the step streams into the caller's stream."""
        for b in range(self.depth):
            if self.works[b] is not None:
                if self.streams is not None:
                    with self.dg.torch.cuda.stream(self.streams[b]):
                        self.works[b].wait()
                else:
                    self.works[b].wait()
                self.works[b] = None
        if self.streams is not None:
            cur = self.dg.torch.cuda.current_stream(self.arena.device)
            for st in self.streams:
                cur.wait_stream(st)

    def close(self):
        if self.comms is not None:
            self.drain()
            self.dg.torch.cuda.synchronize()
            for c in self.comms:
                c.close()
            self.comms = None


class PipelinedSegmentSweep(SeparatorPipeline):
    """The chain-of-segments layout of bench.py's weak scaling: a ring of Manhattan-shaped segments, every rank publishes the
    proposals of its first and last pose (`sep_rows`: Pose2Pose2 table rows, slots 0 / 1); ghost_prev <- previous rank's slot 1,
    ghost_next <- next rank's slot 0."""

    def __init__(self, dg, opts, dist, world, rank, sep_rows, ghost_prev, ghost_next, always_collective=False, depth=2, rccl_comms=None):
        from .factors import Pose2
        super().__init__(dg, opts, dist, world, rank, [("p2p2", int(r)) for r in sep_rows],
                         [(Pose2, ghost_prev, rank - 1, 1), (Pose2, ghost_next, rank + 1, 0)],
                         always_collective=always_collective, depth=depth, rccl_comms=rccl_comms)


class TargetShardedSweep:
    """STRONG scaling of one graph: every rank keeps the whole belief store and the whole convolution table of a
    family, sorted by target variable; rank r owns variables [r·q, (r+1)·q) (q = ceil(V / world)) and sweeps exactly the rows that
    target them -- a contiguous row range of the sorted table, whatever the world size, so the Philox stream of a row (its
    position in the sorted table) and hence every proposal is partition-independent.  After the sweep (and, in a solve loop, the
    product of the owned variables, which finds its proposals contiguous) one all-gather of the owned belief blocks
    (V/world x dim x N doubles per rank; Manhattan: 8.4 MB in total) restores the replicated store."""

    def __init__(self, dg, opts, dist, world, rank, family="p2p2", always_collective=False, rccl_comm=None):
        torch = dg.torch
        self.dg, self.dist, self.world, self.rank = dg, dist, world, rank
        self.collective = world > 1 or always_collective
        tb = dg.family_table(family)
        self.family, vt = family, tb["vt_target"]
        if tb["vt_fixed"] is not vt:
            raise ValueError("TargetShardedSweep shards a single-variable-type family (p2p2 / p3p3)")
        N, d = dg.N, int(vt.dim)
        V = dg.bel[vt].shape[0]
        self.q = q = -(-V // world)
        rows_h = tb["rows4"].cpu().numpy()
        order = np.argsort(rows_h[:, 3], kind="stable")
        sorted_rows = rows_h[order]
        ptr = np.zeros(world * q + 1, dtype=np.int64)
        np.add.at(ptr, sorted_rows[:, 3].astype(np.int64) + 1, 1)
        self.ptr = ptr = np.cumsum(ptr)
        self.order = order                                    # sorted row j = original row order[j]
        self.row_lo, self.row_hi = int(ptr[rank * q]), int(ptr[min((rank + 1) * q, world * q)])
        any_bel = dg.bel[vt]
        self.store = torch.zeros((world * q, d, N), dtype=any_bel.dtype, device=any_bel.device)
        self.store[:V].copy_(any_bel)
        self.rows4 = torch.as_tensor(np.ascontiguousarray(sorted_rows), dtype=torch.int32, device=any_bel.device)
        self.prop = torch.zeros((tb["n"], d, N), dtype=any_bel.dtype, device=any_bel.device)   # sorted-table order
        o = type(opts).from_buffer_copy(opts) if hasattr(type(opts), "from_buffer_copy") else opts
        if hasattr(o, "stream_offset"):
            o.stream_offset = opts.stream_offset + self.row_lo
        n = self.row_hi - self.row_lo
        self.n_rows = n
        mh = {}
        if tb["alt"] is not None:   # multihypo rows travel with their rows through the sort
            idx = torch.as_tensor(order, device=tb["alt"].device)
            self.alt, self.w = tb["alt"][idx].contiguous(), torch.as_tensor(tb["w"])[idx.to(torch.as_tensor(tb["w"]).device)].contiguous()
            mh = dict(alt_var=self.alt[self.row_lo:self.row_hi], hypo_w=self.w[self.row_lo:self.row_hi])
        if tb.get("nh") is not None:   # nullhypo rows likewise
            self.nh = tb["nh"][torch.as_tensor(order, device=tb["nh"].device)].contiguous()
            mh["nullhypo"] = self.nh[self.row_lo:self.row_hi]
        self._mh = mh
        self.plan = dg._plan(tb["fn"], o, n_conv=n, dir_all=tb["dir_all"], rows4=self.rows4[self.row_lo:self.row_hi], mu=tb["mu"], L=tb["L"],
                             bel_fixed=self.store, bel_target=self.store, out=self.prop[self.row_lo:self.row_hi], **mh) if n else (lambda: None)
        self.mine = self.store[rank * q:(rank + 1) * q]
        self.comm = rccl_comm if (rccl_comm is not None and self.collective and any_bel.is_cuda) else None
        self.work = None

    def step(self):
        """sweep of the owned rows, then the all-gather of the owned belief blocks (asynchronous; `wait()` completes it)."""
        self.wait()
        self.plan()
        self.exchange()

    def solve_step(self, opts, sweep=0, gibbs_iters=1):
        """One strong-scaled SOLVE iteration with the reference's operations: sweep of the owned rows, `manikde!` bandwidths of
        those proposals, multiscale Gibbs product (manifoldProduct) of the owned variables -- whose proposals are a contiguous
        row range of the target-sorted table -- then the all-gather of the owned beliefs.  Everything but the final exchange is
        rank-local; bandwidths and product (5.7 of the 5.9 ms of a Manhattan iteration) shard perfectly."""
        dg, torch = self.dg, self.dg.torch
        self.wait()
        if not hasattr(self, "_solve"):
            dev = self.store.device
            n, d = self.prop.shape[0], self.store.shape[1]
            lo_v0 = self.rank * self.q
            self._solve = dict(bw=torch.zeros((max(n, 1), d), dtype=self.store.dtype, device=dev),
                               # CSR of the OWNED variables over the OWNED rows only (rebased to row_lo): the tree build and the
                               # bandwidths then touch this rank's rows and nothing else
                               ptr=torch.as_tensor((self.ptr[lo_v0:lo_v0 + self.q + 1] - self.row_lo).astype(np.int32), device=dev),
                               rows=torch.arange(max(self.n_rows, 1), dtype=torch.int32, device=dev),
                               out=torch.zeros_like(self.mine),
                               max_k=int(max(1, np.diff(self.ptr).max())))
        S = self._solve
        d, N = self.store.shape[1], self.store.shape[2]
        circ = 0b100 if d == 3 else 0
        o = type(opts).from_buffer_copy(opts)
        o.stream_offset = opts.stream_offset + (sweep << 32) + self.row_lo
        lo_v, q = self.rank * self.q, self.q
        if self.n_rows:
            # the sweep of the owned rows with this iteration's Philox streams, then the manikde! bandwidths of those proposals
            self._sweep_plan(o)()
            dg.kde_bandwidth_rows(d, self.n_rows, self.prop[self.row_lo:self.row_hi], circ, S["bw"][self.row_lo:self.row_hi])
        op = type(opts).from_buffer_copy(opts)
        op.stream_offset = opts.stream_offset + (sweep << 32) + dg.STREAM_PROD2 + lo_v      # product stream = global variable id
        dg.product_gibbs_rows(op, d, q, S["ptr"], S["rows"], self.prop[self.row_lo:], S["bw"][self.row_lo:], self.n_rows,
                              self.mine, S["out"], circ, int(gibbs_iters), S["max_k"])
        self.mine.copy_(S["out"])
        self.exchange()

    def _sweep_plan(self, o):
        """The launch of the owned rows (multihypo columns included) with the Philox streams of `o`: built once, afterwards only
        the stream offset of the cached descriptor's options changes."""
        if getattr(self, "_solve_plan", None) is None:
            tb = self.dg.family_table(self.family)
            self._solve_plan = self.dg._plan(tb["fn"], o, n_conv=self.n_rows, dir_all=tb["dir_all"], rows4=self.rows4[self.row_lo:self.row_hi],
                                             mu=tb["mu"], L=tb["L"], bel_fixed=self.store, bel_target=self.store,
                                             out=self.prop[self.row_lo:self.row_hi], **self._mh)
        self._solve_plan._keep[1].stream_offset = o.stream_offset
        return self._solve_plan

    def exchange(self):
        if not self.collective:
            return
        if self.comm is not None:   # in place: the send block is this rank's slice of the receive buffer
            st = self.dg.torch.cuda.current_stream(self.store.device).cuda_stream
            self.comm.all_gather_f64(self.mine.data_ptr(), self.store.data_ptr(), self.mine.numel(), st)
        else:
            self.work = self.dist.all_gather_into_tensor(self.store.view(-1), self.mine.reshape(-1).clone(), async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    def close(self):
        self.wait()
        if self.comm is not None:
            self.dg.torch.cuda.synchronize()
            self.comm.close()
            self.comm = None


class FrontierShard:
    """The clique FRONTIER sharded across GPUs, device-resident (north_star: "the Bayes-tree clique frontier shards
    naturally across the 8 GPUs with separator-belief messages exchanged via RCCL over xGMI").

    Every rank holds the beliefs of the whole graph in a `DeviceStore` (8.4 MB for Manhattan-3500): uploaded ONCE.  A frontier of
    independent cliques is dealt round-robin to the ranks; `plan(cliques)` builds, once per frontier (a tree level: reused by every
    pass of the solve),
      * an `UpsolvePlan` for this rank's share -- row tables on the device, Philox stream ids = positions in the WHOLE frontier's
        tables (`plan_frontier`), so the shares together draw exactly what ONE unsharded call draws: the result does not depend on
        the number of ranks (tests: world 2 and world 8 with empty shares == the single call, bit for bit);
      * the exchange layout: blocks of 6N doubles, `width` = the largest share; rank r's k-th new frontal is block r * width + k of
        the receive buffer -- every rank knows the clique list, so no metadata travels;
      * a `ScatterPlan` (receive buffer -> store) over the blocks of the other ranks.
    `step(plan, opts)` = the share's up-solve (its product kernel writes the new frontal beliefs in place AND into this rank's slice
    of the receive buffer: no gather kernel), ONE all-gather in place (direct `ncclAllGather` on the context's stream when `comm` is
    given, else torch.distributed), ONE scatter launch.  No belief crosses PCIe after the initial upload; nothing synchronises with
    the host.
    store_cls / plan_cls / scatter_cls: injected by the CPU tests (oracle-backed stand-ins over torch CPU tensors)."""

    def __init__(self, store, torch, dist, world, rank, device="cpu", comm=None, always_collective=False, plan_cls=None, scatter_cls=None):
        device = torch.device(device)       # (a string such as "cuda:0" is a natural argument: the stream binding below tests .type)
        self.store, self.torch, self.dist, self.world, self.rank, self.device = store, torch, dist, world, rank, device
        if comm is not None and not (device.type == "cuda" and hasattr(store, "ctx")):
            raise ValueError("a direct RCCL communicator needs a CUDA device and a device store (the collective is ordered on the context's stream)")
        self.comm, self.always_collective = comm, always_collective
        if plan_cls is None or scatter_cls is None:
            from .clique import UpsolvePlan, ScatterPlan
            plan_cls, scatter_cls = plan_cls or UpsolvePlan, scatter_cls or ScatterPlan
        self.plan_cls, self.scatter_cls = plan_cls, scatter_cls

    def shares(self, n_cliques):
        return [list(range(r, n_cliques, self.world)) for r in range(self.world)]

    def _layout(self, labels, vartype_of):
        """PACKED exchange layout: slots of N doubles, a belief of dimension d takes d consecutive slots (Pose2 3, Point2 2, Pose3 6: no
        padding to the widest type); rank r's blocks start at slot r * width, width = the largest share.  Every rank derives the same
        layout from the clique list, so no metadata travels.  -> (width, [{label: first slot} per rank])"""
        offs, tot = [], []
        for ls in labels:
            o, k = {}, 0
            for l in ls:
                o[l] = k; k += vartype_of(l).dim
            offs.append(o); tot.append(k)
        width = max(1, max(tot))
        return width, [{l: r * width + k for l, k in o.items()} for r, o in enumerate(offs)]

    def _exchange_plan(self, labels, vartype_of, make_plan):
        torch, N = self.torch, self.store.N
        width, slots = self._layout(labels, vartype_of)
        recv = torch.zeros(self.world * width * N, dtype=torch.float64, device=self.device)
        send = recv[self.rank * width * N:(self.rank + 1) * width * N]            # in place: this rank's slice of the receive buffer
        up = make_plan({l: b - self.rank * width for l, b in slots[self.rank].items()}) if labels[self.rank] else None   # (slots within `send`)
        # (always_collective: the one-rank measurement form -- the own blocks go through the exchange buffer and the scatter too)
        others = [(l, b) for r in range(self.world) if r != self.rank or self.always_collective for l, b in slots[r].items()]
        sc = self.scatter_cls(self.store, [l for l, _ in others], [b for _, b in others], stride=N) if others else None
        return dict(up=up, scatter=sc, recv=recv, send=send, width=width, U=N, labels=labels)

    def plan(self, cliques, gibbsIters=3, Niter=1, usable=None):
        cliques = [list(c) for c in cliques]
        shares = self.shares(len(cliques))
        labels = [[l for k in sh for l in cliques[k]] for sh in shares]
        fgv = self.store.fg.variables
        return self._exchange_plan(labels, fgv.__getitem__, lambda mirror: self.plan_cls(
            self.store, cliques, share=shares[self.rank], gibbsIters=gibbsIters, Niter=Niter, mirror=mirror, usable=usable))

    def plan_level(self, spec, level_plan_cls):
        """One level of a Bayes tree (tree.LevelSpec: multi-frontal cliques, separator copies, messages of the children) dealt round-robin
        by CLIQUE: a rank solves all updated variables of its cliques; what travels is every block the level writes (frontals, the
        cliques' private copies) -- the next level's messages are read from them on every rank.  level_plan_cls(store, spec, share=,
        mirror=) -> plan (tree.TreeLevelPlan; the CPU tests inject the oracle-backed one)."""
        shares = self.shares(len(spec.cliques))
        owner_rank = {k: r for r, sh in enumerate(shares) for k in sh}
        labels = [[] for _ in range(self.world)]
        for l, k in zip(spec.order, spec.owner):
            labels[owner_rank[k]].append(l)
        return self._exchange_plan(labels, spec.fg.variables.__getitem__, lambda mirror: level_plan_cls(
            self.store, spec, share=shares[self.rank], mirror=mirror))

    def bind_stream(self):
        """launches, collective and scatter on ONE stream -- torch's current stream, which the collective is enqueued on.  Callers that
        launch through the store's context BETWEEN steps (TreeSolver's block operations) bind before their first launch, so the whole
        pass is one in-order queue (rome_ctx_set_stream orders a stream change after the previous stream in any case)."""
        st = None
        if self.device.type == "cuda" and hasattr(self.store, "ctx"):
            st = self.torch.cuda.current_stream(self.device).cuda_stream
            self.store.ctx.set_stream(st)
        return st

    def step(self, plan, opts):
        """up-solve this rank's share, exchange, scatter: afterwards every rank's store holds ALL new frontal beliefs"""
        st = self.bind_stream()
        if plan["up"] is not None:
            plan["up"].run(opts, mirror_out=plan["send"], mirror_stride=plan["U"])
        if self.world > 1 or self.always_collective:
            if self.comm is not None:
                self.comm.all_gather_f64(plan["send"].data_ptr(), plan["recv"].data_ptr(), plan["send"].numel(), st)
            else:
                self.dist.all_gather_into_tensor(plan["recv"], plan["send"].clone())
        if plan["scatter"] is not None:
            plan["scatter"].run(plan["recv"])


class LinearizeShard:
    """Row-sharded `rome_linearize` for the parametric solver (`solveGraphParametric(..., shard=LinearizeShard(...))`): rank k
    evaluates rows shard_range(F, world, k) of every factor kind and an all-gather of the padded (r | Ja | Jb) blocks gives every
    rank the full linearisation, so all ranks take the same Levenberg-Marquardt step.
    On a CUDA device (default kernel) everything between the host's X[ia] gather and the final download stays on the device: the
    factor tables mu / W of the rank's rows are uploaded ONCE, an iteration uploads the gathered coordinates (xa, xb), launches
    `rome_linearize_dev` writing straight into this rank's slice of the receive buffer, all-gathers in place (`comm`: a direct RCCL
    communicator, rome_jl_amd.rccl; else torch.distributed) and downloads the full blocks once.  `stats` accumulates where the time
    went (ms): upload, kernel, exchange, download -- the host sparse solve is timed by solveGraphParametric.
    `kernel` = a per-rank evaluator on HOST arrays (the CPU tests inject a stand-in): the round-trip form."""

    def __init__(self, torch, dist, world, rank, device="cpu", kernel=None, comm=None):
        self.torch, self.dist, self.world, self.rank, self.device = torch, dist, world, rank, torch.device(device)
        self.comm = comm
        self.on_device = kernel is None and self.device.type == "cuda"
        if kernel is None:
            from . import api
            kernel = api.linearize
        self.kernel = kernel
        self.cache = {}
        self.stats = dict(calls=0, upload_ms=0.0, kernel_ms=0.0, exchange_ms=0.0, download_ms=0.0)

    # ---- round 6.  Measured on the box (scripts/pin_cost.py, bench.py parametric_helix10k): a 480 kB upload takes 0.05 ms and a 7.5 MB
    # download 0.15 ms once the host pages exist (pinned or pageable alike; the FIRST touch of a fresh 7.5 MB host buffer costs 8.6 ms),
    # so the transfers were never what linearize_s was made of -- the host's COO -> CSR assembly was (parametric._Problem.csr_perm).
    # A device-side gather X[ia] through torch.index_select (tried first) pays ~80 ms for the lazy load of that operator's code object
    # in a fresh process: the gather of the few thousand rows stays a host numpy take, the result blocks come down into ONE reused buffer.
    def begin(self, X):
        """the coordinates of this linearisation (called once per linearisation by parametric._Problem.linearize)"""
        self._X = X

    def linearize_indexed(self, kind, mu, W, X, ia, ib, ctx=None):
        """as linearize(), the coordinates given as index arrays into the X of begin()"""
        return self.linearize(kind, mu, W, X[ia], None if ib is None else X[ib], ctx)

    def _linearize_dev(self, kind, mu, W, xa, xb, ctx):
        import ctypes as C
        import time
        from . import _lib, api
        torch = self.torch
        ctx = ctx or api.default_context()
        dz, dr, da, db = api._LIN_DIMS[kind]
        F = len(mu)
        lo, hi = shard_range(F, self.world, self.rank)
        n = hi - lo
        q = -(-F // self.world)                                  # rows per rank, padded
        per = dr * (1 + da + db)
        key = int(kind)
        ent = self.cache.get(key)
        # the factor tables of this rank's rows: device-resident across the iterations of ONE solve.  The entry HOLDS the host arrays it
        # was built from (their ids cannot be recycled while it lives) and is valid only for exactly those objects: another problem
        # with the same kind and row count replaces it -- one entry per kind, so repeated / incremental solves do not accumulate tensors.
        if ent is None or ent["mu_host"] is not mu or ent["W_host"] is not W or ent["F"] != F:
            t = lambda a: torch.as_tensor(np.ascontiguousarray(a[lo:hi], dtype=np.float64), device=self.device)   # noqa: E731
            self.cache[key] = dict(mu_host=mu, W_host=W, F=F, pin=None, mu=t(np.asarray(mu).reshape(F, dz)), W=t(np.asarray(W).reshape(F, dr * dr)),
                                   recv=torch.zeros(self.world * q * per, dtype=torch.float64, device=self.device),
                                   xa=torch.empty((max(n, 1), da), dtype=torch.float64, device=self.device),
                                   xb=torch.empty((max(n, 1), max(db, 1)), dtype=torch.float64, device=self.device))
        c = self.cache[key]
        st = torch.cuda.current_stream(self.device)
        ctx.set_stream(st.cuda_stream)
        tick = time.perf_counter
        t0 = tick()
        if n:
            c["xa"][:n].copy_(torch.as_tensor(np.ascontiguousarray(xa[lo:hi], dtype=np.float64)), non_blocking=False)
            if db:
                c["xb"][:n].copy_(torch.as_tensor(np.ascontiguousarray(xb[lo:hi], dtype=np.float64)), non_blocking=False)
        st.synchronize(); t1 = tick()
        send = c["recv"][self.rank * q * per:(self.rank + 1) * q * per]
        base = send.data_ptr()
        if n:
            P = lambda x: C.c_void_p(x)                           # noqa: E731
            _lib.check(_lib.load().rome_linearize_dev(ctx.handle, int(kind), n, P(c["mu"].data_ptr()), P(c["W"].data_ptr()), P(c["xa"].data_ptr()),
                                                      P(c["xb"].data_ptr()) if db else None, P(base), P(base + 8 * q * dr),
                                                      P(base + 8 * q * dr * (1 + da)) if db else None), ctx.handle)
        st.synchronize(); t2 = tick()
        if self.world > 1:
            if self.comm is not None:
                self.comm.all_gather_f64(send.data_ptr(), c["recv"].data_ptr(), send.numel(), st.cuda_stream)
            else:
                self.dist.all_gather_into_tensor(c["recv"], send.clone())
            st.synchronize()
        t3 = tick()
        if c["pin"] is None:
            c["pin"] = torch.empty(c["recv"].numel(), dtype=torch.float64).pin_memory()
        c["pin"].copy_(c["recv"], non_blocking=True); st.synchronize()
        out = c["pin"].numpy().reshape(self.world, q * per)
        t4 = tick()
        S = self.stats
        S["calls"] += 1; S["upload_ms"] += 1e3 * (t1 - t0); S["kernel_ms"] += 1e3 * (t2 - t1); S["exchange_ms"] += 1e3 * (t3 - t2); S["download_ms"] += 1e3 * (t4 - t3)
        cnt = [shard_range(F, self.world, k)[1] - shard_range(F, self.world, k)[0] for k in range(self.world)]
        r = np.concatenate([out[k, :q * dr].reshape(q, dr)[:cnt[k]] for k in range(self.world)])
        Ja = np.concatenate([out[k, q * dr:q * dr * (1 + da)].reshape(q, dr, da)[:cnt[k]] for k in range(self.world)])
        Jb = np.concatenate([out[k, q * dr * (1 + da):].reshape(q, dr, db)[:cnt[k]] for k in range(self.world)]) if db else None
        return r, Ja, Jb

    def linearize(self, kind, mu, W, xa, xb, ctx=None):
        if self.on_device:
            return self._linearize_dev(kind, mu, W, xa, xb, ctx)
        torch = self.torch
        F = len(mu)
        lo, hi = shard_range(F, self.world, self.rank)
        if hi > lo:
            rk, Ja, Jb = self.kernel(kind, mu[lo:hi], W[lo:hi], xa[lo:hi], None if xb is None else xb[lo:hi], ctx=ctx)
            dr, da = Ja.shape[1], Ja.shape[2]
            db = 0 if Jb is None else Jb.shape[2]
        else:   # more ranks than rows: shapes from a one-row probe are not needed, nothing to contribute
            rk = Ja = Jb = None
        # block widths are the same on every rank with rows; agree on them (ranks without rows learn them here)
        dims = torch.tensor([0, 0, 0] if rk is None else [dr, da, db], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(dims, op=self.dist.ReduceOp.MAX)
        dr, da, db = (int(v) for v in dims.tolist())
        per = dr * (1 + da + db)
        q = -(-F // self.world)                                  # rows per rank, padded
        send = torch.zeros(q * per, dtype=torch.float64, device=self.device)
        if rk is not None:
            flat = np.concatenate([rk.reshape(hi - lo, -1), Ja.reshape(hi - lo, -1)] + ([Jb.reshape(hi - lo, -1)] if db else []), axis=1)
            send[:(hi - lo) * per] = torch.as_tensor(flat.ravel(), device=self.device)
        recv = torch.empty(self.world * q * per, dtype=torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(recv, send)
        out = recv.cpu().numpy().reshape(self.world, q, per)
        rows = np.concatenate([out[k, :shard_range(F, self.world, k)[1] - shard_range(F, self.world, k)[0]] for k in range(self.world)])
        r = rows[:, :dr]
        Ja_ = rows[:, dr:dr + dr * da].reshape(F, dr, da)
        Jb_ = rows[:, dr + dr * da:].reshape(F, dr, db) if db else None
        return r, Ja_, Jb_
