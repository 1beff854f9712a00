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


class PipelinedSegmentSweep:
    """Weak-scaling sweep driver with the separator exchange fully off the critical path.

    * the sweep kernel itself mirrors this rank's separator proposals into the RCCL send buffer
      (`rome_conv_dev.mirror_*`: no gather kernel);
    * `all_gather_into_tensor` writes straight into ghost blocks that live in the TAIL of the belief store
      (no scatter kernel): the cut factors' `fixed_var`/`target_var` entries simply point there;
    * send and ghost buffers are `depth`-fold buffered (default 2): sweep k reads the separators gathered after sweep k-depth
      while the collectives k-depth+1 .. k-1 are still in flight, so nothing races and the result is deterministic for a fixed
      schedule.  A collective's latency (event joins + RCCL launch, tens of µs) is hidden as long as it is shorter than depth-1 sweeps.
    Host work per step: wait (no-op in steady state) + one kernel launch + one async collective.
    """

    def __init__(self, dg, opts, dist, world, rank, sep_rows, ghost_prev, ghost_next, always_collective=False, depth=2, rccl_comms=None):
        from .factors import Pose2
        torch = dg.torch
        self.dg, self.dist, self.world, self.rank = dg, dist, world, rank
        self.collective = world > 1 or always_collective
        tb = dg.tab["p2p2"]
        N, n_sep = dg.N, len(sep_rows)
        old = dg.bel[Pose2]
        V = old.shape[0]
        per = world * n_sep
        if depth < 2:
            raise ValueError("PipelinedSegmentSweep needs depth >= 2")
        self.depth = D = int(depth)
        store = torch.zeros((V + D * per, 3, N), dtype=old.dtype, device=old.device)
        store[:V].copy_(old)
        dg.bel[Pose2] = store
        self.store, self.V = store, V
        self.recv = [store[V + b * per: V + (b + 1) * per] for b in range(D)]
        self.send = [torch.zeros((n_sep, 3, N), dtype=old.dtype, device=old.device) for _ in range(D)]
        # one proposal table per step slot: consecutive sweeps run on different streams and may overlap
        self.props = [torch.empty((tb["C"], 3, N), dtype=old.dtype, device=old.device) for _ in range(D)]
        self.prop = self.props[0]
        self.works = [None] * D
        self.plans = []
        # direct RCCL form (rccl.create_comms, one communicator per slot): slot b owns stream b and a rome_ctx bound to it; sweep and
        # ncclAllGather are enqueued back to back on that stream, so the collective needs no event join and the host work per step is
        # two C calls.  Otherwise torch.distributed's collective (≈ 30 µs of host time per call) with stream-level joins.
        self.comms = rccl_comms if (rccl_comms and self.collective and old.is_cuda) else None
        self.streams = None
        slot_ctx = [None] * D
        if self.collective and hasattr(torch, "cuda") and old.is_cuda:
            cur = torch.cuda.current_stream(old.device)
            self.streams = [torch.cuda.Stream(old.device) for _ in range(D)]
            for st in self.streams:
                st.wait_stream(cur)
            if self.comms is not None:
                if len(self.comms) != D:
                    raise ValueError("need one RCCL communicator per pipeline slot")
                from . import _lib as _l
                for b in range(D):
                    slot_ctx[b] = _l.Context(old.device.index or 0)
                    slot_ctx[b].set_stream(self.streams[b].cuda_stream)
        for b in range(D):
            gp = V + b * per + ((rank - 1) % world) * n_sep + 1   # previous segment's LAST pose (slot 1)
            gn = V + b * per + ((rank + 1) % world) * n_sep + 0   # next segment's FIRST pose (slot 0)
            store[gp].copy_(old[ghost_prev]); store[gn].copy_(old[ghost_next])
            fixed = tb["fixed"].clone(); target = tb["target"].clone()
            for arr in (fixed, target):
                arr[arr == ghost_prev] = gp
                arr[arr == ghost_next] = gn
            self.plans.append(dg._plan(dg._lib.rome_conv_pose2pose2_dev, opts, n_conv=tb["C"], dir_all=0,
                                       factor=tb["factor"], dir=tb["dir"], fixed_var=fixed, target_var=target,
                                       mu=tb["mu"], L=tb["L"], bel_fixed=store, bel_target=store, out=self.props[b],
                                       n_mirror=n_sep, mirror_row=tuple(int(r) for r in sep_rows), mirror_out=self.send[b],
                                       **({"_ctx": slot_ctx[b]} if slot_ctx[b] is not None else {})))
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
        self.plans[b]()
        if self.collective:
            self.works[b] = self.dist.all_gather_into_tensor(self.recv[b].view(-1), self.send[b].view(-1), async_op=True)
        else:
            self.recv[b].copy_(self.send[b])

    def step(self):
        b = self.k % self.depth
        if self.comms is not None:
            self.plans[b]()                                  # sweep k on stream b (its rome_ctx is bound to it)
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
            cur = self.dg.torch.cuda.current_stream(self.store.device)
            for st in self.streams:
                cur.wait_stream(st)

    def close(self):
        if self.comms is not None:
            self.drain()
            self.dg.torch.cuda.synchronize()
            for c in self.comms:
                c.close()
            self.comms = None


class LinearizeShard:
    """Row-sharded `rome_linearize` for the parametric solver (`solveGraphParametric(..., shard=LinearizeShard(...))`): rank k
    evaluates rows shard_range(F, world, k) of every factor kind and an `all_gather` of the padded (r, Ja, Jb) blocks gives every
    rank the full linearisation, so all ranks take the same Levenberg-Marquardt step.  `kernel` is the per-rank evaluator
    (default: the HIP entry point through `api.linearize`; the CPU tests inject a stand-in)."""

    def __init__(self, torch, dist, world, rank, device="cpu", kernel=None):
        self.torch, self.dist, self.world, self.rank, self.device = torch, dist, world, rank, device
        if kernel is None:
            from . import api
            kernel = api.linearize
        self.kernel = kernel

    def linearize(self, kind, mu, W, xa, xb, ctx=None):
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
