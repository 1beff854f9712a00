"""Multi-GPU layer: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

The hot path shards by graph segment / clique sub-tree: every rank owns a contiguous range of the
variables and runs the convolutions that TARGET its variables -- no collective on the data path of a
sweep.  The one real exchange step is the separator message of the Bayes tree (IIF `LikelihoodMessage`
holding a `TreeBelief` of N points per separator variable; SURVEY.md §5 "distributed communication
backend"): after a sweep every rank publishes the beliefs of its boundary variables and receives the
ones its cut factors read (`SeparatorExchange`).  Messages are N x dim doubles (2.4 kB per Pose2).
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

    Every rank contributes `n_sep` belief blocks (each [dim, N]); `plan` lists, for each ghost block this
    rank keeps, which (source rank, source slot) feeds it.  One `all_gather_into_tensor` per exchange:
    the payload is tiny (latency-bound), so a single fixed-size collective beats per-edge send/recv.
    """

    def __init__(self, torch, dist, world, rank, n_sep, dim, N, device, dtype=None):
        self.torch, self.dist = torch, dist
        self.world, self.rank = world, rank
        dtype = dtype or torch.float64
        self.send = torch.zeros((n_sep, dim, N), dtype=dtype, device=device)
        self.recv = torch.zeros((world, n_sep, dim, N), dtype=dtype, device=device)
        self.plan = []  # (ghost_block_index, src_rank, src_slot)

    def add_ghost(self, ghost_block, src_rank, src_slot):
        self.plan.append((int(ghost_block), int(src_rank) % self.world, int(src_slot)))

    def exchange(self, publish, beliefs):
        """publish: list of [dim,N] tensors (this rank's separator beliefs, slot order);
        beliefs: the rank's belief store [V, dim, N]; ghost blocks are overwritten in place."""
        for k, blk in enumerate(publish):
            self.send[k].copy_(blk)
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1))  # flat: same for gloo and nccl
        else:
            self.recv[0].copy_(self.send)
        for ghost, src, slot in self.plan:
            beliefs[ghost].copy_(self.recv[src, slot])


def chain_segment_exchange(torch, dist, world, rank, N, device, ghost_prev, ghost_next):
    """The exchange used by bench.py / the weak-scaling layout: segments in a ring, each rank publishes
    (first pose, last pose) of its segment; ghost_prev <- previous rank's last, ghost_next <- next rank's first."""
    ex = SeparatorExchange(torch, dist, world, rank, n_sep=2, dim=3, N=N, device=device)
    ex.add_ghost(ghost_prev, rank - 1, 1)
    ex.add_ghost(ghost_next, rank + 1, 0)
    return ex
