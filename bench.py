#!/usr/bin/env python3
"""bench.py -- factor convolutions/sec (N=100) on a Manhattan-3500-shaped graph.

One "step" = one pass of the hot path over the whole graph: every (factor, direction) Pose2Pose2
convolution (2 x 5453 = 10906) plus the PriorPose2 row in ONE kernel launch, with the belief store
already resident in HBM.  One convolution = what IIF `approxConvBelief` does for one factor
and target: N=100 getSample + inflateCycles(3) x {entropy inflation, 100 per-particle root-finds}.

    python bench.py [--gpus N --steps K --warmup W] [--solver newton|nelder_mead|closed_form]

N>1 (launched by torch.distributed.run, one rank per GPU): weak scaling -- every rank owns one
Manhattan-sized segment of a chain of segments; after each sweep the ranks all-gather their
separator (segment boundary) beliefs over RCCL, exactly the message a Bayes-tree clique boundary
carries (N x 3 doubles per separator variable).  The sweep kernel writes the separator rows straight
into the send buffer, the collective lands straight in ghost blocks of the belief store, both
double-buffered so that collective k-1 overlaps sweep k (rome_jl_amd.distributed.PipelinedSegmentSweep).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# algorithmic HBM bytes per Pose2Pose2 particle root-find (in-kernel RNG: no noise read):
#   closed_form / newton : fixed pose 3 + solution 3 doubles = 48   (the unique root does not depend on the start point: u0 is NOT read)
#   gauss_newton / nelder_mead (start from the belief point): fixed 3 + start point u0 3 + solution 3 = 72
BYTES_PER_PARTICLE_P2P2 = {"closed_form": 48, "newton": 48, "gauss_newton": 72, "nelder_mead": 72}
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~0.1 s of untimed sweeps before ~0.1 s of timed ones -- the device needs ≈ 50-100 ms of back-to-back launches to reach
    # its steady state (with 20 warm-up launches the same kernel reads 50 µs per launch instead of 46 µs)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--solver", default="newton", choices=["closed_form", "newton", "nelder_mead", "gauss_newton"])
    ap.add_argument("--tree-messages", default="relative", choices=["relative", "marginal"], help="message form of the solve.from_tree_clique_forms leg")
    ap.add_argument("--tree-structures", type=int, default=1, help="elimination structures pooled by the solve.from_tree leg (1: one plan set)")
    ap.add_argument("--poses", type=int, default=3500)
    ap.add_argument("--loops", type=int, default=1954)
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--g2o", default=None, help="g2o file (default: tests/golden/manhattan.g2o, the M3500 dataset the reference "
                                                 "ships as examples/manhattan.g2o; 'synthetic' forces the generator)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = one Manhattan-sized segment per GPU + separator all-gather (default); strong = ONE Manhattan graph, "
                         "convolutions sharded by target ownership + all-gather of the owned beliefs (rome_jl_amd.distributed.TargetShardedSweep)")
    ap.add_argument("--dry-run", action="store_true", help="no device, no process group: build every rank's tables / arena / exchange plan on "
                                                          "CPU tensors and check them against each other (what an N-GPU run relies on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the extra per-solver throughput runs")
    ap.add_argument("--cpu-seconds", type=float, default=40.0)
    ap.add_argument("--settle-launches", type=int, default=4000,
                    help="untimed launches BEFORE the --warmup steps so that the device reaches its steady clocks whatever W is "
                         "(the first ~50-100 ms of back-to-back launches run 20-30 %% slower); 0 disables")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def rank_graph(R, args, rank, multi, strong, N):
    """this rank's workload: the Manhattan graph (weak scaling, N > 1: one Manhattan-sized segment per rank + ghost separators of the
    neighbouring segments) -> (fg, workload description, label of the last pose)"""
    default_g2o = os.path.join(ROOT, "tests", "golden", "manhattan.g2o")
    if args.g2o is None and os.path.exists(default_g2o) and (args.poses, args.loops) == (3500, 1954):
        args.g2o = default_g2o
    if args.g2o and args.g2o != "synthetic":
        fg = R.loadG2o(args.g2o, N=N)
        workload = "Manhattan-3500 (M3500 pose graph, %s: the data file the reference ships as examples/manhattan.g2o; prior on x0 as " \
                   "examples/ManhattanDatasetBatch.jl:31)" % os.path.relpath(args.g2o, ROOT) if os.path.abspath(args.g2o) == default_g2o \
            else "g2o:%s" % os.path.basename(args.g2o)
        args.poses = sum(1 for t in fg.variables.values() if t is R.Pose2)
    else:
        fg = R.synth_manhattan(P=args.poses, loops=args.loops, seed=0x524F4D45 + rank, N=N)
        workload = "synth_manhattan(P=%d, loops=%d) [g2o-shaped stand-in for examples/manhattan.g2o]" % (args.poses, args.loops)
    last = "x%d" % (sum(1 for t in fg.variables.values() if t is R.Pose2) - 1)
    if multi and not strong:
        # cut edges to the neighbouring segments: ghost variables hold the neighbours' separator beliefs
        cov = np.diag([1 / 44.6, 1 / 399.0, 1 / 9591.0])
        fg.addVariable("ghost_prev", R.Pose2); fg.addVariable("ghost_next", R.Pose2)
        fg.addFactor(["ghost_prev", "x0"], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], cov)))
        fg.addFactor([last, "ghost_next"], R.Pose2Pose2(R.MvNormal([1.0, 0.0, 0.0], cov)))
    return fg, workload, last


def pipe_depth(world):
    """steps in flight of the weak-scaling pipeline (a ghost belief is `depth` steps old): the step period is
    max(sweep, (sweep + exchange) / depth).  Each slot owns a stream, a send / receive buffer and a communicator -- and a HIP stream
    maps to one of FOUR hardware queues: measured with one rank and the exchange forced (round 4), depth 2 reads 9.4 us per step,
    depth 4 8.0 us, depth 8 12.7 us (eight streams multiplexed onto four queues).  Four it is, for every rank count."""
    return int(os.environ.get("ROME_PIPE_DEPTH", "4"))


def separator_rows(pk, args, last):
    """proposal rows that carry the updated separator estimates of a segment (odometry convolutions targeting its first / last pose)"""
    vf, vt = pk.p2p2["var_from"], pk.p2p2["var_to"]
    f_first = int(np.nonzero((vf == pk.index["x0"]) & (vt == pk.index["x1"]))[0][0])
    f_last = int(np.nonzero((vf == pk.index["x%d" % (args.poses - 2)]) & (vt == pk.index[last]))[0][0])
    return 2 * f_first + 1, 2 * f_last + 0      # odometry x0->x1, dir 1 -> target x0 ; odometry x_{P-2}->x_{P-1}, dir 0 -> target x_{P-1}


def dry_run(args):
    """`bench.py --gpus N --dry-run`: NO device, NO process group -- every rank's tables, arena layout and exchange plan are built on
    CPU tensors (DeviceGraph(plan_only=True)) exactly as the N-rank run builds them, and checked against each other: what the 8-GPU run
    relies on being consistent across ranks is verified on a machine that cannot make that run.  Prints one JSON line."""
    import torch
    import rome_jl_amd as R
    from rome_jl_amd.distributed import PipelinedSegmentSweep, TargetShardedSweep, FrontierShard
    world, N = args.gpus, args.particles
    checks, fails = {}, []

    def check(name, ok, detail=None):
        checks[name] = bool(ok)
        if not ok:
            fails.append("%s: %s" % (name, detail))
    depth = pipe_depth(world)
    # ---- weak scaling: a chain of Manhattan-sized segments, separator all-gather
    pipes = []
    for rank in range(world):
        fg, workload, last = rank_graph(R, args, rank, True, False, N)
        R.dead_reckon_init(fg, seed=11 + rank)
        dg = R.DeviceGraph(fg, plan_only=True); dg.upload_beliefs(fg)
        pk = dg.packed
        opts = R.make_opts(N=N, seed=0x524F4D45, stream_offset=rank * (1 << 32))
        rows = separator_rows(pk, args, last)
        pipes.append((PipelinedSegmentSweep(dg, opts, None, world, rank, list(rows), pk.index["ghost_prev"], pk.index["ghost_next"],
                                            always_collective=False, depth=depth), dg, rows))
    p0 = pipes[0][0]
    check("weak.payload_equal_on_all_ranks", all(p.payload == p0.payload for p, _, _ in pipes), [p.payload for p, _, _ in pipes])
    check("weak.send_recv_sizes", all(all(s.numel() == p.payload for s in p.send) and all(r.numel() == world * p.payload for r in p.recv) for p, _, _ in pipes))
    for rank, (p, dg, rows) in enumerate(pipes):
        pk = dg.packed
        nblk = p.store[R.Pose2].shape[0]
        for b in range(depth):
            st = p.plans[b][0]                       # the Pose2Pose2 family's launch descriptor of slot b
            r4 = st.kw["rows4"].numpy()
            check("weak.rank%d.slot%d.rows_inside_the_arena" % (rank, b), r4[:, 2:].min() >= 0 and r4[:, 2:].max() < nblk, (int(r4[:, 2:].max()), nblk))
            mm = st.kw["mirror_map"].numpy()
            check("weak.rank%d.slot%d.mirror_slots" % (rank, b), sorted(mm[mm >= 0].tolist()) == [0, 1] and mm[rows[0]] == 0 and mm[rows[1]] == 1, mm[mm >= 0].tolist())
            # the ghost variables of this slot are blocks INSIDE receive buffer b, at the neighbour's published slot
            U = 3 * N
            base = (p.recv[b].data_ptr() - p.arena.data_ptr()) // 8 - (p.store[R.Pose2].data_ptr() - p.arena.data_ptr()) // 8
            want_prev = (base + ((rank - 1) % world) * p.payload + 1 * U) // U
            want_next = (base + ((rank + 1) % world) * p.payload + 0 * U) // U
            gp, gn = pk.index["ghost_prev"], pk.index["ghost_next"]
            orig = dg.tab["p2p2"]["rows4"].numpy()
            got_prev = set(r4[:, 2][orig[:, 2] == gp].tolist()) | set(r4[:, 3][orig[:, 3] == gp].tolist())
            got_next = set(r4[:, 2][orig[:, 2] == gn].tolist()) | set(r4[:, 3][orig[:, 3] == gn].tolist())
            check("weak.rank%d.slot%d.ghost_blocks" % (rank, b), got_prev == {want_prev} and got_next == {want_next}, (got_prev, want_prev, got_next, want_next))
    # ---- strong scaling: ONE graph, rows sharded by target ownership
    fg, _, _ = rank_graph(R, args, 0, True, True, N)
    R.dead_reckon_init(fg, seed=11)
    dg = R.DeviceGraph(fg, plan_only=True); dg.upload_beliefs(fg)
    shards = [TargetShardedSweep(dg, R.make_opts(N=N, seed=0x524F4D45), None, world, r) for r in range(world)]
    n_rows = dg.tab["p2p2"]["C"]
    check("strong.row_ranges_partition_the_table", shards[0].row_lo == 0 and shards[-1].row_hi == n_rows and
          all(shards[r].row_hi == shards[r + 1].row_lo for r in range(world - 1)), [(s_.row_lo, s_.row_hi) for s_ in shards])
    check("strong.every_rank_targets_only_its_variables", all(
        (lambda t, r: len(t) == 0 or (t.min() >= r * shards[r].q and t.max() < (r + 1) * shards[r].q))(shards[r].rows4[shards[r].row_lo:shards[r].row_hi, 3].numpy(), r)
        for r in range(world)))
    check("strong.stream_offset_is_the_row_position", all(int(shards[r].plan.opts.stream_offset) == shards[r].row_lo and
                                                          shards[r].plan.kw["n_conv"] == shards[r].n_rows for r in range(world)))
    check("strong.store_holds_every_rank's_block", shards[0].store.shape[0] == world * shards[0].q >= dg.bel[R.Pose2].shape[0])
    # ---- the clique frontier dealt to the ranks (FrontierShard): shares, exchange blocks, scatter lists
    nbr = {l: set() for l in fg.variables}
    for _, labels, _ in fg.factors:
        for a in labels:
            nbr[a].update(b for b in labels if b != a)
    chosen, blocked = [], set()
    for l in fg.variables:
        if l not in blocked:
            chosen.append(l); blocked.add(l); blocked.update(nbr[l])
    cliques = [[l] for l in chosen]

    class _Rec:
        def __init__(self, *a, **kw):
            self.a, self.kw = a, kw

    class _Store:
        N = args.particles
    _Store.fg = fg

    seen_blocks, all_updated = {}, []
    plans = []
    for rank in range(world):
        sh = FrontierShard(_Store(), torch, None, world, rank, plan_cls=_Rec, scatter_cls=_Rec)
        plans.append(sh.plan(cliques, gibbsIters=3))
    widths = {pl["width"] for pl in plans}
    check("frontier.width_equal_on_all_ranks", len(widths) == 1, widths)
    width = plans[0]["width"]
    for rank, pl in enumerate(plans):
        mine = pl["labels"][rank]
        all_updated += mine
        if pl["up"] is not None:
            mir = pl["up"].kw["mirror"]     # PACKED layout: slots of N doubles, a Pose2 belief takes 3 consecutive slots
            check("frontier.rank%d.mirror_slots" % rank, sorted(mir.values()) == list(range(0, 3 * len(mine), 3)) and max(mir.values()) + 3 <= width)
            check("frontier.rank%d.share" % rank, pl["up"].kw["share"] == list(range(rank, len(cliques), world)))
        if pl["scatter"] is not None:
            ls, bl = pl["scatter"].a[1], pl["scatter"].a[2]
            want = [(l, r * width + 3 * k) for r in range(world) if r != rank for k, l in enumerate(pl["labels"][r])]
            check("frontier.rank%d.scatter_list" % rank, list(zip(ls, bl)) == want)
        check("frontier.rank%d.buffers" % rank, pl["recv"].numel() == world * width * N and pl["send"].numel() == width * N and
              pl["send"].data_ptr() == pl["recv"].data_ptr() + rank * width * N * 8 and pl["U"] == N)
    check("frontier.shares_partition_the_frontier", sorted(all_updated) == sorted(chosen))
    out = {"dry_run": True, "n_gpus": world, "depth": depth, "checks": len(checks), "failed": fails, "ok": not fails,
           "weak": {"payload_doubles": p0.payload, "arena_doubles_per_rank": p0.arena.numel()},
           "strong": {"rows_per_rank": [s_.n_rows for s_ in shards], "variables_per_rank": shards[0].q},
           "frontier": {"cliques": len(cliques), "width_slots": width, "exchange_bytes_per_rank": width * N * 8,
                        "layout": "packed: slots of N doubles, Pose2 3 / Point2 2 / Pose3 6 slots per belief (round 4 padded every block to 6 N)"}}
    print(json.dumps(out), flush=True)
    return 0 if not fails else 1


def main():
    args = parse()
    if args.dry_run:
        raise SystemExit(dry_run(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)   # does not return
    import torch
    import torch.distributed as dist
    import rome_jl_amd as R

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # ROME_BENCH_SHARED_DEVICE=1 (tests only, never a measurement): all ranks launch on device 0, the process group is gloo and the
    # separator / belief exchange goes through rome_jl_amd.rccl.HostStagedComm -- RCCL refuses two ranks on one device, and this is
    # how the N > 1 flow of this file (per-rank tables, pipeline, barriers, max-over-ranks timing, rank 0's JSON line) runs as N
    # real processes on a one-GPU box (tests/test_gpu_zz_ranks_one_device.py).  The line it prints says so.
    shared = os.environ.get("ROME_BENCH_SHARED_DEVICE") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # ROME_BENCH_FORCE_EXCHANGE=1: exercise the multi-GPU code path (process group, ghost variables,
    # separator all_gather) even with a single rank -- used to smoke-test the RCCL path on a 1-GPU box
    multi = world > 1 or os.environ.get("ROME_BENCH_FORCE_EXCHANGE") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != world:
            raise SystemExit("bench.py: process group has %d ranks, expected %d" % (dist.get_world_size(), world))
    if multi:
        # a collective that never completes (a rank died, an interconnect problem) must not hang the box: give up loudly
        import threading
        limit = float(os.environ.get("ROME_BENCH_WATCHDOG_S", "600"))

        def _bail():
            print("bench.py: rank %d: multi-GPU run did not finish within %.0f s -- aborting" % (rank, limit), file=sys.stderr, flush=True)
            os._exit(124)
        wd = threading.Timer(limit, _bail); wd.daemon = True; wd.start()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a number "
                         "for a different GPU count" % (args.gpus, world))
    if not shared and torch.cuda.device_count() < (args.gpus if "LOCAL_WORLD_SIZE" not in os.environ else int(os.environ["LOCAL_WORLD_SIZE"])):
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) are visible" % (args.gpus, torch.cuda.device_count()))

    N = args.particles
    solver = {"closed_form": R.SOLVER_CLOSED_FORM, "newton": R.SOLVER_NEWTON, "nelder_mead": R.SOLVER_NELDER_MEAD,
              "gauss_newton": R.SOLVER_GAUSS_NEWTON}[args.solver]

    strong = multi and args.scaling == "strong"
    fg, workload, last = rank_graph(R, args, rank, multi, strong, N)
    R.dead_reckon_init(fg, seed=11 + (0 if strong else rank))   # strong scaling: every rank holds the SAME graph and beliefs
    ctx = R.Context(local)
    dg = R.DeviceGraph(fg, device=dev, ctx=ctx)
    dg.upload_beliefs(fg)
    tb = dg.tab["p2p2"]
    n_conv_step = tb["C"]  # convolutions per step on this rank: 2 x F relative + the PriorPose2 row(s), ONE launch
    opts = R.make_opts(N=N, solver=solver, seed=0x524F4D45, stream_offset=0 if strong else rank * (1 << 32))
    prop = torch.empty((tb["C"], 3, N), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)   # the context follows torch's current stream, set ONCE
    sweep = dg.plan_sweep_pose2pose2(opts, prop, fixed_ctx=ctx)   # pre-built launch descriptor: one C call = one hipLaunchKernel per step

    pk = dg.packed
    pipe = None
    comms = None
    depth = 0
    strong_unit = "convolution sweep"
    if strong:
        from rome_jl_amd.distributed import TargetShardedSweep
        comm = None
        if shared:
            from rome_jl_amd.rccl import HostStagedComm
            comm = HostStagedComm(torch, dist, world, ctx)
        elif os.environ.get("ROME_BENCH_TORCH_COLLECTIVE") != "1":
            try:
                from rome_jl_amd.rccl import create_comms
                cc = create_comms(torch, dist, world, rank, dev, 1)
                comm = cc[0] if cc else None
            except Exception as e:   # noqa: BLE001
                print("direct RCCL binding unavailable (%r): using torch.distributed collectives" % (e,), file=sys.stderr)
        pipe = TargetShardedSweep(dg, opts, dist, world, rank, always_collective=(world == 1), rccl_comm=comm)
        comms = [comm] if comm else None
        n_conv_step = tb["C"]          # the whole graph per step, over all ranks
        # the unit that shards is the SOLVE iteration (sweep of the owned rows + manikde! bandwidths + multiscale Gibbs product of the
        # owned variables, ~2.9 ms on one GPU), followed by ONE all-gather of the changed beliefs; a conv-only step (8 µs) would be
        # all exchange.  ROME_BENCH_STRONG_CONV_ONLY=1 times that conv-only step instead (context).
        if os.environ.get("ROME_BENCH_STRONG_CONV_ONLY") == "1":
            sweep = pipe.step
        else:
            _it = [0]

            def sweep():
                pipe.solve_step(opts, sweep=_it[0]); _it[0] += 1
            args.settle_launches = min(args.settle_launches, 50)
            strong_unit = "solve iteration"
    elif multi:
        from rome_jl_amd.distributed import PipelinedSegmentSweep
        conv_first, conv_last = separator_rows(pk, args, last)
        # pipeline depth = steps in flight (a ghost belief is `depth` steps old): the step period is max(sweep, (sweep + exchange) /
        # depth).  One rank, exchange forced: 12.6–13.7 µs per step for every depth 2..6 (profiles/r02_bench_n1_forced_exchange.json); a
        # ring all-gather over more than two GPUs costs several sweeps of latency, so four slots there
        depth = pipe_depth(world)
        # separator exchange through RCCL directly (one communicator per pipeline slot, enqueued on the sweep's own stream);
        # torch.distributed's collective is the fallback if the direct binding cannot be set up on every rank
        comms = None
        if shared:
            from rome_jl_amd.rccl import HostStagedComm
            comms = [HostStagedComm(torch, dist, world, ctx) for _ in range(depth)]
        elif os.environ.get("ROME_BENCH_TORCH_COLLECTIVE") != "1":
            try:
                from rome_jl_amd.rccl import create_comms
                comms = create_comms(torch, dist, world, rank, dev, depth)
            except Exception as e:   # noqa: BLE001  (create_comms agrees on failure across ranks itself; this is the last resort)
                print("direct RCCL binding unavailable (%r): using torch.distributed collectives" % (e,), file=sys.stderr)
                comms = None
        pipe = PipelinedSegmentSweep(dg, opts, dist, world, rank, [conv_first, conv_last],
                                     pk.index["ghost_prev"], pk.index["ghost_next"], always_collective=(world == 1),
                                     depth=depth, rccl_comms=comms)
        sweep = pipe.step

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # device settle (untimed, not part of the W warm-up steps): the first 50-100 ms of back-to-back launches run at ramping clocks
    if args.solver != "nelder_mead":
        for _ in range(max(0, args.settle_launches - args.warmup)):
            sweep()
    for _ in range(args.warmup):
        sweep()
    barrier()
    # per-launch duration of the dominant kernel: HIP events on the launch stream (torch's current stream, which the
    # launch plan binds the rome_ctx to) inside the timed region.  An event is a barrier packet in the queue (≈ 5 µs of
    # dispatch gap each on this stack), so they are recorded every `stride` launches, not around every launch:
    # consecutive events bracket `stride` back-to-back launches of the dominant kernel [+ the separator exchange when N>1].
    # (short runs -- the driver's --steps 20 -- bracket the whole timed region with two events only: 21 events around 20 launches would
    #  add their own dispatch gaps to the period they measure)
    stride = max(1, args.steps // 20) if args.steps >= 200 else args.steps
    marks = list(range(0, args.steps, stride))

    def timed_block(with_events=True):
        """EXACTLY args.steps steps between barrier + synchronize on both sides -> (wall seconds, max over ranks; event ms per step)"""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(marks) + 1)] if with_events else None
        barrier()
        t0 = time.perf_counter()
        if with_events:
            j = 0
            for k in range(args.steps):
                if j < len(marks) and k == marks[j]:
                    ev[j].record(); j += 1
                sweep()
            ev[len(marks)].record()
        else:
            for k in range(args.steps):
                sweep()
        if strong:
            pipe.wait()
        elif multi:
            pipe.drain()
        barrier()
        t1 = time.perf_counter()
        el = t1 - t0
        if multi:
            tt = torch.tensor([el], dtype=torch.float64, device="cpu" if shared else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        # N>1: even and odd steps run on side streams, which events on the caller's stream do not bracket: the launch period is
        # then this rank's wall-clock between the barriers
        km = 1e3 * (t1 - t0) / args.steps if (multi or not with_events) else float(ev[0].elapsed_time(ev[len(marks)])) / args.steps
        return el, km

    # short runs (the driver's --steps 20: a 0.17 ms region in which every host call between the two synchronizes counts --
    # scripts/short_region_cost.py / profiles/r04_short_region_cost.txt: an event record is 4 us of host time and a barrier packet in the
    # queue): the block of K steps is repeated and the MEDIAN block is reported (every block is exactly K steps between barrier +
    # synchronize; all block times are listed).  The blocks behind `value` carry no event records; the launch period of the dominant
    # kernel for the roofline comes from one more block of the same K steps WITH the two HIP events around it.
    n_blocks = 1 if args.steps >= 1000 else (5 if args.steps >= 100 else 9)
    if n_blocks > 1:
        timed_block()   # one UNTIMED block first: the first block after the barrier is cold (14.5 us per step against 9.5-11 in round 3)
        ev_block = timed_block(True)
        blocks = [timed_block(False) for _ in range(n_blocks)]
    else:
        blocks = [timed_block(True)]
        ev_block = blocks[0]
    order = sorted(range(n_blocks), key=lambda i: blocks[i][0])
    elapsed, _ = blocks[order[n_blocks // 2]]
    # ONE period for `value` and for the roofline: the median block (N=1: a step IS one launch of the dominant kernel; the event-bracketed
    # block is kept beside it as a cross-check -- round 4 quoted the roofline on that block alone, 7.8 us against a reproducible 8.6)
    kern_ms_events = ev_block[1]
    kern_ms = 1e3 * elapsed / args.steps

    data_kind = "synthetic" if (not args.g2o or args.g2o == "synthetic") else \
        "Manhattan M3500 dataset (measurements); beliefs synthetic: dead-reckoned means + N(0, sigma) particles"
    total_conv = n_conv_step * (1 if strong else world) * args.steps
    value = total_conv / elapsed
    alg_bytes = tb["C_rel"] * N * BYTES_PER_PARTICLE_P2P2[args.solver] + tb["P"] * N * 24
    if strong:
        alg_bytes = alg_bytes * pipe.n_rows // max(1, tb["C"])   # this rank's launch covers its own row range
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    out = {
        "metric": "factor convolutions/sec (N=100) on Manhattan-3500; solveTree! wall-clock",
        "value": value, "unit": "convolutions/s", "n_gpus": (dist.get_world_size() if multi else 1), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "timed_blocks_ms_per_step": [1e3 * b[0] / args.steps for b in blocks],
        "timed_block": "median of %d blocks of exactly %d steps, each between barrier + synchronize%s" % (n_blocks, args.steps, ", after one untimed block; kernel_ms_per_launch from one more such block bracketed by two HIP events" if n_blocks > 1 else ""), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f64", "data": data_kind if not shared else
        data_kind + " [ROME_BENCH_SHARED_DEVICE=1: %d ranks on ONE device, host-staged exchange -- a flow test, NOT a measurement]" % world,
        "config": {"workload": workload, "poses_per_gpu": len(pk.labels[R.Pose2]), "pose2pose2_factors_per_gpu": tb["F"],
                   "convolutions_per_step_per_gpu": (pipe.n_rows if strong else n_conv_step), "particles": N,
                   "solver": {"newton": "newton (analytic root: k_conv_flat<P2P2, 0>; no residual evaluation, no start point read, inflate_cycles inert)",
                              "closed_form": "closed_form (analytic root: k_conv_flat<P2P2, 0>; inflate_cycles inert)",
                              "gauss_newton": "gauss_newton (Gauss-Newton on the residual functor from the belief point: k_conv_flat<P2P2, 3>; one pass, inflate_cycles inert on unique-root factors)",
                              "nelder_mead": "nelder_mead (the reference's Optim.NelderMead on the residual functor, inflate_cycles x entropy + solve)"}[args.solver],
                   "inflate_cycles": int(opts.inflate_cycles), "inflation": float(opts.inflation), "noise": "in-kernel philox",
                   "parallelism": ("ONE graph, one %s per step: rows sharded by target ownership (%d of %d on rank 0), all_gather of the owned belief blocks (%s)"
                                   % (strong_unit, pipe.n_rows, tb["C"], ("host-staged (shared device)" if shared else "RCCL direct, in place") if comms else "torch.distributed")) if strong else
                                  (("1 graph segment per GPU, separator all_gather (%s, pipeline depth %d)" % (("host-staged (shared device)" if shared else "RCCL direct") if comms else "torch.distributed", depth))
                                   if multi else "single GPU"),
                   "ranks_seen_by_rccl": (dist.get_world_size() if multi else 1)},
        "roofline": {"bound": "hbm", "kernel": ("rome::k_conv_flat<P2P2> (packed unique-root sweep)" if args.solver in ("newton", "closed_form")
                                                else ("rome::k_conv_flat<P2P2, gauss_newton> (packed sweep, functor iteration)" if args.solver == "gauss_newton"
                                                      else "rome::k_conv<P2P2,%s,PPL=2,lean>" % args.solver)),
                     "bytes_per_particle": BYTES_PER_PARTICLE_P2P2[args.solver],
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_per_launch": kern_ms,
                     "period": "ms_per_step of the median timed block (the period behind `value`)", "kernel_ms_event_bracketed_block": kern_ms_events,
                     "traffic": None},
    }
    # counter-measured HBM traffic of the same kernel on the same table (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    # scripts/profile_round.sh; PMC counters need rocprofv3 around the process, so this run quotes the stored pass)
    tfile = os.path.join(ROOT, "profiles", "r06_hbm_traffic_%s.json" % args.solver)
    if os.path.exists(tfile) and world == 1:
        try:
            with open(tfile) as f:
                tj = json.load(f)
            if tj.get("solver") == args.solver and tj.get("n_conv") == tb["C"]:
                out["roofline"]["traffic"] = tj.get("bytes_per_launch")
                out["roofline"]["traffic_kind"] = "stored"
                out["roofline"]["traffic_source"] = tj.get("source")
                # both fractions side by side: algorithmic bytes / period (frac) and counter-measured HBM bytes / period
                out["roofline"]["hbm_measured"] = {"achieved": tj["bytes_per_launch"] / (kern_ms * 1e-3) / 1e9, "unit": "GB/s",
                                                   "frac": tj["bytes_per_launch"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                   "read_bytes": tj.get("read_bytes_per_launch"), "write_bytes": tj.get("write_bytes_per_launch")}
        except Exception:
            pass

    sfile = os.path.join(ROOT, "profiles", "r06_sq_counters_%s.json" % args.solver)
    if os.path.exists(sfile) and world == 1:
        try:
            with open(sfile) as f:
                sj = json.load(f)
            out["roofline"]["valu_instructions_per_wave_pmc"] = sj["derived"]["valu_instructions_per_wave"]
            # secondary bound: VALU issue.  Busy quad-cycles of the PMC pass (a property of the instruction stream, same
            # kernel, same table) against the SIMD cycles available in THIS run's measured launch period at 2.4 GHz
            busy_cycles = 4.0 * sj["SQ_ACTIVE_INST_VALU_quadcycles"]
            clk = 2.4e9
            avail = 256 * 4 * clk * kern_ms * 1e-3
            out["roofline"]["secondary"] = {"bound": "valu_issue", "achieved": busy_cycles / (kern_ms * 1e-3) / 1e12,
                                            "peak": 256 * 4 * clk / 1e12, "unit": "T SIMD-cycles/s", "frac": busy_cycles / avail,
                                            "source": "SQ_ACTIVE_INST_VALU (profiles/r06_sq_counters_%s.json) / (256 CUs x 4 SIMDs x 2.4 GHz x launch period of this run)" % args.solver}
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_modes:
        modes, by_solver = {}, {}
        for name, sv in (("closed_form", R.SOLVER_CLOSED_FORM), ("newton", R.SOLVER_NEWTON), ("gauss_newton", R.SOLVER_GAUSS_NEWTON),
                         ("nelder_mead", R.SOLVER_NELDER_MEAD)):
            o2 = R.make_opts(N=N, solver=sv, seed=0x524F4D45)
            reps = 20 if sv == R.SOLVER_NELDER_MEAD else 1000
            pl = dg.plan_sweep_pose2pose2(o2, prop)
            pl(); torch.cuda.synchronize()
            a = time.perf_counter()
            for _ in range(reps):
                pl()
            torch.cuda.synchronize()
            dt_l = (time.perf_counter() - a) / reps
            modes[name] = tb["C"] / dt_l
            bts = tb["C_rel"] * N * BYTES_PER_PARTICLE_P2P2[name] + tb["P"] * N * 24
            by_solver[name] = {"bytes_per_particle": BYTES_PER_PARTICLE_P2P2[name], "algorithmic_bytes_per_launch": bts, "kernel_ms_per_launch": 1e3 * dt_l,
                               "achieved": bts / dt_l / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bts / dt_l / 1e9 / HBM_PEAK_GBS, "bound": "hbm",
                               "launches_timed": reps}
            sqf = os.path.join(ROOT, "profiles", "r06_sq_counters_%s.json" % name)
            if os.path.exists(sqf):   # secondary bound of the functor-iterating solvers: VALU issue (stored PMC pass of the same kernel and table)
                try:
                    with open(sqf) as f:
                        sj = json.load(f)
                    by_solver[name]["valu_busy_frac"] = 4.0 * sj["SQ_ACTIVE_INST_VALU_quadcycles"] / (256 * 4 * 2.4e9 * dt_l)
                    by_solver[name]["valu_instructions_per_wave"] = sj["derived"]["valu_instructions_per_wave"]
                except Exception:
                    pass
        # the residual-EVALUATING default solver: NEWTON with a status array (k_conv_flat<P2P2, VERIFY>): the analytic root, then the RoME
        # residual functor (through points / rotation matrices) at every particle for the convergence status -- 48 B + 4 B status per particle
        status = torch.zeros((tb["C"], N), dtype=torch.int32, device=dev)
        plv = dg.plan_sweep_pose2pose2(R.make_opts(N=N, solver=R.SOLVER_NEWTON, seed=0x524F4D45), prop, status=status)
        plv(); torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(1000):
            plv()
        torch.cuda.synchronize()
        tv = (time.perf_counter() - a) / 1000
        modes["newton_with_status"] = tb["C"] / tv
        out["newton_with_status"] = {"what": "NEWTON + status array: analytic root, then the residual functor evaluated at every particle (k_conv_flat<P2P2, VERIFY=true>)",
                                     "us_per_sweep": 1e6 * tv, "unconverged_particles": int(status.sum().item()),
                                     "algorithmic_GBps": (tb["C_rel"] * N * 52 + tb["P"] * N * 28) / tv / 1e9,
                                     "frac_of_hbm_peak": (tb["C_rel"] * N * 52 + tb["P"] * N * 28) / tv / 1e9 / HBM_PEAK_GBS, "bytes_per_particle": 52}
        out["gpu_convolutions_per_s_by_solver"] = modes
        # north_star's literal path is "residual + numerical root-find": gauss_newton (functor iteration) and nelder_mead (the reference's
        # optimizer) beside the analytic-root headline, each with its own algorithmic bytes (start point read: 72 B) and measured period
        out["roofline_by_solver"] = by_solver
        # the other half of the metric ("solveTree! wall-clock"): one iteration of the device-resident solve loop on the
        # same graph = all convolutions (one launch) + the proposal product of every variable (one launch); DESIGN.md §11
        o3 = R.make_opts(N=N, solver=R.SOLVER_NEWTON, seed=0x524F4D45)
        saved = dg.bel[R.Pose2].clone()
        dg.conv_step(o3, 0); dg.product_step(o3, 0); torch.cuda.synchronize()
        a = time.perf_counter()
        for s in range(50):
            dg.conv_step(o3, s); dg.product_step(o3, s)
        torch.cuda.synchronize()
        out["solve_loop"] = {"ms_per_iteration": (time.perf_counter() - a) * 1e3 / 50,
                             "what": "conv sweep + importance-sampling product (round-1 stand-in, in-kernel Silverman bandwidths) of all %d variables" % len(pk.labels[R.Pose2])}
        # the same iteration with the reference's bandwidth rule: leave-one-out likelihood bandwidths of every proposal (manikde!) first
        dg.conv_step(o3, 0); dg.product_step(o3, 0, "lcv"); torch.cuda.synchronize()
        a = time.perf_counter()
        for s in range(5):
            dg.conv_step(o3, s); dg.product_step(o3, s, "lcv")
        torch.cuda.synchronize()
        out["solve_loop"]["ms_per_iteration_lcv_bandwidths"] = (time.perf_counter() - a) * 1e3 / 5
        dg.bel[R.Pose2].copy_(saved)
        # ... and with the reference's own product as well: manikde! bandwidths + multiscale Gibbs product (AMP manifoldProduct)
        dg.conv_step(o3, 0); dg.product_step(o3, 0, "lcv", "gibbs"); torch.cuda.synchronize()
        a = time.perf_counter()
        for s in range(5):
            dg.conv_step(o3, s); dg.product_step(o3, s, "lcv", "gibbs")
        torch.cuda.synchronize()
        out["solve_loop"]["ms_per_iteration_reference_product"] = (time.perf_counter() - a) * 1e3 / 5
        out["solve_loop"]["what_reference_product"] = "conv sweep + manikde! bandwidths of all proposals + multiscale Gibbs product (manifoldProduct restated) of all variables; whole-graph Jacobi schedule, no Bayes tree"
        dg.bel[R.Pose2].copy_(saved)
        # ---- solve-level number (the other half of the metric): run to a stated convergence criterion with the reference's
        # operations (conv sweep + manikde! bandwidths + multiscale Gibbs product per iteration, whole-graph Jacobi schedule).
        # Start: the parametric solution (IIF initialises the non-parametric solve from solveGraphParametric: initParametricFrom!).
        # Criterion: the RMS distance of the pose means to the parametric solution changes by < 1e-3 m over a block of 5 iterations.
        def pose_means():
            m, _ = dg.belief_stats(R.Pose2)
            return m.cpu().numpy()[:len(pk.labels[R.Pose2])]
        a = time.perf_counter(); xp = R.solveGraphParametric(fg); t_par = time.perf_counter() - a
        mp = np.array([xp[l] for l in pk.labels[R.Pose2]])

        def rms_to_parametric():
            d = pose_means() - mp
            return float(np.sqrt(np.mean(np.sum(d[:, :2] ** 2, axis=1))))
        rms_dead = rms_to_parametric()
        dg.init_from_means(xp); torch.cuda.synchronize()
        trace = [rms_to_parametric()]
        it, t_np, conv_at = 0, 0.0, None
        while it < 200:
            torch.cuda.synchronize(); a = time.perf_counter()
            for s in range(5):
                dg.conv_step(o3, 1000 + it + s); dg.product_step(o3, 1000 + it + s, "lcv", "gibbs")
            torch.cuda.synchronize(); t_np += time.perf_counter() - a
            it += 5
            trace.append(rms_to_parametric())
            if it >= 10 and abs(trace[-1] - trace[-2]) < 1e-3:
                conv_at = it
                break
        out["solve"] = {"what": "Manhattan-3500, N=100: parametric solve (batched Jacobian kernel + sparse LM on the host) -> non-parametric iterations "
                                "(conv sweep + manikde! bandwidths + multiscale Gibbs product; whole-graph Jacobi schedule, no Bayes tree) until the RMS "
                                "distance of the pose means to the parametric solution changes by < 1e-3 m over 5 iterations",
                        "parametric_solve_s": t_par, "nonparametric_iterations": conv_at, "nonparametric_s": t_np, "wall_clock_s": t_par + t_np,
                        "rms_translation_to_parametric_m": trace[-1], "rms_trace_every_5_iterations": trace,
                        "converged": conv_at is not None}
        # the same loop started from the dead-reckoned beliefs: a Jacobi schedule moves information one factor per iteration, so
        # the 3500-pose chain does NOT converge in any practical number of iterations (the reference's solveTree! eliminates on
        # the Bayes tree instead) -- stated, not hidden
        dg.bel[R.Pose2].copy_(saved); torch.cuda.synchronize()
        a = time.perf_counter()
        for s in range(100):
            dg.conv_step(o3, 3000 + s); dg.product_step(o3, 3000 + s, "lcv", "gibbs")
        torch.cuda.synchronize()
        out["solve"]["from_dead_reckoning"] = {"rms_to_parametric_m_start": rms_dead, "rms_to_parametric_m_after_100_iterations": rms_to_parametric(),
                                               "seconds_100_iterations": time.perf_counter() - a}
        dg.bel[R.Pose2].copy_(saved)
        # ---- the ORDERED schedule, from NO beliefs at all (IIF initAll! order, then Gauss-Seidel over colour classes; device-resident
        # rome_store + one rome_upsolve_plan per independent group, rome_jl_amd.schedule.OrderedSolve): iterations and seconds to the same
        # criterion, next to the Jacobi figures above
        try:
            from rome_jl_amd.clique import DeviceStore
            from rome_jl_amd.schedule import OrderedSolve
            fg0 = R.loadG2o(args.g2o, N=N) if (args.g2o and args.g2o != "synthetic") else R.synth_manhattan(P=args.poses, loops=args.loops, seed=0x524F4D45 + rank, N=N)
            a = time.perf_counter()
            store = DeviceStore(fg0, ctx=ctx, upload=False)
            osv = OrderedSolve(store, kind="colour")
            t_plan = time.perf_counter() - a
            ls0 = list(fg0.variables)
            mp0 = np.array([xp[l] for l in ls0])
            import ctypes

            def rms_store():
                bel = np.zeros((len(ls0), 3, N))
                R._lib.check(R._lib.load().rome_store_download(store.handle, 0, 0, 0, len(ls0), bel.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), ctx.handle)
                m, _ = R.belief_stats(bel)
                return float(np.sqrt(np.mean(np.sum((m[:, :2] - mp0[:, :2]) ** 2, axis=1))))
            ctx.synchronize(); a = time.perf_counter()
            osv.init(R.make_opts(N=N, seed=1)); ctx.synchronize()
            t_init = time.perf_counter() - a
            tr = [rms_store()]
            t_sw, it2, conv2 = 0.0, 0, None
            while it2 < 100:
                ctx.synchronize(); a = time.perf_counter()
                osv.sweep(R.make_opts(N=N, seed=100 + it2), 5); ctx.synchronize()
                t_sw += time.perf_counter() - a
                it2 += 5
                tr.append(rms_store())
                if abs(tr[-1] - tr[-2]) < 1e-3:
                    conv2 = it2
                    break
            out["solve"]["from_init_all_ordered"] = {
                "what": "NO starting beliefs: IIF initAll!-order init pass (%d rounds, %d independent groups), then Gauss-Seidel sweeps over %d colour classes; "
                        "device-resident plans; criterion as above" % (len(osv.levels), len(osv.init_plans), len(osv.sweep_plans)),
                "plan_build_s": t_plan, "init_pass_s": t_init, "rms_to_parametric_m_after_init": tr[0], "sweeps": conv2, "sweeps_s": t_sw,
                "wall_clock_s": t_plan + t_init + t_sw, "rms_to_parametric_m": tr[-1], "rms_trace_every_5_sweeps": tr, "converged_by_criterion": conv2 is not None,
                "note": "the criterion is met because the sweeps STALL, not because they reach the parametric solution: a sweep multiplies neighbours' "
                        "BELIEFS (not cavity messages), the belief spread collapses to the measurement-noise level within ~3 sweeps and the large-loop error "
                        "left by the init pass is frozen in (DESIGN.md section 11); the reference's own solveTree! result on Manhattan-500 sits metres from the MAP too"}
        except Exception as e:   # noqa: BLE001
            out["solve"]["from_init_all_ordered"] = {"error": repr(e)}

        # ---- the TREE solve (solveTree!-shaped), from the FACTORS ALONE: no parametric start, no dead reckoning, no init pass.
        # Headline form: variable elimination in relative-factor algebra (rome_jl_amd.elimination; `R.solveTree(fg, messages="elimination")`):
        # sampled relative edges, compositions for the pair marginals of an eliminated pose's neighbours, the reference's product for
        # parallel edges, back substitution from anchor blocks.  Context: the clique forms of rome_jl_amd.tree (round 5) from initAll!.
        def load_fg():
            return R.loadG2o(args.g2o, N=N) if (args.g2o and args.g2o != "synthetic") else R.synth_manhattan(P=args.poses, loops=args.loops, seed=0x524F4D45 + rank, N=N)

        def rms_of(fgx, lsx, mpx, aligned=False):
            m, _ = R.belief_stats(np.stack([fgx.getVal(l) for l in lsx]))
            A, B = m[:, :2], mpx[:, :2]
            if aligned:       # after the best rigid transform (what remains is not a gauge error)
                A, B = A - A.mean(0), B - B.mean(0)
                U_, _, Vt = np.linalg.svd(A.T @ B); Rr = (U_ @ Vt).T
                if np.linalg.det(Rr) < 0:
                    Rr = (U_ @ np.diag([1.0, -1.0]) @ Vt).T
                A = A @ Rr.T
            return float(np.sqrt(np.mean(np.sum((A - B) ** 2, axis=1))))
        try:
            from rome_jl_amd.elimination import RelativeEliminationSolver
            fg1 = load_fg()
            ls1 = list(fg1.variables)
            mp1 = np.array([xp[l] for l in ls1])
            ctx.synchronize(); a = time.perf_counter()
            esv = RelativeEliminationSolver(fg1, ctx=ctx, structures=args.tree_structures)
            ctx.synchronize(); t_build = time.perf_counter() - a
            passes = []
            for ps in range(8):      # the pooled sequence: what solveTree(..., passes=8) leaves after each pass
                o4 = R.make_opts(N=N, seed=100 + ps)
                ctx.synchronize(); a = time.perf_counter(); esv.solve(o4); ctx.synchronize(); tp = time.perf_counter() - a
                esv.download(fg1)
                passes.append({"pass_s": tp, "rms_to_parametric_m": rms_of(fg1, ls1, mp1), "rms_after_rigid_alignment_m": rms_of(fg1, ls1, mp1, True)})
            single = []
            for ps in range(8):      # independent single passes (what ONE pass gives, seed by seed)
                esv.reset(); esv.solve(R.make_opts(N=N, seed=300 + ps)); esv.download(fg1)
                single.append(rms_of(fg1, ls1, mp1))
            st_ = esv.stats()
            rr = [p_["rms_to_parametric_m"] for p_ in passes]
            out["solve"]["from_tree"] = {
                "what": "Manhattan-3500 from the factors alone (NO starting beliefs, no init pass): variable elimination in relative-factor algebra -- %d rounds of "
                        "independent lowest-loss variables (%d merges of parallel edges by the reference's product, %d compositions, %d eliminations kept as a star "
                        "about the tightest neighbour), back substitution from anchor blocks; %d launch steps per pass, device-resident; passes pool their particles"
                        % (st_["rounds"], st_["merges"], st_["compositions"], st_["approximated_eliminations"], st_["launch_steps"]),
                "reference": "rms to the MAP = solveGraphParametric incl. its undamped polish steps (a damped LM run alone stops ~0.8 m RMS from the MAP on this graph)",
                "structure_and_plans_build_s": t_build, "passes": passes, "wall_clock_s_first_pass": t_build + passes[0]["pass_s"],
                "seconds_per_pass": float(np.median([p_["pass_s"] for p_ in passes])), "elimination": st_,
                "rms_to_parametric_m_over_passes": {"min": min(rr), "median": float(np.median(rr)), "max": max(rr)},
                "rms_after_rigid_alignment_m_over_passes": {"min": min(p_["rms_after_rigid_alignment_m"] for p_ in passes),
                                                            "median": float(np.median([p_["rms_after_rigid_alignment_m"] for p_ in passes])),
                                                            "max": max(p_["rms_after_rigid_alignment_m"] for p_ in passes)},
                "independent_single_passes_rms_to_parametric_m": single,
                "note": "N = 100 particles.  The error of a pass is the bias of the star approximations (deterministic given the elimination structure: 0.5 m in the "
                        "Gaussian restatement of this schedule) plus sampling noise; pooling passes averages the noise, `structures` > 1 the bias as well"}
            del esv
        except Exception as e:   # noqa: BLE001
            out["solve"]["from_tree"] = {"error": repr(e)}
        try:
            from rome_jl_amd.tree import TreeSolver
            fg1 = load_fg()
            a = time.perf_counter(); R.initAllOrdered(fg1, seed=1, ctx=ctx); t_ini = time.perf_counter() - a
            ls1 = list(fg1.variables)
            mp1 = np.array([xp[l] for l in ls1])
            r_init, r_init_al = rms_of(fg1, ls1, mp1), rms_of(fg1, ls1, mp1, True)
            a = time.perf_counter(); tsv = TreeSolver(fg1, messages=args.tree_messages, ctx=ctx); t_build = time.perf_counter() - a
            tsv.upload()
            passes = []
            for ps in range(4):
                o4 = R.make_opts(N=N, seed=100 + ps)
                ctx.synchronize(); a = time.perf_counter(); tsv.up(o4); ctx.synchronize(); tu = time.perf_counter() - a
                a = time.perf_counter(); tsv.down(o4); ctx.synchronize(); td = time.perf_counter() - a
                tsv.download()
                passes.append({"up_s": tu, "down_s": td, "rms_to_parametric_m": rms_of(fg1, ls1, mp1), "rms_after_rigid_alignment_m": rms_of(fg1, ls1, mp1, True)})
            st_ = tsv.stats()
            # IIF's own message form (per-variable separator beliefs, gibbsIters = 3 up / 1 down) on the same tree, from the same init
            fg2 = load_fg()
            R.initAllOrdered(fg2, seed=1, ctx=ctx)
            tsm = TreeSolver(fg2, tree=tsv.tree, messages="marginal", ctx=ctx); tsm.upload()
            o5 = R.make_opts(N=N, seed=100)
            ctx.synchronize(); a = time.perf_counter(); tsm.solve(o5); ctx.synchronize(); t_marg = time.perf_counter() - a
            tsm.download()
            marg = {"seconds_per_pass": t_marg, "rms_to_parametric_m": rms_of(fg2, ls1, mp1), "rms_after_rigid_alignment_m": rms_of(fg2, ls1, mp1, True),
                    "what": "IIF's message form (TreeBelief per separator variable; upGibbsCliqueDensity with 3 iterations, down pass 1): on a pose graph with ONE prior "
                            "the cliques below the prior's clique hold no absolute information -- the solve stays where the init pass left it"}
            out["solve"]["from_tree_clique_forms"] = {
                "what": "context (round 5's solver): initAll!-order init pass, then the Bayes tree (minimum-degree elimination -> %d cliques -> %d levels, widest %d) up + down "
                        "pass with '%s' messages, one rome_upsolve_plan per level; one-shot outward clique solves and a belief-weighted down pass: no better than its init pass"
                        % (st_["cliques"], st_["levels"], st_["width_max"], args.tree_messages),
                "init_all_s": t_ini, "rms_to_parametric_m_after_init": r_init, "rms_after_rigid_alignment_m_after_init": r_init_al,
                "tree_and_plans_build_s": t_build, "passes": passes,
                "wall_clock_s_first_pass": t_ini + t_build + passes[0]["up_s"] + passes[0]["down_s"], "tree": st_,
                "seconds_per_pass": float(np.median([p_["up_s"] + p_["down_s"] for p_ in passes])), "marginal_messages": marg,
                "frontier_width_by_level": [len(l) for l in tsv.tree.levels]}
        except Exception as e:   # noqa: BLE001
            out["solve"]["from_tree_clique_forms"] = {"error": repr(e)}

        # ---- BASELINE configs[4] ("synthetic SE(3) helix, 10k Pose3 + Pose3Pose3, parametric Gauss-Newton batched Jacobians"): where the time
        # of the parametric solve goes.  The GPU part is the batched residual / Jacobian kernels (k_lin<...>, tables device-resident across
        # iterations); the sparse normal equations are solved on the HOST (scipy) -- stated, so nobody reads the wall-clock as GPU work
        try:
            from rome_jl_amd.distributed import LinearizeShard
            fgh = R.synth_helix3d(P=10000, N=8)
            R.dead_reckon_init_pose3(fgh, seed=7)
            lsh = LinearizeShard(torch, None, 1, 0, device=dev)
            stt = {}
            a = time.perf_counter(); R.solveGraphParametric(fgh, max_iters=40, ctx=ctx, shard=lsh, stats=stt); t_h = time.perf_counter() - a
            sh_ = stt.get("shard", {})
            out["parametric_helix10k"] = {
                "what": "10 000 Pose3, %d Pose3Pose3 + 1 PriorPose3, Levenberg-Marquardt: every iteration = one k_lin launch per factor kind (rows of this rank) + "
                        "host sparse solve of the 60 000-unknown normal equations" % (len(fgh.factors) - 1),
                "wall_clock_s": t_h, "iterations": stt["iterations"], "linearizations": stt["linearizations"], "setup_s": stt["setup_s"],
                "linearize_s": stt["linearize_s"], "host_solve_s": stt["host_solve_s"],
                "linearize_split_ms": {k: sh_.get(k) for k in ("upload_ms", "kernel_ms", "exchange_ms", "download_ms")},
                "host_fraction": stt["host_solve_s"] / max(t_h, 1e-9),
                "note": "linearize_s includes the host-side gather X[ia], the upload of the gathered coordinates, the kernels, the download and the sparse assembly "
                        "(values permuted into the CSR structure computed once per problem); kernel_ms is the synchronised launch time of the k_lin kernels alone "
                        "(profiles/r06_linearize_trace.md has their rocprof rows)"}
        except Exception as e:   # noqa: BLE001
            out["parametric_helix10k"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(R, pk, fg, N, args.cpu_seconds)
        # the other half of the metric next to a CPU number: one Bayes-tree pass (up + down, relative messages) on a bounded SAMPLE -- the
        # first 200 poses of the same g2o file with their loop closures -- by the oracle's restatement on the host (oracle/cpu_tree_bench.py,
        # fresh process) and by the device path on the SAME sub-graph
        if args.g2o and args.g2o != "synthetic" and not args.no_modes:
            try:
                import subprocess
                p_ = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_tree_bench.py"), args.g2o, "--poses", "200", "--particles", str(N)],
                                    capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))   # (row-by-row calls: one thread)
                cpu_t = json.loads(p_.stdout.strip().splitlines()[-1]) if p_.returncode == 0 else {"error": p_.stderr[-400:]}
                from rome_jl_amd.tree import TreeSolver
                fgs = R.initfg(N)
                rows_ = [ln.split() for ln in open(args.g2o) if ln.startswith("EDGE_SE2")]
                rows_ = [t for t in rows_ if int(t[1]) < 200 and int(t[2]) < 200]
                for k in sorted({int(x) for t in rows_ for x in t[1:3]}):
                    fgs.addVariable("x%d" % k, R.Pose2)
                fgs.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), np.diag([0.01, 0.01, 0.0025]))))
                for t in rows_:
                    u = [float(x) for x in t[6:12]]
                    Cc = np.linalg.inv(np.array([[u[0], u[1], u[2]], [u[1], u[3], u[4]], [u[2], u[4], u[5]]]))
                    fgs.addFactor(["x%s" % t[1], "x%s" % t[2]], R.Pose2Pose2(R.MvNormal(np.array([float(x) for x in t[3:6]]), 0.5 * (Cc + Cc.T))))
                R.dead_reckon_init(fgs, seed=1)
                tss = TreeSolver(fgs, messages="relative", ctx=ctx); tss.upload()
                tss.solve(R.make_opts(N=N, seed=3)); ctx.synchronize()
                a = time.perf_counter()
                for ps in range(5):
                    tss.solve(R.make_opts(N=N, seed=4 + ps))
                ctx.synchronize()
                g_t = (time.perf_counter() - a) / 5
                out["cpu_baseline"]["tree_pass_sample"] = {
                    "sample": "Bayes-tree pass (up + down, relative messages) of the first 200 poses of the same file with their loop closures (%s factors, N = %d)" % (cpu_t.get("factors"), N),
                    "cpu_port": cpu_t, "gpu_seconds_per_pass": g_t, "ratio": (cpu_t["seconds_per_pass"] / g_t) if "seconds_per_pass" in cpu_t else None,
                    "note": "the CPU side is the oracle's restatement driven row by row from Python (test infrastructure), not the reference's Julia: a scale, not a race; "
                            "a 184-clique tree is 34 narrow levels, i.e. the GPU side is launch latency"}
            except Exception as e:   # noqa: BLE001
                out["cpu_baseline"]["tree_pass_sample"] = {"error": repr(e)}
        gpu = dict(out.get("gpu_convolutions_per_s_by_solver", {}))
        gpu[args.solver] = value
        # GPU/CPU ratios pair the SAME algorithm on both sides (the CPU side at its best thread count):
        #   closed_form, newton        <-> the oracle's closed form: device NEWTON returns the analytic root in one pass (no start point, no
        #                                  inflation cycle) -- the oracle's iterative Newton mode is NOT its counterpart
        #   gauss_newton               <-> the oracle's Newton mode (= Gauss-Newton iteration on the residual functor, inflate_cycles x)
        #   nelder_mead                <-> the oracle's Nelder-Mead (the reference's algorithm)
        cpu_of = {"closed_form": "closed_form", "newton": "closed_form", "newton_with_status": "closed_form", "gauss_newton": "newton",
                  "nelder_mead": "nelder_mead"}
        by = out["cpu_baseline"].get("by_solver") or {}
        out["cpu_baseline"]["gpu_over_cpu_same_solver"] = {k: gpu[k] / by[cpu_of[k]]["conv_per_s"] for k in gpu if cpu_of.get(k) in by}
        out["cpu_baseline"]["pairing"] = {k: "cpu " + v for k, v in cpu_of.items()}
        if "newton" in by and "newton" in gpu:   # NOT a same-algorithm pair: kept under its own name
            out["cpu_baseline"]["gpu_analytic_newton_over_cpu_iterative_newton"] = gpu["newton"] / by["newton"]["conv_per_s"]
        if cpu_of[args.solver] in by:   # the pair to read next to `value`: same algorithm on both sides
            out["cpu_baseline"]["same_solver"] = cpu_of[args.solver]
            out["cpu_baseline"]["same_solver_value"] = by[cpu_of[args.solver]]["conv_per_s"]
            out["cpu_baseline"]["same_solver_cores"] = by[cpu_of[args.solver]].get("threads")
        if "gauss_newton" in gpu and "newton" in by:   # the functor-iterating root-find, the same algorithm on both sides
            out["cpu_baseline"]["functor_iteration_pair"] = {"gpu_gauss_newton": gpu["gauss_newton"], "cpu_newton": by["newton"]["conv_per_s"],
                                                             "ratio": gpu["gauss_newton"] / by["newton"]["conv_per_s"], "cores": by["newton"].get("threads")}

    # the JSON line must be the LAST thing on stdout: RCCL's version banner sits in the C stdio buffer of the ranks that
    # initialised a communicator and would otherwise be flushed at exit, after the line
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if multi:
        dist.barrier()
        torch.cuda.synchronize()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        pipe.close()
        dist.destroy_process_group()


def cpu_baseline(R, pk, fg, N, budget_s):
    """The oracle (C port of the reference algorithm) on the host cores of this box, on a bounded prefix of the SAME
    convolution table: oracle/cpu_bench.py in fresh processes (OpenMP configured before start, no torch thread pool beside it),
    built there with -O3 -march=native, >= 5 repetitions, median; Nelder-Mead (the reference's Optim algorithm: `value`),
    Newton and closed form, at 1 thread and on all cores.  Reported baseline only -- never part of the product path."""
    import subprocess
    import tempfile
    factor, dr, fixed, target = R.PackedGraph.conv_table(pk.p2p2)
    L = R.cholesky_lower(pk.p2p2["cov"])
    bel = pk.beliefs(fg, R.Pose2)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "table.npz")
        np.savez(path, mu=pk.p2p2["mu"], L=L, bel=bel, factor=factor, dir=dr, fixed=fixed, target=target)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), path, "--budget", str(budget_s)],
                           capture_output=True, text=True)
    if p.returncode != 0:
        return {"value": None, "unit": "convolutions/s", "kind": "port", "error": p.stderr[-800:]}
    r = json.loads(p.stdout.strip().splitlines()[-1])
    nm = r["best"]["nelder_mead"]
    # the reference itself: bench/ref_conv.jl times IIF's approxConvBelief iff a Julia with RoME/IIF exists here
    import shutil
    julia = shutil.which("julia")
    if julia:
        try:
            jp = subprocess.run([julia, os.path.join(ROOT, "bench", "ref_conv.jl")], capture_output=True, text=True, timeout=900,
                                env=dict(os.environ, JULIA_NUM_THREADS=str(nm["threads"])))
            jref = json.loads(jp.stdout.strip().splitlines()[-1]) if jp.returncode == 0 else {"error": jp.stderr[-400:]}
        except Exception as e:   # noqa: BLE001
            jref = {"error": repr(e)}
    else:
        jref = "julia reference not runnable on this box (no julia on PATH; bench/ref_conv.jl is the provided script)"
    return {"value": nm["conv_per_s"], "unit": "convolutions/s", "cores": nm["threads"], "kind": "port",
            "algorithm": "Optim.jl-default Nelder-Mead per particle (reference algorithm), inflate_cycles=3",
            "sample": "first %d of %d (factor,direction) convolutions of the same graph, N=%d, median of %d runs of %.2f s"
                      % (nm["sample_convolutions"], len(factor), N, nm["reps"], nm["median_s"]),
            "flags": r["flags"], "omp": r["omp"], "host": r["host"],
            "one_thread_conv_per_s": nm["one_thread_conv_per_s"], "parallel_efficiency": nm["parallel_efficiency"],
            "by_solver": r["best"], "runs": r["runs"], "julia_reference": jref}


if __name__ == "__main__":
    main()
