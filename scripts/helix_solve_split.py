"""the parametric solve of the 10k-pose SE(3) helix (BASELINE configs[4]) with the time split bench.py reports as parametric_helix10k"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import rome_jl_amd as R
from rome_jl_amd.distributed import LinearizeShard
dev = torch.device("cuda", 0)
ctx = R.default_context()
for rep in range(2):
    fgh = R.synth_helix3d(P=10000, N=8)
    R.dead_reckon_init_pose3(fgh, seed=7)
    lsh = LinearizeShard(torch, None, 1, 0, device=dev)
    stt = {}
    a = time.perf_counter(); R.solveGraphParametric(fgh, max_iters=40, ctx=ctx, shard=lsh, stats=stt); t_h = time.perf_counter() - a
    print("run %d: wall %.3f s  setup %.3f  linearize %.3f (%d linearisations)  host solve %.3f  shard split ms %s" %
          (rep, t_h, stt["setup_s"], stt["linearize_s"], stt["linearizations"], stt["host_solve_s"], {k: round(v, 2) for k, v in stt.get("shard", {}).items()}))
