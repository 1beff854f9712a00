"""device vs oracle restatement of one elimination pass on a Manhattan prefix, seed by seed: fraction of particles within 1e-6 and the worst
|difference of a pose mean| (a single categorical draw that flips -- the device's reciprocals are an ulp from the oracle's divisions --
moves one particle, and every product downstream of it sees other bandwidths).
    python scripts/elim_parity_seeds.py [poses=1000] [seeds=61,62,...]      ROME_MI355_LIB / ROME_ORACLE_SO select the pair"""
import os, sys, tempfile, pathlib
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_elimination as T
poses = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seeds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [61, 62, 63, 64, 65, 66]
tmp = pathlib.Path(tempfile.mkdtemp())
for sd in seeds:
    fg = T.manhattan_subgraph(poses, 100, tmp)
    dev, worst = T._both(fg, sd)
    print("seed %d: %.4f of the particles within 1e-6, worst |mean difference| %.3e" % (sd, worst[0][0], worst[0][1]), flush=True)
