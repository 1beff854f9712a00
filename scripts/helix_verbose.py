import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, rome_jl_amd as R
fgh = R.synth_helix3d(P=10000, N=8); R.dead_reckon_init_pose3(fgh, seed=7)
R.solveGraphParametric(fgh, max_iters=40, verbose=True)
