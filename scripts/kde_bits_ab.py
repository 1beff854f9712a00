#!/usr/bin/env python3
"""Bitwise A/B of rome_kde_bandwidth_dev between two builds (scripts/ubench/lib_kde_old.so = scripts/build_variant.sh of an older
rome_kde.hip, and the in-tree library): the proposals of four Manhattan solve iterations + beliefs with headings uniform on the circle and
straddling +-pi.  Round 4: the kernel before the wrap-free body / the golden section without derivative sums / the merged logarithms
(commit 2a0970f) against the final one -- 131 076 bandwidths, bit-identical."""
import os, sys, subprocess, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = r'''
import os, sys
sys.path.insert(0, os.environ["ROOT"])
import numpy as np, torch, rome_jl_amd as R
from rome_jl_amd import _lib
fg = R.loadG2o(os.path.join(os.environ["ROOT"], "tests/golden/manhattan.g2o"), N=100); R.dead_reckon_init(fg, seed=1)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=100, solver=1, seed=11)
outs = []
lib = _lib.load()
for s in range(4):
    dg.conv_step(o, s)
    prop = dg.prop[R.Pose2]; rows = dg.n_prop[R.Pose2]
    bw = torch.empty((prop.shape[0], 3), dtype=torch.float64, device="cuda")
    _lib.check(lib.rome_kde_bandwidth_dev(dg.ctx.handle, 3, rows, 100, prop.data_ptr(), 0b100, 0.0, 0.0, bw.data_ptr()), dg.ctx.handle)
    dg.ctx.synchronize(); torch.cuda.synchronize()
    outs.append(bw[:rows].cpu().numpy().copy())
    dg.product_step(o, s, "lcv", "gibbs")
# also wide headings: uniform on the circle (the wrap path) and a belief straddling +-pi
rng = np.random.default_rng(0)
bel = rng.normal(size=(64, 3, 100)); bel[:32, 2] = rng.uniform(-np.pi, np.pi, size=(32, 100)); bel[32:, 2] = ((np.pi + 0.2 * rng.normal(size=(32, 100))) + np.pi) % (2 * np.pi) - np.pi
from rome_jl_amd import api
outs.append(api.kde_bandwidth(bel))
np.save(os.environ["OUT"], np.concatenate([o.ravel() for o in outs]))
'''
res = {}
for name, lib in (("old", os.path.join(root, "scripts/ubench/lib_kde_old.so")), ("new", "")):
    env = dict(os.environ, ROOT=root, OUT="/tmp/kde_%s.npy" % name)
    if lib: env["ROME_MI355_LIB"] = lib
    else: env.pop("ROME_MI355_LIB", None)
    subprocess.run([sys.executable, "-c", code], env=env, check=True, stderr=subprocess.DEVNULL)
    res[name] = np.load(env["OUT"])
a, b = res["old"], res["new"]
print("bandwidths compared:", a.size, " bit-identical:", bool(np.array_equal(a, b)), " max |rel diff|: %.3g" % float(np.max(np.abs(a - b) / np.abs(a))))
