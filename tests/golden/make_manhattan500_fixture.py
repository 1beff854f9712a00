#!/usr/bin/env python3
"""Builds tests/golden/manhattan500_reference_solve.npz from the reference's data artefact
examples/fg-after-solve.tar.gz (byte-identical to examples/manhattan-batch-500-fg.tar.gz; SURVEY.md §8(c) row
"Artefact"): the DistributedFactorGraphs save of the first 500 Manhattan edges AFTER a reference `solveTree!`
(Feb-2020 package versions).  It is the only output of the reference's own non-parametric solver that ships with
the repository, so it pins the convolution / solve statistically (tests/test_gpu_reference_solve.py).

Runs only where /root/reference exists (this container); the .npz is data: factor list (variable indices, μ, Σ),
prior, and per variable the N=100 posterior particles (x, y, θ), KDE bandwidths and the three PPE estimates.
Particles are stored as float32 (used statistically only), everything else float64.

    python tests/golden/make_manhattan500_fixture.py [/root/reference/examples/fg-after-solve.tar.gz]
"""
import io
import json
import os
import re
import sys
import tarfile

import numpy as np

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/examples/fg-after-solve.tar.gz"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "manhattan500_reference_solve.npz")

_NUM = r"[-+0-9.eE]+"


def parse_fullnormal(s):
    mu = np.array([float(x) for x in re.search(r"μ: \[([^\]]*)\]", s).group(1).split(",")])
    rows = re.search(r"Σ: \[([^\]]*)\]", s).group(1).split(";")
    cov = np.array([[float(x) for x in r.split()] for r in rows])
    assert mu.shape == (3,) and cov.shape == (3, 3)
    return mu, cov


variables, factors = {}, {}
with tarfile.open(SRC, "r:gz") as tf:
    for m in tf.getmembers():
        if not m.isfile() or not m.name.endswith(".json"):
            continue
        obj = json.load(io.TextIOWrapper(tf.extractfile(m), encoding="utf-8"))
        (variables if "/variables/" in m.name else factors)[obj["label"]] = obj

labels = sorted(variables, key=lambda s: int(s[1:]))
assert labels == ["x%d" % i for i in range(len(labels))]
V = len(labels)
pts = np.zeros((V, 100, 3), np.float32)
bw = np.zeros((V, 3))
ppe = np.zeros((V, 3, 3))  # [:, 0] suggested, [:, 1] max, [:, 2] mean
for i, lb in enumerate(labels):
    sd = json.loads(variables[lb]["solverDataDict"])["default"]
    vv = np.asarray(sd["vecval"], dtype=np.float64)
    assert vv.size == 300 and int(sd["dimval"]) == 3
    pts[i] = vv.reshape(100, 3)  # column-major 3 x N == N rows of (x, y, θ)
    b = sd["vecbw"]
    bw[i] = json.loads(b) if isinstance(b, str) else b
    pp = json.loads(variables[lb]["ppeDict"])["default"]
    ppe[i, 0], ppe[i, 1], ppe[i, 2] = pp["suggested"], pp["max"], pp["mean"]

edges, mus, covs, prior = [], [], [], None
for lb, f in factors.items():
    d = json.loads(f["data"])
    fnc = d["fnc"]
    mu, cov = parse_fullnormal(fnc.get("datastr", fnc.get("str")))
    ids = [int(s[1:]) for s in d["fncargvID"]]
    if f["fnctype"] == "PriorPose2":
        assert ids == [0]
        prior = (mu, cov)
    else:
        assert f["fnctype"] == "Pose2Pose2" and len(ids) == 2
        edges.append(ids); mus.append(mu); covs.append(cov)
order = np.lexsort((np.array(edges)[:, 1], np.array(edges)[:, 0]))
edges = np.array(edges, np.int32)[order]
mus = np.array(mus)[order]
covs = np.array(covs)[order]
assert edges.shape == (500, 2) and prior is not None
np.savez_compressed(OUT, edges=edges, mu=mus, cov=covs, prior_mu=prior[0], prior_cov=prior[1],
                    particles=pts, bandwidth=bw, ppe=ppe)
print("wrote", OUT, os.path.getsize(OUT), "bytes;", V, "variables,", len(edges), "Pose2Pose2 + 1 PriorPose2")
