"""saveDFG / loadDFG: the file format either side of the path (host-side data conversion only, no compute).

The reference stores factor graphs as a `.tar.gz` of one JSON document per variable and per factor
(DistributedFactorGraphs `saveDFG` / `loadDFG!`; used by test/testG2oExportSE3.jl:21 on test/testdata/g2otest.tar.gz and by the
example artefact examples/fg-after-solve.tar.gz).  Two generations of that format exist in the reference's own data files and both
are read here:

  * DFG 0.25 (`"_version": "0.25.1"`, test/testdata/g2otest.tar.gz): variable JSON with `variableType` ("RoME.Pose3"), `solverData`
    (a list, one entry per solveKey: `vecval` = dim x N coordinates column-major, `vecbw`, `dimval`, ...), `ppes`; factor JSON with
    `fnctype`, `_variableOrderSymbols` and `data` (a JSON string: `fnc` = the Packed<F> struct of src/factors/*.jl -- `Z` /
    `bearstr`, `rangstr` as `IncrementalInference.PackedFullNormal` / `PackedNormal` / ... --, `multihypo`, `nullhypo`, `inflation`).
  * the early-2020 layout of examples/fg-after-solve.tar.gz: `solverDataDict` / `ppeDict` JSON strings keyed by solveKey, `softtype`
    text, and the measurement as the text `FullNormal(dim: 3 μ: [...] Σ: [...; ...])`.

`saveDFG` writes the 0.25 layout (what a current reference `loadDFG!` reads), `loadDFG` returns this package's FactorGraph with the
beliefs (`vals`), bandwidths (`bws`) and point estimates (`ppes`) the file carries.  The Packed<F> <-> factor conversion is the dict
form of the reference's `convert(::Type{Packed<F>}, ::F)` pairs (Pose2D.jl:76-84, PriorPose2.jl:55-63, BearingRange2D.jl:76-88,
Pose3Pose3.jl:46-54, Pose3D.jl:28-36, Point2D.jl:49-77).
"""
import base64
import io
import json
import re
import tarfile
import time

import numpy as np

from .factors import (MvNormal, Normal, Uniform, Pose2, Point2, Pose3, Pose2Pose2, PriorPose2, Pose2Point2BearingRange, Pose3Pose3,
                      PriorPose3, PriorPoint2, Point2Point2)
from .graph import FactorGraph

_VERSION = "0.25.1"
_VARTYPES = {"Pose2": Pose2, "Point2": Point2, "Pose3": Pose3}
_FACTORS = {c.__name__: c for c in (Pose2Pose2, PriorPose2, Pose2Point2BearingRange, Pose3Pose3, PriorPose3, PriorPoint2, Point2Point2)}


# ---------------------------------------------------------------- Packed beliefs (IIF PackedSamplableBelief JSON forms)
def packBelief(b):
    if isinstance(b, Normal):
        return {"_type": "IncrementalInference.PackedNormal", "mu": float(b.mu), "sigma": float(b.sigma)}
    if isinstance(b, Uniform):
        return {"_type": "IncrementalInference.PackedUniform", "a": float(b.a), "b": float(b.b), "PackedSamplableTypeJSON": "IncrementalInference.PackedUniform"}
    if isinstance(b, MvNormal):
        return {"_type": "IncrementalInference.PackedFullNormal", "mu": [float(x) for x in b.mu],
                "cov": [float(x) for x in np.asarray(b.cov).flatten(order="F")]}     # Julia `cov[:]`: column-major
    raise TypeError("cannot pack %r" % (b,))


def unpackBelief(d):
    t = d["_type"].split(".")[-1]
    if t == "PackedNormal":
        return Normal(d["mu"], d["sigma"])
    if t == "PackedUniform":
        return Uniform(d["a"], d["b"])
    if t in ("PackedFullNormal", "PackedZeroMeanFullNormal"):
        cov = np.asarray(d["cov"], dtype=np.float64)
        n = int(round(np.sqrt(cov.size)))
        mu = np.zeros(n) if t == "PackedZeroMeanFullNormal" else np.asarray(d["mu"], dtype=np.float64)
        return MvNormal(mu, cov.reshape(n, n, order="F"))
    if t in ("PackedDiagNormal", "PackedZeroMeanDiagNormal"):
        diag = np.asarray(d["diag"], dtype=np.float64)
        mu = np.zeros(diag.size) if t == "PackedZeroMeanDiagNormal" else np.asarray(d["mu"], dtype=np.float64)
        return MvNormal(mu, np.diag(diag))
    raise ValueError("unsupported packed belief %s" % d["_type"])


def packFactor(f):
    """factor -> the field dict of its Packed<F> struct."""
    if isinstance(f, Pose2Point2BearingRange):
        return {"bearstr": packBelief(f.bearing), "rangstr": packBelief(f.range)}
    return {"Z": packBelief(f.Z)}


def unpackFactor(fnctype, fnc):
    name = fnctype.split(".")[-1]
    if name.startswith("Packed"):
        name = name[len("Packed"):]
    if name not in _FACTORS:
        raise ValueError("factor type %s is not on the supported path" % fnctype)
    if name == "Pose2Point2BearingRange":
        return Pose2Point2BearingRange(unpackBelief(fnc["bearstr"]), unpackBelief(fnc["rangstr"]))
    return _FACTORS[name](unpackBelief(fnc["Z"]))


# ---------------------------------------------------------------- early-2020 text forms
def _parse_distribution_text(s):
    m = re.search(r"μ: \[([^\]]*)\]", s)
    if m and "Σ" in s:
        mu = np.array([float(x) for x in m.group(1).split(",")])
        rows = re.search(r"Σ: \[([^\]]*)\]", s).group(1).split(";")
        return MvNormal(mu, np.array([[float(x) for x in r.split()] for r in rows]))
    m = re.search(r"Normal\{Float64\}\(μ=([-+0-9.eE]+), σ=([-+0-9.eE]+)\)", s)
    if m:
        return Normal(float(m.group(1)), float(m.group(2)))
    raise ValueError("cannot parse distribution text %r" % s[:80])


def _legacy_factor(fnctype, fnc):
    if fnctype == "Pose2Point2BearingRange":
        return Pose2Point2BearingRange(_parse_distribution_text(fnc["bearstr"]), _parse_distribution_text(fnc["rangstr"]))
    return _FACTORS[fnctype](_parse_distribution_text(fnc.get("datastr", fnc.get("str"))))


# ---------------------------------------------------------------- load
def _label_key(s):
    m = re.match(r"^([A-Za-z_]+)(\d+)$", s)
    return (m.group(1), int(m.group(2))) if m else (s, -1)


def loadDFG(path, solveKey="default"):
    """-> FactorGraph with `.vals` (label -> (dim, N)), `.bws` (label -> bandwidths), `.ppes` (label -> {solveKey: {suggested, max, mean}}),
    `.solverParams` (dict, when the file carries one).  Variables are added in numeric label order, factors in file order."""
    variables, factors, meta = {}, [], None
    with tarfile.open(path, "r:*") as tf:
        for m in tf.getmembers():
            if not m.isfile() or not m.name.endswith(".json"):
                continue
            obj = json.load(io.TextIOWrapper(tf.extractfile(m), encoding="utf-8"))
            parts = m.name.split("/")
            if "variables" in parts:
                variables[obj["label"]] = obj
            elif "factors" in parts:
                factors.append(obj)
            elif parts[-1] == "dfg.json":
                meta = obj
    fg = FactorGraph()
    fg.bws, fg.ppes, fg.solverParams = {}, {}, (meta or {}).get("solverParams")
    if fg.solverParams and "N" in fg.solverParams:
        fg.N = int(fg.solverParams["N"])
    order = (meta or {}).get("addHistory") or sorted(variables, key=_label_key)
    order = [l for l in order if l in variables] + [l for l in sorted(variables, key=_label_key) if l not in set(order)]
    pending = {}
    for lb in order:
        v = variables[lb]
        if "solverData" in v or "variableType" in v:            # DFG 0.25
            vt = _VARTYPES.get(v["variableType"].split(".")[-1])
            sds = {sd["solveKey"]: sd for sd in v.get("solverData", [])}
            ppes = {p["solveKey"]: {k: np.asarray(p[k], dtype=np.float64) for k in ("suggested", "max", "mean") if k in p} for p in v.get("ppes", [])}
        else:                                                    # early-2020 layout
            sds = json.loads(v["solverDataDict"])
            st = next(iter(sds.values()))["softtype"]
            vt = _VARTYPES.get(re.match(r"^(?:RoME\.)?(\w+)", st).group(1))
            ppes = {k: {kk: np.asarray(p[kk], dtype=np.float64) for kk in ("suggested", "max", "mean") if kk in p}
                    for k, p in json.loads(v.get("ppeDict", "{}")).items()}
        if vt is None:
            raise ValueError("variable %s: type %s is not on the supported path" % (lb, v.get("variableType", "?")))
        fg.addVariable(lb, vt)
        fg.ppes[lb] = ppes
        sd = sds.get(solveKey)
        if sd is not None:
            vv = np.asarray(sd["vecval"], dtype=np.float64)
            d = int(sd["dimval"])
            bw = sd["vecbw"]
            fg.bws[lb] = np.asarray(json.loads(bw) if isinstance(bw, str) else bw, dtype=np.float64)[:d]
            if vv.size and vv.size % d == 0:
                pending[lb] = (np.ascontiguousarray(vv.reshape(-1, d).T), bool(sd.get("initialized", True)))
    ns = {a.shape[1] for a, _ in pending.values()}
    if len(ns) == 1:
        fg.N = ns.pop()
    for lb, (a, init) in pending.items():
        if a.shape[1] == fg.N and init:
            fg.initVariable(lb, a)
    for f in factors:
        d = json.loads(f["data"]) if isinstance(f["data"], str) else f["data"]
        labels = f.get("_variableOrderSymbols") or d.get("fncargvID")
        if isinstance(labels, str):
            labels = json.loads(labels)
        if "_version" in f:
            fac = unpackFactor(f["fnctype"], d["fnc"])
        else:
            fac = _legacy_factor(f["fnctype"], d["fnc"])
        mh = d.get("multihypo") or None
        fl = fg.addFactor(labels, fac, multihypo=mh)
        if d.get("nullhypo"):
            fg.nullhypo[fl] = float(d["nullhypo"])
    return fg


# ---------------------------------------------------------------- save
def _now():
    return time.strftime("%Y-%m-%dT%H:%M:%S.000+00:00", time.gmtime())


def _variable_doc(fg, lb, solveKey):
    vt = fg.variables[lb]
    d = vt.dim
    init = fg.isInitialized(lb)
    vals = fg.getVal(lb) if init else np.zeros((d, fg.N))
    bw = np.asarray(getattr(fg, "bws", {}).get(lb, np.zeros(d)), dtype=np.float64)
    sd = {"vecval": [float(x) for x in np.asarray(vals).T.reshape(-1)], "dimval": d, "vecbw": [float(x) for x in bw], "dimbw": d,
          "BayesNetOutVertIDs": [], "dimIDs": [], "dims": d, "eliminated": False, "BayesNetVertID": "NOTHING", "separator": [],
          "variableType": "RoME.%s" % vt.name, "initialized": bool(init), "infoPerCoord": [0.0] * d, "ismargin": False,
          "dontmargin": False, "solveInProgress": 0, "solvedCount": 0, "solveKey": solveKey, "covar": [], "_version": _VERSION}
    ppes = [dict({"solveKey": k, "_type": "DistributedFactorGraphs.MeanMaxPPE", "_version": _VERSION},
                 **{kk: [float(x) for x in vv] for kk, vv in p.items()}) for k, p in getattr(fg, "ppes", {}).get(lb, {}).items()]
    return {"label": lb, "tags": ["VARIABLE"], "timestamp": _now(), "nstime": "0", "ppes": ppes, "blobEntries": [],
            "variableType": "RoME.%s" % vt.name, "_version": _VERSION, "metadata": base64.b64encode(b"{}").decode(), "solvable": 1,
            "solverData": [sd]}


def _factor_doc(fg, flabel, labels, fac):
    mh = fg.multihypo.get(flabel)
    data = {"eliminated": False, "potentialused": False, "edgeIDs": [], "fnc": packFactor(fac),
            "multihypo": [1.0, mh[0], mh[1]] if mh else [], "certainhypo": list(range(1, len(labels) + 1)) if not mh else [1],
            "nullhypo": float(getattr(fg, "nullhypo", {}).get(flabel, 0.0)), "solveInProgress": 0, "inflation": 5.0}
    return {"label": flabel, "tags": ["FACTOR"], "_variableOrderSymbols": list(labels), "timestamp": _now(), "nstime": "0",
            "fnctype": type(fac).__name__, "solvable": 1, "data": json.dumps(data, separators=(",", ":")),
            "metadata": base64.b64encode(b"{}").decode(), "_version": _VERSION}


def saveDFG(fg, path, solveKey="default"):
    """Writes `path` (a .tar.gz) in the DFG 0.25 layout: <name>/dfg.json, <name>/variables/<label>.json, <name>/factors/<label>.json."""
    meta = {"description": "", "addHistory": list(fg.ls()), "solverParams": dict(getattr(fg, "solverParams", None) or {}, N=int(fg.N)),
            "solverParams_type": "SolverParams", "typePackedVariable": False, "typePackedFactor": False, "graphLabel": "factorgraph",
            "graphTags": [], "graphMetadata": {}}
    with tarfile.open(path, "w:gz") as tf:
        def add(name, obj):
            b = json.dumps(obj).encode("utf-8")
            ti = tarfile.TarInfo(name)
            ti.size = len(b)
            ti.mtime = int(time.time())
            tf.addfile(ti, io.BytesIO(b))
        add("dfg.json", meta)
        for lb in fg.ls():
            add("variables/%s.json" % lb, _variable_doc(fg, lb, solveKey))
        for flabel, labels, fac in fg.factors:
            add("factors/%s.json" % flabel, _factor_doc(fg, flabel, labels, fac))
    return path
