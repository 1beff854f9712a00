#!/usr/bin/env python3
"""Transcribes the reference's deterministic known-answer assertions for the four hot-path
residual functors into tests/golden/residual_kats.json.

Every case is DATA copied from an assertion in the reference test-suite (file:line given per
case, paths relative to /root/reference).  Poses are coordinates (x, y, theta) meaning the point
((x,y), R(theta)); measurements are tangent coordinates.  `expect` is what the reference asserts:
  {"r": [...], "atol": a}            -> isapprox(r, expect, atol=a)
  {"abs_r": [...], "atol": a}        -> isapprox(abs.(r), expect)
  {"r_idx": {"i": v, ...}, "atol": {...}}  -> per-component asserts
  {"norm_lt": a}                     -> norm(r) < a
Run:  python tests/golden/make_residual_kats.py
"""
import json
import math
import os

PI = math.pi
kats = {"pose2pose2": [], "pose2point2bearingrange": [], "pose3pose3": []}

# ---- Pose2Pose2: test/testParametricSimulated.jl ----
S = "test/testParametricSimulated.jl"
kats["pose2pose2"] += [
    dict(id="K1", src=S + ":37-40", z=[0, 0, -PI], p=[0, 0, 0], q=[0, 0, 0],
         expect=dict(abs_r=[0, 0, PI], atol=5e-8)),
    dict(id="K2", src=S + ":42-43", z=[0, 0, -PI], p=[0, 0, 0], q=[0, 0, -PI], expect=dict(r=[0, 0, 0], atol=1e-14)),
    dict(id="K3", src=S + ":45-46", z=[0, 0, -PI], p=[0, 0, 0], q=[0, 0, PI], expect=dict(r=[0, 0, 0], atol=1e-14)),
]
_z = [10.0, 0.0, 1.0471975511965976]
_p = [15.000000000016204, 8.660254037814505, 2.0943951023931953]
kats["pose2pose2"] += [
    dict(id="K4", src=S + ":105-130", z=_z, p=_p, q=[10.00004891350537, 17.320479835550103, 4.498439149584132e-6],
         expect=dict(r_idx={"0": 0.0, "1": 0.0}, abs_idx={"2": PI}, atol=1e-4)),
    dict(id="K5", src=S + ":133-137", z=_z, p=_p, q=[10.00004891350537, 17.320479835550103, PI],
         expect=dict(r_idx={"0": 0.0, "1": 0.0}, abs_idx={"2": 0.0}, atol=1e-4)),
    dict(id="K6", src=S + ":140-144", z=_z, p=_p, q=[10.00004891350537, 17.320479835550103, -PI],
         expect=dict(r_idx={"0": 0.0, "1": 0.0}, abs_idx={"2": 0.0}, atol=1e-4)),
]

# ---- Pose2Point2BearingRange: test/testBearingRange2D.jl ----
B = "test/testBearingRange2D.jl"
br = kats["pose2point2bearingrange"]
br += [
    dict(id="BR1", src=B + ":55-69", z=[0, 20.0], p=[0, 0, 0], l=[20.0, 0], expect=dict(norm_lt=1e-14)),
    dict(id="BR2", src=B + ":73-84", z=[PI / 2, 20.0], p=[0, 0, 0], l=[0, 20.0], expect=dict(norm_lt=1e-14)),
    dict(id="BR3", src=B + ":88-98", z=[0.0, 20.0], p=[0, 0, PI / 2], l=[0, 20.0], expect=dict(norm_lt=1e-14)),
    dict(id="BR4", src=B + ":102-116", z=[PI / 2, 20.0], p=[0, 0, -PI / 2], l=[20.0, 0], expect=dict(norm_lt=1e-14)),
]
# cases 5-20 use the literal points x1 = ([0,0],[1 0;0 1]) and x2 = ([0,0],[0 -1;1 0]) (:119-120);
# p_pt is the native point [tx,ty,R11,R21,R12,R22] (column-major R).
X1 = [0.0, 0.0, 1.0, 0.0, 0.0, 1.0]
X2 = [0.0, 0.0, 0.0, 1.0, -1.0, 0.0]
_s, _c = 10 * math.sin(0.001), 10 * math.cos(0.001)
_r2 = 10 / math.sqrt(2)
pairs = [
    ("BR5", ":123-130", [0.0, 10], X1, [10.0, 0], dict(r=[0, 0], atol=1e-9)),
    ("BR6", ":131-135", [0.0, 10], X2, [0.0, 10], dict(r=[0, 0], atol=1e-9)),
    ("BR7", ":138-145", [PI / 2, 10], X1, [0.0, 10], dict(r=[0, 0], atol=1e-9)),
    ("BR8", ":146-150", [PI / 2, 10], X2, [-10.0, 0], dict(r=[0, 0], atol=1e-9)),
    ("BR9", ":153-160", [PI, 10.0], X1, [-10.0, 0], dict(r=[0, 0], atol=1e-9)),
    ("BR10", ":161-165", [PI, 10.0], X2, [0.0, -10], dict(r=[0, 0], atol=1e-9)),
    ("BR11", ":168-175", [-PI / 2, 10.0], X1, [0.0, -10], dict(r=[0, 0], atol=1e-9)),
    ("BR12", ":176-180", [-PI / 2, 10.0], X2, [10.0, 0], dict(r=[0, 0], atol=1e-9)),
    ("BR13", ":186-193", [0.0, 10], X1, [11.0, 0], dict(r=[0, -1], atol=1e-9)),
    ("BR14", ":194-198", [0.0, 10], X2, [0.0, 11], dict(r=[0, -1], atol=1e-9)),
    ("BR15", ":201-208", [0.0, 10], X1, [9.0, 0], dict(r=[0, 1], atol=1e-9)),
    ("BR16", ":209-213", [0.0, 10], X2, [0.0, 9], dict(r=[0, 1], atol=1e-9)),
    ("BR17", ":216-226", [0.0, 10], X1, [_c, _s], dict(r_idx={"0": -0.001}, atol=1e-9, r_idx_loose={"1": 0.0}, atol_loose=0.1)),
    ("BR18", ":227-233", [0.0, 10], X2, [_s, _c], dict(r_idx={"0": 0.001}, atol=1e-9, r_idx_loose={"1": 0.0}, atol_loose=0.1)),
    ("BR19", ":238-246", [0.0, 10], X1, [_r2, _r2], dict(r=[-PI / 4, 0], atol=1e-9)),
    ("BR20", ":247-250", [0.0, 10], X2, [_r2, _r2], dict(r=[PI / 4, 0], atol=1e-9)),
]
for i, s, z, ppt, l, e in pairs:
    br.append(dict(id=i, src=B + s, z=z, p_pt=ppt, l=l, expect=e))

# ---- Pose3Pose3 ----
T = "test/threeDimLinearProductTest.jl"
p3 = kats["pose3pose3"]
p3 += [
    # q given as a literal point ((10,0,0), I); X = hat([10,0,0,0,0,0])
    dict(id="P3K1", src=T + ":150-156", z=[10.0, 0, 0, 0, 0, 0], p_coords=[0] * 6,
         q_pt=[10.0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1], expect=dict(norm_lt=1e-10)),
    # q = getPoint(Pose3, c), X = hat(c), c = [10,0,0,pi,pi,pi]
    dict(id="P3K2", src=T + ":162-167", z=[10.0, 0, 0, PI, PI, PI], p_coords=[0] * 6,
         q_coords=[10.0, 0, 0, PI, PI, PI], expect=dict(norm_lt=1e-10)),
]
# test/testPartialPose3.jl:398-436: wTx1 = identity, wTx2 = (xyz, RotXYZ(rpy)), X = log(eps, x1Tx2)
phi = theta = psi = 0.1
table = [
    [10., 0, 0, 0, 0, 0], [0., 10, 0, 0, 0, 0], [0., 0, 10, 0, 0, 0],
    [10., 0, 0, phi, 0, 0], [0., 10, 0, phi, 0, 0], [0., 0, 10, phi, 0, 0],
    [10., 0, 0, 0, theta, 0], [0., 10, 0, 0, theta, 0], [0., 0, 10, 0, theta, 0], [0., 15, 10, 0, theta, 0],
    [10., 0, 0, 0, 0, psi], [0., 10, 0, 0, 0, psi], [0., 0, 10, 0, 0, psi], [0., 15, 10, 0, 0, psi],
]
for k, row in enumerate(table):
    p3.append(dict(id="P3K%d" % (k + 3), src="test/testPartialPose3.jl:398-436", xyz_rpy=row,
                   note="q = (xyz, RotXYZ(rpy)); p = identity; z = vee(log(identity, q))", expect=dict(norm_lt=1e-10)))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "residual_kats.json")
with open(out, "w") as f:
    json.dump(kats, f, indent=1)
print("wrote", out, {k: len(v) for k, v in kats.items()})
