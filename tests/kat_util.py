"""Shared checker for tests/golden/residual_kats.json (same checks run against the CPU oracle
and against the HIP residual entry points)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kats():
    with open(os.path.join(GOLDEN, "residual_kats.json")) as f:
        return json.load(f)


def check_expect(r, e, label=""):
    r = np.asarray(r, dtype=float)
    if "norm_lt" in e:
        assert np.linalg.norm(r) < e["norm_lt"], (label, r)
    if "r" in e:
        assert np.allclose(r, e["r"], rtol=0, atol=e["atol"]), (label, r)
    if "abs_r" in e:
        assert np.allclose(np.abs(r), e["abs_r"], rtol=0, atol=e["atol"]), (label, r)
    for k, v in e.get("r_idx", {}).items():
        assert abs(r[int(k)] - v) <= e["atol"], (label, r)
    for k, v in e.get("abs_idx", {}).items():
        assert abs(abs(r[int(k)]) - v) <= e["atol"], (label, r)
    for k, v in e.get("r_idx_loose", {}).items():
        assert abs(r[int(k)] - v) <= e["atol_loose"], (label, r)


def rotxyz_np(r, p, y):
    """Rotations.jl RotXYZ(r,p,y) = Rx(r) @ Ry(p) @ Rz(y), independent numpy restatement."""
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def pose3_case_inputs(case):
    """-> (z coords(6), p point(12), q point(12)) using scipy's independent SO(3) exp/log."""
    from scipy.spatial.transform import Rotation as Rot

    def pt_from_coords(c):
        c = np.asarray(c, dtype=float)
        R = Rot.from_rotvec(c[3:]).as_matrix()
        return np.concatenate([c[:3], R.flatten(order="F")])

    if "xyz_rpy" in case:
        row = case["xyz_rpy"]
        R = rotxyz_np(*row[3:])
        q = np.concatenate([row[:3], R.flatten(order="F")])
        z = np.concatenate([row[:3], Rot.from_matrix(R).as_rotvec()])
        p = pt_from_coords([0] * 6)
        return z, p, q
    z = np.asarray(case["z"], dtype=float)
    p = pt_from_coords(case["p_coords"])
    q = np.asarray(case["q_pt"], dtype=float) if "q_pt" in case else pt_from_coords(case["q_coords"])
    return z, p, q
