"""Device math of rome.jl_amd/csrc/rome_device_math.hpp (fast_sincos, wrap_pi, fast_sqrt, fast_log, fast_exp_neg, fast_atan2, quaternion
Exp/Log) against numpy on 2e5 random arguments spanning the magnitudes the kernels see: compiled on the fly with hipcc."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_math_against_numpy(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "math_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-o", exe,
                           os.path.join(ROOT, "tests", "hip", "math_check.hip")])
    rng = np.random.default_rng(0)
    n = 200000
    x = rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 2.5, n)
    y = rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 2.5, n)
    x[8:2008] = rng.uniform(-1e5, 1e5, 2000)                                         # sin/cos: headings far from the principal range
    x[:8] = [0.0, np.pi, -np.pi, 1e-300, 3.0 * np.pi, -0.0, 745.0, 1.0]
    y[:8] = [0.0, 0.0, 1e-300, 1.0, -2.0, 5.0, 1e-12, 1.0]
    np.concatenate([x, y]).tofile(str(tmp_path / "in.bin"))
    out = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert out.returncode == 0 and "math_check ok" in out.stdout, out.stdout + out.stderr
    o = np.fromfile(str(tmp_path / "out.bin")).reshape(9, n)
    assert np.abs(o[0] - np.sin(x)).max() < 5e-16 + 1e-18 * np.abs(x).max() and np.abs(o[1] - np.cos(x)).max() < 5e-16 + 1e-18 * np.abs(x).max()
    w = np.arctan2(np.sin(x), np.cos(x))
    dw = np.abs(o[2] - w); dw = np.minimum(dw, np.abs(dw - 2 * np.pi))            # the ±π tie may land on either end
    assert dw.max() < 1e-13 and np.abs(o[2]).max() <= np.pi + 1e-15
    assert np.abs(o[3] - np.sqrt(np.abs(x))).max() <= 2e-16 * np.sqrt(np.abs(x)).max() and (np.abs(o[3] / np.maximum(np.sqrt(np.abs(x)), 1e-300) - 1.0)[np.abs(x) > 1e-290] < 4e-16).all()
    ref = np.log(np.abs(y) + 1e-300)
    assert (np.abs(o[4] - ref) <= 4e-16 * np.maximum(1.0, np.abs(ref))).all()
    ref = np.exp(-np.abs(x))
    assert (np.abs(o[5] - ref) <= 5e-16 * ref + 1e-320).all()
    ref = np.arctan2(y, x)
    keep = ~((x == 0) & (y == 0)) & ~(np.signbit(x) & (x == 0))                     # atan2(±0, -0) conventions are not reproduced
    assert np.abs(o[6] - ref)[keep].max() < 7e-16
    wn = 0.01 * np.sqrt(x * x + y * y + (x - y) ** 2)                               # |ω| of the quaternion round trip
    assert np.abs(o[7])[wn < 3.0].max() < 2e-15 and np.abs(o[8]).max() < 1e-15      # Log(Exp(ω)) = ω below π; unit norm always
