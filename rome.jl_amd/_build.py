"""Builds librome_mi355.so (hipcc, gfx950 only) in-tree next to this file."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "librome_mi355.so")
SOURCES = [os.path.join(HERE, "csrc", f) for f in ("rome_kernels.hip", "rome_parametric.hip", "rome_product.hip", "rome_kde.hip", "rome_capi.hip")]
DEPS = SOURCES + [os.path.join(HERE, "csrc", f) for f in ("rome_kernels.h", "rome_device_math.hpp")] + \
    [os.path.join(os.path.dirname(HERE), "include", "rome_mi355.h")]


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-pass-failed", "-o", SO] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
