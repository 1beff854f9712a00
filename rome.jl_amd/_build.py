"""Builds librome_mi355.so (hipcc, gfx950 only) in-tree next to this file.

Every .hip translation unit is compiled to its own object (in parallel: hipcc takes 10-50 s per file) and the
objects are linked into one shared library; objects are reused when neither their source nor a header changed."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "librome_mi355.so")
OBJDIR = os.path.join(HERE, "build")
UNITS = ("rome_kernels.hip", "rome_parametric.hip", "rome_product.hip", "rome_kde.hip", "rome_gibbs.hip", "rome_capi.hip")
SOURCES = [os.path.join(HERE, "csrc", f) for f in UNITS if os.path.exists(os.path.join(HERE, "csrc", f))]
HEADERS = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc"))) if f.endswith((".h", ".hpp"))] + \
    [os.path.join(os.path.dirname(HERE), "include", "rome_mi355.h")]
DEPS = SOURCES + HEADERS
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]


def _obj(src):
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJDIR, exist_ok=True)
    th = max(os.path.getmtime(h) for h in HEADERS)

    def compile_one(src):
        obj = _obj(src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(th, os.path.getmtime(src)):
            return
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
